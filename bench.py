#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched rigid-body hot path on MI355X (BASELINE.json metric).

Workload (config.workload): Anymal-class quadruped (13 bodies / 12 revolute joints + free base / 13 cylinder
colliders + ground plane), SolverXPBD(iterations=2), dt = 1e-3, 4096 environments PER GPU.  One "step" is one
frame of the reference's caller loop (newton/examples/basic/example_basic_urdf.py:117-141): 10 substeps of
{clear_forces; CollisionPipeline.collide; SolverXPBD.step; swap}, executed as ONE launch of the fused
gfx950 rollout kernel (the CUDA-graph replacement).  env-steps/s = envs * substeps * steps / T  (Newton's
world-steps/s, docs/guide/development.rst:818-824).

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Environments shard embarrassingly: each rank owns its own 4096 envs, no data-path collective (weak scaling).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ENVS_PER_GPU = 4096
SUBSTEPS = 10
DT = 1e-3
HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def algorithmic_bytes_per_env_step(t, contacts_per_env: float) -> float:
    """Compulsory HBM bytes if each env's working set is touched once per substep:
    state_in 76B + state_out 52B + clear_forces 24B + body params 100B per body, 85 B/joint, 40 B/dof,
    72 B/shape (incl. the shared plane), 8 B/pair, 2*80 B per contact (write + read of the Contacts boundary)."""
    B, J, D, S, P = t.nb, t.nj, t.nd, t.ns + t.ng, t.np
    return (76 + 52 + 24 + 100) * B + 85 * J + 40 * D + 72 * S + 8 * P + 160.0 * contacts_per_env


def cpu_baseline(envs_per_core=256, sample_substeps=2500, max_cores=32):
    """The C++ oracle (a restatement of Newton's kernels, NOT Newton/Warp itself) on the host cores: one independent
    env shard per core (the path shards embarrassingly on the CPU too), ctypes releases the GIL during each call."""
    import threading

    from oracle_bridge import Oracle, OracleState
    from scenes import quadruped_scene

    cores = max(1, min(os.cpu_count() or 1, max_cores))
    shards = []
    for k in range(cores):
        model = quadruped_scene(envs_per_core, seed=100 + k)
        o = Oracle(model)
        shards.append((o, OracleState(model), OracleState(model), o.contacts(), o.control()))

    def run(shard):
        o, s0, s1, ct, ctrl = shard
        o.xpbd_rollout(s0, s1, ctrl, ct, DT, sample_substeps)  # the whole loop in one foreign call (GIL released)

    threads = [threading.Thread(target=run, args=(sh,)) for sh in shards]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    T = time.perf_counter() - t0
    return {
        "value": cores * envs_per_core * sample_substeps / T, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{cores} shards x {envs_per_core} envs x {sample_substeps} substeps of the same quadruped XPBD workload, "
                  f"C++ oracle (restatement of Newton's kernels), one shard per host core, {T:.1f} s wall",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--envs-per-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["quadruped", "box_stack", "quadruped_featherstone", "hull_bin"], default="quadruped",
                    help="quadruped = the BASELINE.json metric (default); box_stack = config C2 (convex MPR/GJK path), "
                         "a secondary measurement that is never the headline value")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))

    import newton_amd as nt
    from scenes import quadruped_scene

    # every rank owns its own shard of environments (distinct seed => distinct per-env jitter)
    if args.workload in ("quadruped", "quadruped_featherstone"):
        model = quadruped_scene(args.envs_per_gpu, device=f"cuda:{local_rank}", seed=1 + rank)
        iterations, workload_name = 2, (
            "Anymal-class quadruped (in-repo stand-in geometry: 13 bodies, 12 revolute + free base, "
            "13 cylinder colliders + ground plane)")
    elif args.workload == "hull_bin":
        from scenes import hull_bin_scene

        # C5 without the SDF / hydroelastic contact models: 64 hulls in a five-wall bin, 2 336 candidate pairs per env
        # (pass --envs-per-gpu 2048 for the BASELINE.json size; contact records then take ~3 GB of HBM)
        model = hull_bin_scene(args.envs_per_gpu, 64, device=f"cuda:{local_rank}", seed=2 + rank)
        iterations, workload_name = 2, ("C5 geometry without SDF / hydroelastic: 64 convex hulls (16-32 vertices) in a five-wall "
                                        "bin, all 2 336 pairs per env through MPR/GJK + manifold, contact records in HBM")
    else:
        from scenes import box_stack_scene

        model = box_stack_scene(args.envs_per_gpu, device=f"cuda:{local_rank}", seed=1 + rank)
        iterations, workload_name = 4, "C2: 8-box stack on a ground plane (box-box pairs through MPR/GJK + manifold)"
    s0, s1 = model.state(), model.state()
    ctrl = model.control()
    pipe = nt.CollisionPipeline(model, envs_per_block=args.envs_per_block)
    contacts = pipe.contacts()
    if args.workload == "quadruped_featherstone":
        # C3: clear_forces; collide; SolverFeatherstone.step; swap
        fs = nt.solvers.SolverFeatherstone(model, envs_per_block=args.envs_per_block)
        workload_name = workload_name.replace("SolverXPBD", "SolverFeatherstone")

        solver = fs  # SolverFeatherstone.rollout: the same loop fused into one launch
    else:
        solver = nt.solvers.SolverXPBD(model, iterations=iterations, envs_per_block=args.envs_per_block)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.rollout(s0, s1, ctrl, contacts, DT, SUBSTEPS)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        solver.rollout(s0, s1, ctrl, contacts, DT, SUBSTEPS)
    ev1.record()
    barrier()
    T = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream (torch current stream)

    from newton_amd.sharding import max_over_ranks

    T = max_over_ranks(T, device=f"cuda:{local_rank}")  # MAX over ranks (no-op at N=1)

    # validity gate of the reference benchmark (asv/benchmarks/benchmark_metrics.py:67-99): finite state,
    # normalised quaternions
    q = s0.body_q
    ok = bool(torch.isfinite(q).all()) and bool(((q[:, 3:].norm(dim=1) - 1.0).abs() < 1e-3).all())
    c_per_env = float(contacts.rigid_contact_count_per_env.float().mean().item())

    if rank == 0:
        t = model.env
        total_env_steps = world * args.envs_per_gpu * SUBSTEPS * args.steps
        value = total_env_steps / T
        bytes_per_env_step = algorithmic_bytes_per_env_step(t, c_per_env)
        launch_bytes = bytes_per_env_step * args.envs_per_gpu * SUBSTEPS
        achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("xpbd_rollout_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec at 4096 batched envs (Anymal, XPBD)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * T / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "valid_state": ok,
            "config": {
                "workload": f"{workload_name}, " + ("SolverFeatherstone defaults" if args.workload == "quadruped_featherstone"
                                                    else f"SolverXPBD iterations={iterations}") + ", dt=1e-3, "
                            f"{args.envs_per_gpu} envs per GPU, 1 step = 1 frame = {SUBSTEPS} substeps of "
                            "clear_forces+collide+step fused in one rollout launch",
                "envs_per_gpu": args.envs_per_gpu, "substeps_per_step": SUBSTEPS, "parallelism": f"env-shard x{world}",
                "mean_contacts_per_env": c_per_env,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "kernel": "xpbd_rollout_kernel", "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_env_step": bytes_per_env_step, "algorithmic_bytes_per_launch": launch_bytes,
            },
        }
        if args.workload != "quadruped":
            out["metric"] = f"env-steps/sec, {args.workload} (secondary; not the BASELINE.json metric)"
        if not args.no_cpu_baseline and world == 1 and args.workload == "quadruped":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
