#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched rigid-body hot path on MI355X (BASELINE.json metric).

Workload (config.workload): Anymal-class quadruped (13 bodies / 12 revolute joints + free base / 13 cylinder
colliders + ground plane), SolverXPBD(iterations=2), dt = 1e-3, 4096 environments PER GPU.  One "step" is one
frame of the reference's caller loop (newton/examples/basic/example_basic_urdf.py:117-141): 10 substeps of
{clear_forces; CollisionPipeline.collide; SolverXPBD.step; swap}, executed as ONE launch of the fused
gfx950 rollout kernel (the CUDA-graph replacement).  env-steps/s = envs * substeps * steps / T  (Newton's
world-steps/s, docs/guide/development.rst:818-824).

The scene is PRE-SETTLED before any warm-up or timed step: the robots are lowered so that their feet touch the ground and
--settle-frames untimed frames are run, so even a 20-step run measures the standing regime (16 live contacts per
environment), not free fall.  The validity gate is the reference benchmark's (asv/benchmarks/simulation/
bench_quadruped_xpbd.py:58-66 + example_basic_urdf.py:145-162): finite state, unit quaternions, body speeds <= 0.3,
root height 0.46 +- 0.01.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Environments shard embarrassingly: ONE global model of N x 4096 environments is built (same seed on every rank) and rank r
keeps worlds shard_range(total, r, N) of it (newton_amd.sharding.shard_model); no data-path collective (weak scaling).

Secondary workloads (--workload; never the headline value): quadruped_convex (config C4's convex-convex variant),
box_stack (C2), quadruped_featherstone (C3), hull_bin (C5's geometry through MPR/GJK).  --sweep runs the BASELINE.md
section 5 env-count sweep of the headline workload and writes a table instead of the single JSON line.
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
HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
BASE_ENVS = 4096        # larger shards are tiled from a 4096-env build (the Python builder needs ~0.7 ms per env)

# name -> (scene function, solver, iterations, dt, dominant kernel, root drop [m], default settle frames, description)
WORKLOADS = {
    "quadruped": dict(solver="xpbd", iterations=2, dt=1e-3, kernel="xpbd_rollout_kernel<16,false>", drop=0.22, settle=100,
                      name="Anymal-class quadruped (in-repo stand-in geometry: 13 bodies, 12 revolute + free base, "
                           "13 cylinder colliders + ground plane)"),
    "quadruped_convex": dict(solver="xpbd", iterations=2, dt=1e-3, kernel="xpbd_rollout_kernel<16,true>", drop=0.22, settle=100,
                             name="C4 convex-convex variant: the same quadruped with box links on a static box slab "
                                  "(13 box-box pairs per env through MPR/GJK + manifold)"),
    "quadruped_featherstone": dict(solver="featherstone", iterations=0, dt=1e-3, kernel="featherstone_rollout_kernel<4,false>",
                                   drop=0.22, settle=100,
                                   name="C3: Anymal-class quadruped, SolverFeatherstone (generalized coordinates, LDS-resident "
                                        "mass matrix), cylinder colliders + ground plane"),
    "box_stack": dict(solver="xpbd", iterations=4, dt=1.0 / 240.0, kernel="xpbd_rollout_kernel<16,true>", drop=0.0, settle=10,
                      name="C2: 8-box stack on a ground plane (box-box pairs through MPR/GJK + manifold)"),
    "hull_bin": dict(solver="xpbd", iterations=2, dt=1.0 / 1200.0, kernel="xpbd_rollout_kernel<1,true,true>", drop=0.0, settle=30,
                     name="C5 geometry without SDF / hydroelastic: 64 convex hulls (16-32 vertices) in a five-wall bin, all "
                          "2 336 pairs per env through MPR/GJK + manifold, contact records in HBM"),
    # config C5 with its contact model: every pair through the mesh-SDF leg of CollisionPipeline.collide (candidate pairs per
    # world -> edge-vs-SDF narrow phase + global reduction -> deterministic rows), SolverXPBD consumes the rows inside its step
    # kernel.  The collide chain is several launches, so the frame is the reference's loop launch by launch (loop=True)
    "sdf_bin": dict(solver="xpbd", iterations=2, dt=1.0 / 1200.0, kernel="mesh_sdf_collide_reduced_kernel", drop=0.0, settle=40,
                    loop=True, envs=2048,
                    name="C5: 64 convex hulls (16-32 vertices, uint16 texture SDFs) in a five-wall bin of SDF boxes, every pair "
                         "through the SDF narrow phase + global contact reduction inside CollisionPipeline.collide(broad_phase='sap'), contact gap 5 mm"),
    "hydro_bin": dict(solver="xpbd", iterations=2, dt=1.0 / 1200.0, kernel="hydro_pairs_kernel<true>", drop=0.0, settle=40, loop=True, envs=256,
                      name="C5 with hydroelastic contacts: the same bin, every shape HYDROELASTIC (kh = 1e10): every pair through the "
                           "SDF-SDF leg (SAT, octree in LDS, marching cubes) with HydroelasticSDF.Config() as it comes (reduce_contacts, "
                           "pre_prune_contacts, normal_matching), contact gap 5 mm"),
    "hydro_bin_faces": dict(solver="xpbd", iterations=2, dt=1.0 / 1200.0, kernel="hydro_pairs_kernel<false>", drop=0.0, settle=40, loop=True,
                            envs=256,
                            name="the hydroelastic bin with HydroelasticSDF.Config(reduce_contacts=False): every marching-cubes face is a contact row"),
    # the headline scene through the per-call API an RL loop with per-substep control uses: collide and step as separate launches
    "quadruped_api": dict(solver="xpbd", iterations=2, dt=1e-3, kernel="xpbd_step_kernel<16,false> + collide_kernel<16,false>",
                          drop=0.22, settle=100, loop=True,
                          name="Anymal-class quadruped through the per-call API: clear_forces + CollisionPipeline.collide + "
                               "SolverXPBD.step as separate launches per substep"),
}


def algorithmic_bytes_per_env_step(t, contacts_per_env: float, generalized: bool = False) -> float:
    """Compulsory HBM bytes if each env's working set is touched once per substep:
    state_in 76B + state_out 52B + clear_forces 24B + body params 100B per body, 85 B/joint, 40 B/dof,
    72 B/shape (incl. the shared plane), 8 B/pair, 2*80 B per contact (write + read of the Contacts boundary);
    generalized-coordinate solvers add joint_q / joint_qd in + out."""
    B, J, D, S, P = t.nb, t.nj, t.nd, t.ns + t.ng, t.np
    extra = 2 * (t.nc + t.nd) * 4 if generalized else 0
    return (76 + 52 + 24 + 100) * B + 85 * J + 40 * D + 72 * S + 8 * P + 160.0 * contacts_per_env + extra


def build_id() -> str:
    from newton_amd import _lib

    return _lib.load().nt_build_info().decode()


def measured_traffic(workload: str, envs: int):
    """HBM bytes per launch from the PMC passes (tools/pmc_traffic.py), only if they were taken on THIS build of the
    library (same nt_build_info source hash) for this workload and env count; otherwise None."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):  # newest round first
        try:
            rec = json.load(open(path)).get(f"{workload}@{envs}")
            if rec and rec.get("build_id") == build_id():
                rec["source"] = "profiles/" + os.path.basename(path)
                return rec
        except Exception:
            pass
    return None


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
        o.xpbd_rollout(s0, s1, ctrl, ct, 1e-3, sample_substeps)  # the whole loop in one foreign call (GIL released)

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


def build_shard(workload: str, envs_per_gpu: int, rank: int, world: int, device: str):
    """This rank's shard of ONE global model (same seeds on every rank): worlds shard_range(total, rank, world) of it.  The
    quadruped workloads build only those worlds; shards above BASE_ENVS environments are tiled from the rank's BASE_ENVS-env slice
    (10^5..10^6 envs would take minutes in the Python builder)."""
    import newton_amd as nt
    from newton_amd.sharding import shard_model
    from newton_amd.worlds import tile_worlds
    import scenes

    base = min(envs_per_gpu, BASE_ENVS)
    if envs_per_gpu % base:
        raise SystemExit(f"--envs-per-gpu above {BASE_ENVS} must be a multiple of it")
    total = base * world
    drop = WORKLOADS[workload]["drop"]
    if workload in ("quadruped", "quadruped_featherstone", "quadruped_api", "quadruped_convex"):
        # the rank builds ONLY its own worlds of the global scene (per-world jitter drawn for all worlds, sliced): the same model
        # shard_model(global, rank, world) gives (tests/test_sharding_gloo.py), without 23 s of Python builder per rank at 8 x 4096
        from newton_amd.sharding import shard_range

        b, e = shard_range(total, rank, world)
        fn = scenes.quadruped_convex_scene if workload == "quadruped_convex" else scenes.quadruped_scene
        m = fn(total, seed=1, world_range=(b, e))
        if drop > 0.0:
            m.joint_q.reshape(e - b, -1)[:, 2] -= drop
            m.body_q, m.body_qd = nt.articulation.eval_fk_numpy(m, m.joint_q, m.joint_qd)
        m = shard_model(m, 0, 1, device=device)
        if envs_per_gpu > base:
            m = tile_worlds(m, envs_per_gpu // base, device=device, filter_pairs=False)
        return m
    if workload in ("hydro_bin", "hydro_bin_faces"):
        g = scenes.hull_bin_scene(total, 64, seed=2, sdf=True, mu=0.5, shape_cfg=dict(gap=0.005), hydroelastic=True)
    elif workload == "sdf_bin":
        # contact gap 5 mm (Newton's default rigid_gap of 0.1 m is larger than a hull: every pair of the bin would be a candidate)
        g = scenes.hull_bin_scene(total, 64, seed=2, sdf=True, mu=0.5, shape_cfg=dict(gap=0.005))
    elif workload == "box_stack":
        g = scenes.box_stack_scene(total, seed=1)
    else:
        g = scenes.hull_bin_scene(total, 64, seed=2)
    if drop > 0.0:  # feet onto the ground: the free fall from z = 0.7 is not the regime the metric is quoted on
        g.joint_q.reshape(total, -1)[:, 2] -= drop
        g.body_q, g.body_qd = nt.articulation.eval_fk_numpy(g, g.joint_q, g.joint_qd)
    m = shard_model(g, rank, world, device=device)
    if envs_per_gpu > base:
        m = tile_worlds(m, envs_per_gpu // base, device=device, filter_pairs=False)
    return m


def validity_gate(workload: str, model, state) -> dict:
    """asv/benchmarks/benchmark_metrics.py:67-99 (finite, |quat| = 1 +- 1e-3, speed bounds 0.3 from
    bench_quadruped_xpbd.py:58-66) + example_basic_urdf.py:145-162 (root height 0.46 +- 0.01) for the quadruped workloads."""
    q, qd = state.body_q, state.body_qd
    finite = bool(torch.isfinite(q).all()) and bool(torch.isfinite(qd).all())
    quat_err = float((q[:, 3:].norm(dim=1) - 1.0).abs().max())
    lin = float(qd[:, :3].norm(dim=1).max())
    ang = float(qd[:, 3:].norm(dim=1).max())
    gate = {"finite": finite, "max_quat_norm_error": quat_err, "max_linear_speed": lin, "max_angular_speed": ang}
    ok = finite and quat_err < 1e-3
    if workload.startswith("quadruped"):
        nb = model.env.nb
        root_z = q[::nb, 2]
        gate["root_height_min"], gate["root_height_max"] = float(root_z.min()), float(root_z.max())
        gate["speed_limit"], gate["root_height_target"] = 0.3, [0.45, 0.47]
        if workload == "quadruped":  # the reference's gate is stated for this scene
            ok = ok and lin <= 0.3 and ang <= 0.3 and gate["root_height_min"] > 0.45 and gate["root_height_max"] < 0.47
    gate["ok"] = bool(ok)
    return gate


def run(args, rank, local_rank, world, dist):
    import newton_amd as nt
    from newton_amd.sharding import max_over_ranks

    W = WORKLOADS[args.workload]
    device = f"cuda:{local_rank}"
    model = build_shard(args.workload, args.envs_per_gpu, rank, world, device)
    dt = W["dt"]
    s0, s1 = model.state(), model.state()
    ctrl = model.control()
    extra = {}
    if args.workload == "hydro_bin":
        extra = dict(sdf_hydroelastic_config=nt.geometry.HydroelasticSDF.Config(), sdf_contacts_per_shape=400, sdf_hydro_faces_per_shape=600)
    elif args.workload == "hydro_bin_faces":
        extra = dict(sdf_hydroelastic_config=nt.geometry.HydroelasticSDF.Config(reduce_contacts=False), sdf_contacts_per_shape=400)
    pipe = nt.CollisionPipeline(model, envs_per_block=args.envs_per_block,
                                broad_phase="sap" if args.workload in ("sdf_bin", "hydro_bin", "hydro_bin_faces") else None, **extra)
    contacts = pipe.contacts()
    if W["solver"] == "featherstone":
        solver = nt.solvers.SolverFeatherstone(model, envs_per_block=args.envs_per_block)
    else:
        solver = nt.solvers.SolverXPBD(model, iterations=W["iterations"], envs_per_block=args.envs_per_block)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    st = [s0, s1]

    class _Frame:  # one frame = SUBSTEPS substeps: the fused rollout launch, or (loop workloads) the reference loop call by call
        @staticmethod
        def rollout(*_a):
            if not W.get("loop"):
                return solver_.rollout(st[0], st[1], ctrl, contacts, dt, SUBSTEPS)
            for _ in range(SUBSTEPS):
                st[0].clear_forces()
                pipe.collide(st[0], contacts)
                solver_.step(st[0], st[1], ctrl, contacts, dt)
                st[0], st[1] = st[1], st[0]
            return st[0]

    solver_, solver = solver, _Frame
    settle = W["settle"] if args.settle_frames < 0 else args.settle_frames
    t_settle = time.perf_counter()
    for _ in range(settle):  # untimed, outside warm-up: reach the standing regime
        solver.rollout(s0, s1, ctrl, contacts, dt, SUBSTEPS)
    torch.cuda.synchronize()
    # ... and keep the (settled) scene running until the GPU has been busy for half a second: a fresh box reaches its
    # sustained clocks only after a few hundred milliseconds of load (same-box measurements: the first run after idle is
    # 7 % slower than every later one)
    while settle > 0 and time.perf_counter() - t_settle < 0.5:
        for _ in range(20):
            solver.rollout(s0, s1, ctrl, contacts, dt, SUBSTEPS)
        torch.cuda.synchronize()
    graph = None
    if args.graph and W.get("loop"):
        graph = nt.graph.capture(lambda: solver.rollout(s0, s1, ctrl, contacts, dt, SUBSTEPS), warmup=0)

        class _Replay:
            @staticmethod
            def rollout(*_a):
                graph.launch()

        solver = _Replay
    for _ in range(args.warmup):
        solver.rollout(s0, s1, ctrl, contacts, dt, SUBSTEPS)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        solver.rollout(s0, s1, ctrl, contacts, dt, SUBSTEPS)
    ev1.record()
    barrier()
    T = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream (torch current stream)
    dry = os.environ.get("NT_BENCH_DRY_SINGLE_GPU", "0") == "1"
    T = max_over_ranks(T, device=None if dry else device)  # MAX over ranks (no-op at N=1)

    solver = solver_
    gate = validity_gate(args.workload, model, st[0] if W.get("loop") else s0)
    c_per_env = float(contacts.rigid_contact_count_per_env.float().mean().item())
    sdf_info = None
    if getattr(contacts, "_flat", None) is not None:  # the SDF leg's rows count as contacts of the boundary
        f = contacts._flat
        live = int((f.shape0[: int(f.row_start[-1].item())] >= 0).sum().item())
        c_per_env += live / model.env.env_count
        sdf_info = pipe._sdf_leg.overflow(f)
        sdf_info["live_rows"] = live
        q = (st[0] if W.get("loop") else s0).body_q
        gate["z_min"], gate["z_max"] = float(q[:, 2].min()), float(q[:, 2].max())
        gate["ok"] = bool(gate["ok"] and gate["z_min"] > -0.01 and gate["z_max"] < 1.0 and gate["max_linear_speed"] < 3.0
                          and not sdf_info["overflow"])
    if rank != 0:
        return None
    t = model.env
    total_env_steps = world * args.envs_per_gpu * SUBSTEPS * args.steps
    bytes_per_env_step = algorithmic_bytes_per_env_step(t, c_per_env, W["solver"] == "featherstone")
    launch_bytes = bytes_per_env_step * args.envs_per_gpu * SUBSTEPS
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
    epb = int(model.device_model().envs_per_block) if W["solver"] != "featherstone" else 4
    kernel = W["kernel"].replace("<16,", f"<{epb},") if W["solver"] != "featherstone" else W["kernel"]
    if W["solver"] == "xpbd" and not W.get("loop"):  # the launch shape the library really dispatches for this model (name as profilers print it)
        import ctypes as C
        from newton_amd import _lib
        dm, shape = model.device_model(), (C.c_int32 * 5)()
        p_ = solver._params()
        _lib.check(dm.lib.nt_xpbd_rollout_shape(C.byref(dm.desc), C.byref(p_), None, shape), "nt_xpbd_rollout_shape")
        b = lambda x: "true" if x else "false"  # noqa: E731
        kernel = (f"xpbd_rollout_kernel<{shape[0] + 256 * shape[3]}, {b(shape[4] & 1)}, {b(shape[4] & 2)}, {shape[1]}, "
                  f"{shape[2]}>")
    roof = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "traffic": None, "kernel": kernel, "kernel_ms": kernel_ms,
        "frac_algorithmic": achieved / HBM_PEAK_GBPS, "frac_measured": None,
        "algorithmic_bytes_per_env_step": bytes_per_env_step, "algorithmic_bytes_per_launch": launch_bytes,
        "build_id": build_id(),
    }
    rec = measured_traffic(args.workload, args.envs_per_gpu)
    if rec is not None:  # counters taken on this very build (tools/pmc_traffic.py): bytes that really moved
        roof["traffic"] = rec["bytes_per_launch"]
        roof["frac_measured"] = min(rec["bytes_per_launch"], launch_bytes) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        roof["traffic_source"] = rec["source"]
    out = {
        "metric": "env-steps/sec at 4096 batched envs (Anymal, XPBD)", "value": total_env_steps / T, "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * T / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "valid_state": gate["ok"], "validity_gate": gate,
        "config": {
            "workload": f"{W['name']}, " + ("SolverFeatherstone defaults" if W["solver"] == "featherstone"
                                            else f"SolverXPBD iterations={W['iterations']}") + f", dt={dt:.6g}, "
                        f"{args.envs_per_gpu} envs per GPU, pre-settled ({settle} untimed frames + >= 0.5 s of the same frames for the clocks, feet on the ground), "
                        f"1 step = 1 frame = {SUBSTEPS} substeps of clear_forces+collide+step fused in one rollout launch",
            "envs_per_gpu": args.envs_per_gpu, "substeps_per_step": SUBSTEPS, "parallelism": f"env-shard x{world}",
            "mean_contacts_per_env": c_per_env, "settle_frames": settle,
        },
        "roofline": roof,
    }
    if W.get("loop"):
        out["config"]["workload"] = out["config"]["workload"].replace(
            "fused in one rollout launch", "as separate launches (per-call API)" + (", the frame replayed as one hipGraph" if args.graph else ""))
        out["config"]["hip_graph"] = bool(args.graph)
        roof["kernel_ms_is"] = "whole frame (every launch of the 10 substeps), not one kernel: see the rocprof kernel stats for the split"
    if sdf_info is not None:
        out["sdf_leg"] = sdf_info
    if args.workload != "quadruped":
        out["metric"] = f"env-steps/sec, {args.workload} (secondary; not the BASELINE.json metric)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="environments per GPU (0: the workload's default, 4096 unless stated)")
    ap.add_argument("--envs-per-block", type=int, default=0)
    ap.add_argument("--settle-frames", type=int, default=-1, help="untimed frames before warm-up (-1: the workload's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="call-by-call workloads (quadruped_api, sdf_bin, hydro_bin): record the frame's launches into one hipGraph "
                         "after settling and replay it per step, as the reference's examples do with wp.ScopedCapture")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="quadruped",
                    help="quadruped = the BASELINE.json metric (default); the others are secondary measurements")
    ap.add_argument("--sweep", default="", help="comma-separated env counts (e.g. 4096,16384,65536,262144,1048576): run the "
                    "headline workload at each size on one GPU and print one JSON line per size + write --sweep-out")
    ap.add_argument("--sweep-out", default=os.path.join(ROOT, "gpurun_out", "env_sweep.json"))
    args = ap.parse_args()
    if args.envs_per_gpu <= 0:
        args.envs_per_gpu = WORKLOADS[args.workload].get("envs", ENVS_PER_GPU)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node N")
    # NT_BENCH_DRY_SINGLE_GPU=1 (logic check of the N > 1 path on a 1-GPU box, never a measurement): every rank uses cuda:0
    # and the process group runs on gloo; the printed line carries "dry_run": true
    dry = os.environ.get("NT_BENCH_DRY_SINGLE_GPU", "0") == "1"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))

    if args.sweep:
        rows = []
        for n in [int(x) for x in args.sweep.split(",")]:
            args.envs_per_gpu = n
            steps = max(20, min(args.steps, int(2.0e8 / (n * SUBSTEPS))))  # ~2e8 env-steps per size
            a = argparse.Namespace(**{**vars(args), "steps": steps, "warmup": min(args.warmup, 20)})
            out = run(a, rank, local_rank, world, dist)
            torch.cuda.empty_cache()
            if out is not None:
                r = out["roofline"]
                rows.append({"envs_per_gpu": n, "steps": steps, "env_steps_per_s": out["value"], "kernel_ms": r["kernel_ms"],
                             "frac_algorithmic": r["frac_algorithmic"], "frac_measured": r["frac_measured"],
                             "traffic": r["traffic"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                             "mean_contacts_per_env": out["config"]["mean_contacts_per_env"], "valid_state": out["valid_state"],
                             "kernel": r["kernel"]})
                print(json.dumps(rows[-1]), flush=True)
        if rank == 0:
            os.makedirs(os.path.dirname(args.sweep_out), exist_ok=True)
            json.dump({"workload": args.workload, "build_id": build_id(), "rows": rows}, open(args.sweep_out, "w"), indent=1)
    else:
        out = run(args, rank, local_rank, world, dist)
        if out is not None:
            if dry:
                out["dry_run"] = True
            if not args.no_cpu_baseline and world == 1 and args.workload == "quadruped":
                out["cpu_baseline"] = cpu_baseline()
            print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
