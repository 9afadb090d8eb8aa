#!/usr/bin/env python
"""Config C5 with its real contact model, collide stage only: E environments x 64 random convex hulls (meshes with sparse
texture SDFs, uint16, narrow band +-0.1 -- SURVEY.md section 8(d) C5) resting in the bin, through the reference's pipeline
shape for mesh-mesh pairs (collide.py:1925-2050 -> narrow_phase.py:2588-2760):

    world AABBs of the hulls  ->  per-world sort-and-sweep on the device (nt_broadphase_sap_device)
                              ->  mesh-vs-SDF edge contacts + global contact reduction (nt_mesh_sdf_collide_reduced)

The poses come from settling the same scene with the convex (MPR / GJK) XPBD path first.  Everything stays on the device; the
timed region is K passes of the three launches with HIP events.  Prints one JSON line (pairs / s, contacts, per-stage ms).
Usage: python tools/sdf_bin_bench.py --envs 2048 [--steps 20] [--sdf-resolution 24] [--out profiles/....json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def step_mode(args):
    """C5 with its SDF contact model, stepped: env-steps/s of {clear_forces, SDF stage collide + forces, tile collide, step}."""
    import torch

    import newton_amd as nt
    import scenes
    from newton_amd import _lib
    from newton_amd.sdf_device import MeshSdfContactStage

    E, H = args.envs, args.hulls
    cfg = dict(ke=2.0e3, kd=10.0, kf=20.0, mu=0.5, gap=args.gap)  # kf * n_contacts * dt / m < 2 on a 25 g hull with ~10 rows
    # inertia armature: the explicit penalty contacts are unstable on the bare 25 g hulls (scenes.hull_bin_scene docstring)
    model = scenes.hull_bin_scene(E, H, device="cuda:0", seed=2, hull_pairs=False, shape_cfg=cfg, inertia_armature=1.0e-4)
    t0 = time.perf_counter()
    stage = MeshSdfContactStage(model, sdf_resolution=args.sdf_resolution, threads=args.threads)
    t_sdf = time.perf_counter() - t0
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverSemiImplicit(model)
    s0, s1, ctrl = model.state(), model.state(), model.control()
    dt = 1.0 / 4000.0

    def substep():
        nonlocal s0, s1
        s0.clear_forces()
        stage.collide(s0)
        stage.apply_forces(s0)
        pipe.collide(s0, contacts)
        solver.step(s0, s1, ctrl, contacts, dt)
        s0, s1 = s1, s0

    for _ in range(args.settle_frames * 10):
        substep()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_sub = args.steps * 10
    ev0.record()
    for _ in range(n_sub):
        substep()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / n_sub
    q = s0.body_q
    out = {"workload": f"C5 stepped with its SDF contact model: {E} envs x {H} hulls (mesh + uint16 texture SDF, resolution "
                       f"{args.sdf_resolution}), per substep clear_forces + [AABBs, SAP, mesh-SDF edge contacts + global reduction, "
                       "write_contact rows, eval_body_contact into body_f] + tile collide (hull-wall pairs) + SolverSemiImplicit.step, "
                       f"dt = {dt:g}", "envs": E, "hulls_per_env": H, "ms_per_substep": ms, "env_steps_per_s": E / (ms * 1e-3),
           "sdf_rows": int(stage.row_count.item()), "candidate_pairs": int(stage.pair_count.item()),
           "finite": bool(torch.isfinite(q).all().item()),
           "valid_state": bool(torch.isfinite(q).all().item() and q[:, 2].min().item() > 0.0 and q[:, 2].max().item() < 1.0
                               and s0.body_qd[:, :3].abs().max().item() < 5.0),
           "v_max": float(s0.body_qd[:, :3].abs().max().item()), "inertia_armature": 1.0e-4, "contact_cfg": cfg, "z_min": float(q[:, 2].min().item()), "z_max": float(q[:, 2].max().item()),
           "host_s": {"sdf_build": t_sdf}, "threads_per_pair": args.threads, "build_id": _lib.load().nt_build_info().decode()}
    line = json.dumps(out)
    print(line)
    if args.out:
        open(args.out, "w").write(line + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--hulls", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-frames", type=int, default=60)
    ap.add_argument("--sdf-resolution", type=int, default=24)
    ap.add_argument("--gap", type=float, default=0.005)
    ap.add_argument("--unreduced", action="store_true", help="also time the unreduced kernel")
    ap.add_argument("--threads", type=int, default=64, help="workgroup size per pair of the reduced kernel (64 / 128 / 256)")
    ap.add_argument("--step", action="store_true", help="instead of the collide-only line: step the scene with SolverSemiImplicit, "
                    "hull-hull contacts from MeshSdfContactStage (penalty forces into body_f), hull-wall contacts from the tiles")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.step:
        return step_mode(args)

    import torch

    import newton_amd as nt
    import scenes
    from newton_amd import geometry
    from newton_amd import sdf as S
    from newton_amd.mesh import mesh_edge_tables
    from newton_amd.sdf_device import DeviceSDF, MeshSdfNarrowPhase

    dev = "cuda:0"
    E, H = args.envs, args.hulls
    t0 = time.perf_counter()
    model = scenes.hull_bin_scene(E, H, device=dev, seed=2)
    # -- settle with the convex path (the same scene bench.py --workload hull_bin steps)
    s0, s1, ctrl = model.state(), model.state(), model.control()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    for _ in range(args.settle_frames):
        solver.rollout(s0, s1, ctrl, contacts, 1.0 / 1200.0, 10)
    torch.cuda.synchronize()
    body_q = s0.body_q.reshape(E, H, 7).contiguous()  # hull k of env e is body e * H + k; its shape sits at the body origin
    assert torch.isfinite(body_q).all()
    t_settle = time.perf_counter() - t0

    # -- the 64 hull meshes (shared by every environment): edges, SDFs, reduction tables
    t0 = time.perf_counter()
    meshes = [model.shape_source[k] for k in range(H)]  # every environment shares the hull set (shapes 0 .. H-1 of env 0)
    assert all(m is not None and len(m.vertices) >= 4 for m in meshes)
    ecs, ehs, er, sdfs = [], [], [], []
    e0 = 0
    for m in meshes:
        ec, eh = mesh_edge_tables(m.vertices, m.indices.reshape(-1, 3))
        ecs.append(ec)
        ehs.append(eh)
        er.append((e0, len(ec)))
        e0 += len(ec)
        sdfs.append(S.create_texture_sdf_from_mesh(m.vertices, m.indices.reshape(-1, 3), margin=0.02, narrow_band_range=(-0.1, 0.1),
                                                   max_resolution=args.sdf_resolution, quantization_mode=S.QuantizationMode.UINT16))
    lo, hi, res = S.mesh_reduction_tables([m.vertices for m in meshes], [(1, 1, 1)] * H)
    t_sdf = time.perf_counter() - t0
    n = E * H
    tile = lambda a: np.tile(np.asarray(a), (E,) + (1,) * (np.asarray(a).ndim - 1))  # noqa: E731
    narrow = MeshSdfNarrowPhase(shape_data=tile(np.array([[1, 1, 1, 0.0]] * H, np.float32)),
                                shape_gap=np.full(n, args.gap, np.float32), shape_sdf_index=tile(np.arange(H, dtype=np.int32)),
                                sdfs=[DeviceSDF(t, device=dev) for t in sdfs], shape_edge_range=tile(np.array(er, np.int32)),
                                edge_centers=np.concatenate(ecs), edge_halves=np.concatenate(ehs),
                                reduce_tables=(tile(lo), tile(hi), tile(res)), device=dev)
    # -- world AABBs: rotate the 8 corners of every hull's local box (conservative, like a box proxy)
    corners = np.array([[(lo[k][0], hi[k][0])[i], (lo[k][1], hi[k][1])[j], (lo[k][2], hi[k][2])[l]]
                        for k in range(H) for i in (0, 1) for j in (0, 1) for l in (0, 1)], np.float32).reshape(H, 8, 3)
    t_corners = torch.from_numpy(corners).to(dev)
    shape_world = torch.arange(E, dtype=torch.int32, device=dev).repeat_interleave(H)
    group = torch.ones(n, dtype=torch.int32, device=dev)
    gap = torch.full((n,), args.gap, dtype=torch.float32, device=dev)
    bp = geometry.BroadPhaseSAP(shape_world.cpu().numpy(), None, device=dev)
    pair_cap = E * H * 12
    pairs = torch.zeros((pair_cap, 2), dtype=torch.int32, device=dev)
    pair_count = torch.zeros(1, dtype=torch.int32, device=dev)
    cap = E * H * 40
    o_count = torch.zeros(1, dtype=torch.int32, device=dev)
    o_pair, o_key = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.int32, device=dev)
    o_data = torch.zeros((cap, 9), dtype=torch.float32, device=dev)

    def aabbs():
        p, q = body_q[..., :3], body_q[..., 3:]
        qv, w = q[..., None, :3], q[..., None, 3:]
        qv, c = torch.broadcast_tensors(qv, t_corners[None])  # [E, H, 8, 3]
        rot = c * (2.0 * w * w - 1.0) + torch.cross(qv, c, dim=-1) * w * 2.0 + qv * (qv * c).sum(-1, keepdim=True) * 2.0
        world = rot + p[..., None, :]
        return world.amin(dim=2).reshape(n, 3).contiguous(), world.amax(dim=2).reshape(n, 3).contiguous()

    X = body_q.reshape(n, 7)

    def one_pass(reduce=True):
        lower, upper = aabbs()
        bp.launch(lower, upper, gap, group, shape_world, n, pairs, pair_count)
        o_count.zero_()
        narrow.launch(X, pairs, pair_count, o_count, o_pair, o_key, o_data, reduce=reduce, threads=args.threads)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def timed(reduce):
        for _ in range(args.warmup):
            one_pass(reduce)
        torch.cuda.synchronize()
        tb = tn = ta = 0.0
        for _ in range(args.steps):
            ev[0].record()
            lower, upper = aabbs()
            ev[1].record()
            bp.launch(lower, upper, gap, group, shape_world, n, pairs, pair_count)
            ev[2].record()
            o_count.zero_()
            narrow.launch(X, pairs, pair_count, o_count, o_pair, o_key, o_data, reduce=reduce, threads=args.threads)
            ev[3].record()
            torch.cuda.synchronize()
            ta += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
            tn += ev[2].elapsed_time(ev[3])
        return ta / args.steps, tb / args.steps, tn / args.steps

    ms_aabb, ms_bp, ms_np = timed(True)
    n_pairs, n_contacts = int(pair_count.item()), int(o_count.item())
    assert n_pairs <= pair_cap and n_contacts <= cap, (n_pairs, n_contacts)
    touching = int(torch.unique(o_pair[:n_contacts]).numel())
    # determinism: a second pass gives the same rows (blocks land in another order; sort by (pair, key))
    def snapshot():
        k = o_pair[:n_contacts].to(torch.int64) * (1 << 32) + o_key[:n_contacts].to(torch.int64)
        o = torch.argsort(k)
        return pairs[o_pair[:n_contacts][o].long()].clone(), o_key[:n_contacts][o].clone(), o_data[:n_contacts][o].clone()
    a = snapshot()
    one_pass(True)
    torch.cuda.synchronize()
    b = snapshot()
    same = all(torch.equal(x, y) for x, y in zip(a, b))
    out = {"workload": f"C5 contact model, collide only: {E} envs x {H} convex hulls as meshes with uint16 texture SDFs "
                       f"(resolution {args.sdf_resolution}, narrow band +-0.1), SAP broad phase per world + mesh-SDF edge "
                       "contacts + global contact reduction, poses settled by the convex XPBD path",
           "envs": E, "hulls_per_env": H, "edges_per_hull_mean": float(np.mean([r[1] for r in er])),
           "candidate_pairs": n_pairs, "touching_pairs": touching, "reduced_contacts": n_contacts,
           "contacts_per_env": n_contacts / E, "ms_aabb": ms_aabb, "ms_broad_phase": ms_bp, "ms_narrow_phase_reduced": ms_np,
           "ms_collide": ms_aabb + ms_bp + ms_np, "candidate_pairs_per_s": n_pairs / (ms_np * 1e-3),
           "collides_per_s_envs": E / ((ms_aabb + ms_bp + ms_np) * 1e-3), "deterministic_rows": bool(same), "threads_per_pair": args.threads,
           "host_s": {"settle": t_settle, "sdf_build": t_sdf}}
    if args.unreduced:
        _, _, ms_un = timed(False)
        out["ms_narrow_phase_unreduced"] = ms_un
        out["unreduced_contacts"] = int(o_count.item())
    from newton_amd import _lib

    out["build_id"] = _lib.load().nt_build_info().decode()
    line = json.dumps(out)
    print(line)
    if args.out:
        open(args.out, "w").write(line + "\n")


if __name__ == "__main__":
    main()
