"""Mixed-world timing on one MI355X (secondary measurement, not the BASELINE metric): a model whose worlds differ in topology
(quadrupeds | 3-box stacks | quadrupeds | double pendulums) stepped through the unchanged Newton-shaped calls, which dispatch to
the world groups (newton_amd/hetero.py).  One frame = SolverXPBD.rollout of 10 substeps = one launch per group.  Prints the
frame time with the groups on sibling HIP streams and with all launches on the caller's stream."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))


def main():
    import torch
    from test_heterogeneous_worlds import mixed_model

    import newton_amd as nt
    from newton_amd import _lib

    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=200)
    args = ap.parse_args()
    k = args.scale
    layout = (("quadruped", 2 * k), ("boxes3", k), ("quadruped", k), ("pendulum", 2 * k))
    t0 = time.perf_counter()
    model = mixed_model(layout, device="cuda:0")
    t_build = time.perf_counter() - t0
    solver = nt.solvers.SolverXPBD(model)
    contacts = nt.CollisionPipeline(model).contacts()
    s0, s1, ctrl = model.state(), model.state(), model.control()
    W = model.world_count
    out = {"workload": f"mixed worlds: {2 * k} quadrupeds | {k} 3-box stacks | {k} quadrupeds | {2 * k} double pendulums, "
                       "SolverXPBD iterations=2, dt=0.001, 1 frame = 10 substeps fused per world group",
           "worlds": W, "groups": [[b, e] for b, e in model.world_groups.ranges], "host_build_s": t_build,
           "build_id": _lib.load().nt_build_info().decode()}
    for mode in ("streams", "serial"):
        model.world_groups.concurrent = mode == "streams"
        for _ in range(30):
            r = solver.rollout(s0, s1, ctrl, contacts, 1e-3, 10)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(args.frames):
            r = solver.rollout(s0, s1, ctrl, contacts, 1e-3, 10)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / args.frames
        out[f"ms_per_frame_{mode}"] = ms
        out[f"world_steps_per_s_{mode}"] = W * 10 / (ms * 1e-3)
    q = r.body_q
    out["finite"] = bool(torch.isfinite(q).all().item())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
