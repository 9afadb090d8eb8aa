#!/usr/bin/env python
"""The reference's basic URDF example (newton/examples/basic/example_basic_urdf.py) on newton_amd: N quadrupeds, XPBD, 10
substeps per frame -- the caller loop is the reference's, the fused `rollout` is the CUDA-graph replacement.

    python examples/quadruped_xpbd.py [--worlds 100] [--frames 200] [--fused]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import newton_amd as newton  # noqa: E402
from scenes import quadruped_scene  # noqa: E402  (builder -> URDF -> replicate -> ground plane, like the reference example)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=100)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--fused", action="store_true", help="one launch per frame instead of 3 x substeps launches")
    args = ap.parse_args()

    model = quadruped_scene(args.worlds, device="cuda:0")
    solver = newton.solvers.SolverXPBD(model, iterations=2)
    state_0, state_1, control = model.state(), model.state(), model.control()
    pipeline = newton.CollisionPipeline(model)
    contacts = pipeline.contacts()
    fps, substeps = 100, 10
    dt = 1.0 / fps / substeps

    for _ in range(args.frames):
        if args.fused:
            out = solver.rollout(state_0, state_1, control, contacts, dt, substeps)
            if out is state_1:
                state_0, state_1 = state_1, state_0
        else:
            for _ in range(substeps):  # the reference's simulate()
                state_0.clear_forces()
                pipeline.collide(state_0, contacts)
                solver.step(state_0, state_1, control, contacts, dt)
                state_0, state_1 = state_1, state_0

    # the example's own acceptance test (example_basic_urdf.py:145-162): robots rest on their feet
    q, qd = state_0.body_q, state_0.body_qd
    root_z = q.reshape(args.worlds, -1, 7)[:, 0, 2]
    print(f"root height {float(root_z.mean()):.3f} m (expected 0.46 +/- 0.01), max |qd| {float(qd.abs().max()):.3f}")


if __name__ == "__main__":
    main()
