"""Diagnostic: the bench's terrain scene frame by frame -- which primitive kinds leave the surface, and when."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402

import newton_amd as nt  # noqa: E402

E, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 2
bp = sys.argv[3] if len(sys.argv) > 3 else "sap"
model = scenes.terrain_scene(E, 8, device="cuda:0", seed=6)
pipe = nt.CollisionPipeline(model, broad_phase=bp)
contacts = pipe.contacts()
solver = nt.solvers.SolverXPBD(model, iterations=iters)
s0, s1 = model.state(), model.state()
dt = 1.0 / 600.0
for frame in range(40):
    for _ in range(10):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, dt)
        s0, s1 = s1, s0
    if frame % 5 == 4 or frame < 3:
        q = s0.body_q.cpu().numpy().reshape(E, 8, 7)
        qd = s0.body_qd.cpu().numpy().reshape(E, 8, 6)
        h = q[:, :, 2] - scenes.terrain_height(q[:, :, 0], q[:, :, 1])
        f = contacts._flat
        n = int(f.row_start[-1].item())
        live = int((f.shape0[:n] >= 0).sum().item())
        print(f"frame {frame}: rows {n} live {live} overflow {pipe._sdf_leg.overflow(f)['overflow']}")
        for k, name in enumerate(["box", "sphere", "capsule", "cylinder"] * 2):
            print(f"   {k} {name:8s} h min {h[:, k].min():+.4f} max {h[:, k].max():+.4f}  |v| max {np.abs(qd[:, k, :3]).max():.3f}  |w| max {np.abs(qd[:, k, 3:]).max():.3f}")
