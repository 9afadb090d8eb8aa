#!/usr/bin/env python
"""Find the first environment / frame of the hull_bin workload whose state stops being finite or explodes on the device and save
its state one frame earlier (gpurun_out/hull_nan.npz) so that the oracle can replay it on the CPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import newton_amd as nt
from scenes import hull_bin_scene

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
model = hull_bin_scene(E, 64, device="cuda:0")
t = model.env
pipe = nt.CollisionPipeline(model)
contacts = pipe.contacts()
solver = nt.solvers.SolverXPBD(model, iterations=2)
s0, s1 = model.state(), model.state()
dt = 1.0 / 600.0
prev = (s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy())
for f in range(frames):
    out = solver.rollout(s0, s1, None, contacts, dt, 10)
    torch.cuda.synchronize()
    q, qd = out.body_q.cpu().numpy(), out.body_qd.cpu().numpy()
    bad = ~np.isfinite(q).all(axis=1) | ~np.isfinite(qd).all(axis=1) | (np.abs(qd[:, :3]).max(axis=1) > 50.0)
    speed = np.nanmax(np.abs(qd[:, :3]))
    print(f"frame {f}: max lin speed {speed:.2f} max ang {np.nanmax(np.abs(qd[:, 3:])):.1f} bad bodies {int(bad.sum())}", flush=True)
    if bad.any():
        env = int(np.flatnonzero(bad)[0] // t.nb)
        sl = slice(env * t.nb, (env + 1) * t.nb)
        np.savez(os.path.join(ROOT, "gpurun_out", "hull_nan.npz"), env=env, frame=f, q_prev=prev[0][sl], qd_prev=prev[1][sl],
                 q_bad=q[sl], qd_bad=qd[sl])
        print("saved env", env, "frame", f)
        break
    prev = (q, qd)
