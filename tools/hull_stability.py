#!/usr/bin/env python
"""How many environments of the hull-bin workload leave the finite / sane range within N frames, per solver setting (GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import newton_amd as nt
from scenes import hull_bin_scene

E, frames = 1024, 60
CONFIGS = [
    ("base dt600 it2", dict(), dict(iterations=2), 1 / 600.0),
    ("mu0.3", dict(mu=0.3), dict(iterations=2), 1 / 600.0),
    ("mu0.1", dict(mu=0.1), dict(iterations=2), 1 / 600.0),
    ("relax0.4", dict(), dict(iterations=2, rigid_contact_relaxation=0.4), 1 / 600.0),
    ("it4 relax0.4", dict(), dict(iterations=4, rigid_contact_relaxation=0.4), 1 / 600.0),
    ("no weighting off", dict(), dict(iterations=2, rigid_contact_con_weighting=False), 1 / 600.0),
    ("dt1200", dict(), dict(iterations=2), 1 / 1200.0),
    ("damp1", dict(), dict(iterations=2, angular_damping=1.0), 1 / 600.0),
]
for name, skw, kw, dt in CONFIGS:
    model = hull_bin_scene(E, 64, device="cuda:0", **skw)
    t = model.env
    contacts = nt.CollisionPipeline(model).contacts()
    solver = nt.solvers.SolverXPBD(model, **kw)
    s0, s1 = model.state(), model.state()
    worst = 0.0
    for f in range(frames):
        out = solver.rollout(s0, s1, None, contacts, dt, 10)
    torch.cuda.synchronize()
    qd = out.body_qd.cpu().numpy().reshape(E, t.nb, 6)
    bad = ~np.isfinite(qd).all(axis=(1, 2)) | (np.abs(np.nan_to_num(qd[:, :, 3:], nan=1e30)).max(axis=1).max(axis=1) > 1000.0)
    fin = np.abs(qd[np.isfinite(qd).all(axis=(1, 2))][:, :, 3:])
    print(f"{name:22s} bad envs {int(bad.sum()):4d} / {E}   median max-ang {np.median(fin.max(axis=(1, 2))):8.2f}  p99 {np.percentile(fin.max(axis=(1, 2)), 99):10.1f}", flush=True)
