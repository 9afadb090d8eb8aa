"""Stability sweep of the C5 stepped contact model (MeshSdfContactStage + SolverSemiImplicit, explicit penalty contacts) on the
MI355X: for each (ke, kd, kf, dt, inertia armature) variant, step 4 x 12 hulls for `--seconds` and print one JSON line with
the first substep at which any speed exceeds 50 m/s (None = never), the final height range, speeds, live rows and the deepest
remaining separation.  Explicit damping is stable only while kd * n_contacts * dt / m (and kd * r^2 * n * dt / I) stay below 2;
the lightest hull of the scene weighs 25 g with a smallest principal inertia of 2.8e-6 kg m^2."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(E, H, ke, kd, kf, dt, armature, seconds):
    import torch
    from scenes import hull_bin_scene

    import newton_amd as nt
    from newton_amd.sdf_device import MeshSdfContactStage

    cfg = dict(ke=ke, kd=kd, kf=kf, mu=0.5, gap=0.004)
    model = hull_bin_scene(E, H, device="cuda:0", seed=2, hull_pairs=False, shape_cfg=cfg)
    if armature > 0.0:
        model.body_inertia = (model.body_inertia + np.eye(3, dtype=np.float32) * armature).astype(np.float32)
        model.body_inv_inertia = np.linalg.inv(model.body_inertia.astype(np.float64)).astype(np.float32)
    stage = MeshSdfContactStage(model, sdf_resolution=24)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverSemiImplicit(model)
    s0, s1, ctrl = model.state(), model.state(), model.control()
    n_steps = int(round(seconds / dt))
    blew, seen, max_rows_per_body = None, 0, 0
    for k in range(n_steps):
        s0.clear_forces()
        stage.collide(s0)
        stage.apply_forces(s0)
        pipe.collide(s0, contacts)
        solver.step(s0, s1, ctrl, contacts, dt)
        s0, s1 = s1, s0
        if k % 200 == 199:
            v = float(s0.body_qd.abs().max().item())
            n = int(stage.row_count.item())
            seen = max(seen, n)
            if n:
                a, b = stage.rigid_contact_shapes()
                live = a >= 0
                cnt = torch.bincount(torch.cat([a[live], b[live]]).long(), minlength=E * H)
                max_rows_per_body = max(max_rows_per_body, int(cnt.max().item()))
            if blew is None and (not np.isfinite(v) or v > 50.0):
                blew = k + 1
                break
    torch.cuda.synchronize()
    q, qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
    n = int(stage.row_count.item())
    sep = stage._row_data[:n, 6].cpu().numpy() if n else np.zeros(0)
    return {"ke": ke, "kd": kd, "kf": kf, "dt": dt, "armature": armature, "steps": n_steps, "blew_up_at": blew,
            "finite": bool(np.all(np.isfinite(q))), "z": [float(np.nanmin(q[:, 2])), float(np.nanmax(q[:, 2]))],
            "xy_max": float(np.nanmax(np.abs(q[:, :2]))), "v_max": float(np.nanmax(np.abs(qd[:, :3]))),
            "w_max": float(np.nanmax(np.abs(qd[:, 3:]))), "rows_seen": seen, "rows_last": n,
            "max_rows_per_body": max_rows_per_body, "sep_min": float(sep.min()) if len(sep) else None}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--envs", type=int, default=4)
    ap.add_argument("--hulls", type=int, default=12)
    args = ap.parse_args()
    variants = [  # ke, kd, kf, dt, armature -- second sweep: rotational part fixed by the armature, now the linear friction
        # gain (kf * n_contacts * dt / m < 2: kf = 200 allows ONE contact on the 25 g hull at dt = 1/4000)
        (2.0e3, 20.0, 200.0, 1 / 4000, 1.0e-4),
        (2.0e3, 5.0, 50.0, 1 / 4000, 1.0e-4),
        (2.0e3, 5.0, 20.0, 1 / 4000, 1.0e-4),
        (2.0e3, 2.0, 20.0, 1 / 4000, 1.0e-4),
        (2.0e3, 10.0, 20.0, 1 / 4000, 1.0e-4),
        (2.0e3, 5.0, 10.0, 1 / 4000, 1.0e-4),
        (1.0e3, 5.0, 20.0, 1 / 4000, 1.0e-4),
    ]
    for v in variants:
        print(json.dumps(run(args.envs, args.hulls, *v, args.seconds)), flush=True)
