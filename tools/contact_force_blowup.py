#!/usr/bin/env python
"""MEASUREMENT TOOL (run on the GPU box): the contact-force scene of tests/test_contact_force.py with the reference's exactly touching
pyramid cubes, stepped until a body exceeds 5 m/s; the state history and the contacts of that step go to gpurun_out/cf_blowup.npz --
the input of tests/golden/make_touching_cubes_vectors.py (round 6: frame 110)."""
import sys, numpy as np
sys.path.insert(0, "tests")
import newton_amd as nt
I4=[0,0,0,1.0]; h=0.5
sphere_r, heavy_r = 0.25, 0.5
b = nt.ModelBuilder()
ground = b.add_ground_plane()
b.default_shape_cfg.density = 1000.0
sphere = b.add_body(xform=[0.0, 0.0, sphere_r, *I4]); b.add_shape_sphere(sphere, radius=sphere_r)
b.default_shape_cfg.density = 2000.0
heavy = b.add_body(xform=[10.0, 0.0, heavy_r, *I4]); b.add_shape_sphere(heavy, radius=heavy_r)
b.default_shape_cfg.density = 1000.0
box = b.add_body(xform=[20.0, 0.0, h, *I4]); b.add_shape_box(box, hx=h, hy=h, hz=h)
left = b.add_body(xform=[30.0 - h, 0.0, h, *I4]); b.add_shape_box(left, hx=h, hy=h, hz=h)
right = b.add_body(xform=[30.0 + h, 0.0, h, *I4]); b.add_shape_box(right, hx=h, hy=h, hz=h)
top = b.add_body(xform=[30.0, 0.0, 3.0 * h, *I4]); b.add_shape_box(top, hx=h, hy=h, hz=h)
b.request_contact_attributes("force")
model = b.finalize(device="cuda:0")
shape_body = np.asarray(model.shape_body)
solver = nt.solvers.SolverXPBD(model, iterations=32, rigid_contact_con_weighting=True)
pipe = nt.CollisionPipeline(model); ct = pipe.contacts()
s0, s1 = model.state(), model.state()
dt = 1.0/60.0/8
hist = []
import os
for fr in range(130):
    for k in range(8):
        if fr >= 95:
            hist.append((s0.body_q.cpu().numpy().copy(), s0.body_qd.cpu().numpy().copy()))
        s0.clear_forces(); pipe.collide(s0, ct); solver.step(s0, s1, None, ct, dt); s0, s1 = s1, s0
        if fr >= 95 and float(np.abs(s0.body_qd.cpu().numpy()).max()) > 5.0 and not os.path.exists("gpurun_out/cf_blowup.npz"):
            n = int(ct.rigid_contact_count.cpu().numpy()[0])
            ex = {k_: getattr(ct, "rigid_contact_" + k_).cpu().numpy()[:n] for k_ in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")}
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/cf_blowup.npz", q=np.stack([h_[0] for h_ in hist]), qd=np.stack([h_[1] for h_ in hist]), q_after=s0.body_q.cpu().numpy(), qd_after=s0.body_qd.cpu().numpy(), **ex)
            print("BLOWUP at frame", fr, "substep", k, "steps recorded", len(hist))
    if fr % 20 == 0 or fr > 255:
        solver.update_contacts(ct, s0)
        n = int(ct.rigid_contact_count.cpu().numpy()[0])
        a, c, f = ct.rigid_contact_shape0.cpu().numpy()[:n], ct.rigid_contact_shape1.cpu().numpy()[:n], ct.force.cpu().numpy()[:n]
        tot = {}
        for i in range(n):
            key = (int(a[i]), int(c[i])); tot[key] = tot.get(key, 0.0) + float(f[i, 2])
        q = s0.body_q.cpu().numpy()
        print(fr, {k: round(v) for k, v in tot.items()}, "z:", np.round(q[:, 2], 4).tolist(), "x:", np.round(q[3:, 0], 4).tolist())
