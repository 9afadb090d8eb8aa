#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/flat_contact_reference_vectors.npz by EXECUTING, in this container (see
tests/golden/refshim), the two reference functions the flat contact stage stands in for:
  * write_contact (newton/_src/sim/collide.py:203-254): ContactData rows -> the Contacts arrays (body-frame points, offsets,
    normal, margins), rows beyond the gap rejected;
  * eval_body_contact (newton/_src/solvers/semi_implicit/kernels_contact.py:381-556): penalty force of every contact row,
    accumulated into body_f -- with and without per-contact stiffness / damping / friction.
Run from the repo root:  python tests/golden/make_flat_contact_reference_vectors.py"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "refshim"))
import lazy_ref  # noqa: E402

lazy_ref.install(dummies={"newton._src.sim": ("Contacts", "Control", "Model", "State", "ModelBuilder")},
                 dummy_modules=("newton._src.geometry.sdf_hydroelastic", "newton._src.sim.builder", "newton._src.geometry.sdf_contact",
                                "newton._src.geometry.sdf_utils", "newton._src.geometry.sdf_texture"))
import warp as wp  # noqa: E402  (the stand-in)

collide = importlib.import_module("newton._src.sim.collide")
kc = importlib.import_module("newton._src.solvers.semi_implicit.kernels_contact")
cd = importlib.import_module("newton._src.geometry.contact_data")


@wp.kernel
def write_rows(shape_a, shape_b, center, normal, distance, margin_a, margin_b, key, writer_data):
    i = wp.tid()
    c = cd.ContactData()
    c.contact_point_center = center[i]
    c.contact_normal_a_to_b = normal[i]
    c.contact_distance = distance[i]
    c.radius_eff_a = 0.0
    c.radius_eff_b = 0.0
    c.margin_a = margin_a[i]
    c.margin_b = margin_b[i]
    c.shape_a = shape_a[i]
    c.shape_b = shape_b[i]
    c.sort_sub_key = key[i]
    collide.write_contact(c, writer_data, -1)


def run(case):
    A = wp.to_array
    r = case["rows"]
    n = len(r["key"])
    w = collide.ContactWriterData()
    w.contact_max = n
    w.body_q, w.shape_body, w.shape_gap = A(case["body_q"], wp.transform), A(case["shape_body"], int), A(case["shape_gap"], float)
    w.contact_count = wp.zeros(1, dtype=int)
    w.out_shape0, w.out_shape1 = wp.full(n, -1, dtype=int), wp.full(n, -1, dtype=int)
    for k in ("out_point0", "out_point1", "out_offset0", "out_offset1", "out_normal"):
        setattr(w, k, wp.zeros(n, dtype=wp.vec3))
    w.out_margin0, w.out_margin1, w.out_tids = wp.zeros(n, dtype=float), wp.zeros(n, dtype=float), wp.zeros(n, dtype=int)
    w.out_stiffness = w.out_damping = w.out_friction = wp.zeros(0, dtype=float)
    w.out_sort_key = wp.zeros(0, dtype=int)
    w.shape_transform, w.shape_linear_velocity, w.shape_angular_velocity = wp.zeros(0, dtype=wp.transform), wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=wp.vec3)
    w.collision_update_dt = w.max_speculative_extension = wp.f32(0.0)
    wp.launch(write_rows, dim=n, inputs=[A(r["shape_a"], int), A(r["shape_b"], int), A(r["center"], wp.vec3), A(r["normal"], wp.vec3),
                                         A(r["distance"], float), A(r["margin_a"], float), A(r["margin_b"], float), A(r["key"], int), w])
    m = int(w.contact_count[0])
    v3 = lambda a: np.array([[float(c) for c in x] for x in a[:m]], np.float32).reshape(m, 3)  # noqa: E731
    out = {"count": np.array([m]), "shape0": np.array(w.out_shape0[:m], np.int32), "shape1": np.array(w.out_shape1[:m], np.int32),
           "point0": v3(w.out_point0), "point1": v3(w.out_point1), "offset0": v3(w.out_offset0), "offset1": v3(w.out_offset1),
           "normal": v3(w.out_normal), "margin0": np.array(w.out_margin0[:m], np.float32), "margin1": np.array(w.out_margin1[:m], np.float32)}
    # rows are written in ascending tid order (the stand-in's launch order), so accepted row k is the k-th row that passed
    B = len(case["body_q"])
    mat = case["mat"]
    props = case["props"]
    if props is not None:  # per-contact properties follow the accepted rows
        d = np.asarray(r["distance"], np.float32)
        # which input rows were accepted: replay the writer's count by matching normals / shapes is fragile; instead rerun the
        # writer row by row
        accepted = []
        for i in range(n):
            before = int(w.contact_count[0])
            w.contact_count[0] = 0
            wp.launch(write_rows, dim=1, inputs=[A(r["shape_a"][i:i + 1], int), A(r["shape_b"][i:i + 1], int), A(r["center"][i:i + 1], wp.vec3),
                                                 A(r["normal"][i:i + 1], wp.vec3), A(r["distance"][i:i + 1], float),
                                                 A(r["margin_a"][i:i + 1], float), A(r["margin_b"][i:i + 1], float),
                                                 A(r["key"][i:i + 1], int), w])
            if int(w.contact_count[0]) == 1:
                accepted.append(i)
            w.contact_count[0] = before
        accepted = np.array(accepted)
        assert len(accepted) == m
        stiff, damp, fric = (A(props[k][accepted], float) for k in ("stiffness", "damping", "friction"))
        out["accepted"] = accepted.astype(np.int32)
    else:
        stiff = damp = fric = None
    body_f = wp.zeros(B, dtype=wp.spatial_vector)
    wp.launch(kc.eval_body_contact, dim=max(m, 1),
              inputs=[A(case["body_q"], wp.transform), A(case["body_qd"], wp.spatial_vector), A(case["body_com"], wp.vec3),
                      A(mat["ke"], float), A(mat["kd"], float), A(mat["kf"], float), A(mat["ka"], float), A(mat["mu"], float),
                      A(case["shape_body"], int), wp.to_array(np.array([m]), int), A(out["point0"], wp.vec3), A(out["point1"], wp.vec3),
                      A(out["normal"], wp.vec3), A(out["shape0"], int), A(out["shape1"], int), A(out["margin0"], float),
                      A(out["margin1"], float), stiff, damp, fric, False, float(case["friction_smoothing"])],
              outputs=[body_f])
    out["body_f"] = np.array([[float(c) for c in x] for x in body_f], np.float32).reshape(B, 6)
    return out


def main():
    import flat_contact_cases as fc

    rec = {}
    for name in fc.CASES:
        out = run(fc.make(name))
        print(name, "rows", len(fc.make(name)["rows"]["key"]), "accepted", int(out["count"][0]), "max |f|", float(np.abs(out["body_f"]).max()))
        for k, v in out.items():
            rec[f"{name}/{k}"] = v
    path = os.path.join(HERE, "flat_contact_reference_vectors.npz")
    np.savez_compressed(path, **rec)
    print("wrote", len(rec), "arrays to", path)


if __name__ == "__main__":
    main()
