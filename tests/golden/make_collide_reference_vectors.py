#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/collide_reference_vectors.npz by EXECUTING the reference's own collision
kernels in this container (see tests/golden/refshim): compute_shape_aabbs (newton/_src/sim/collide.py:283-472), the analytic
primitive narrow phase (create_narrow_phase_primitive_kernel, newton/_src/geometry/narrow_phase.py:403-1014), the GJK / MPR +
manifold narrow phase (create_narrow_phase_kernel_gjk_mpr, :1017-1219) and the pipeline's contact writer (write_contact,
collide.py:166-254), on the candidate pairs of the in-repo checker's broad phase.  tests/test_reference_vectors.py compares
the checker's collide() (AABBs, contact ids, body-frame points, offsets, normals, margins, append order) with the record.
Run from the repo root:  python tests/golden/make_collide_reference_vectors.py            (collide_reference_vectors.npz)
                         python tests/golden/make_collide_reference_vectors.py --barrel   (collide_barrel_reference_vectors.npz:
                         the barrel-cylinder scenes of collide_cases.barrel_cases())"""
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

narrow = importlib.import_module("newton._src.geometry.narrow_phase")
collide = importlib.import_module("newton._src.sim.collide")
PRIMITIVE = narrow.create_narrow_phase_primitive_kernel(collide.write_contact)
GJK_MPR = narrow.create_narrow_phase_kernel_gjk_mpr(True, collide.write_contact)


def arr(a, dtype):
    return wp.to_array(np.asarray(a), dtype)


def reference_collide(model, body_q, pairs, cmax):
    """-> dict of flat contact arrays + AABBs, produced by the reference kernels."""
    S = len(model.shape_type)
    m = model
    bq = arr(body_q, wp.transform)
    lo, hi = wp.zeros(S, dtype=wp.vec3), wp.zeros(S, dtype=wp.vec3)
    geom_data, geom_xform = wp.zeros(S, dtype=wp.vec4), wp.zeros(S, dtype=wp.transform)
    zero3 = arr(np.zeros((S, 3), np.float32), wp.vec3)
    shape_body, shape_type = arr(m.shape_body, int), arr(m.shape_type, int)
    shape_gap, shape_margin = arr(m.shape_gap, float), arr(m.shape_margin, float)
    # convex hulls: a Mesh of the unscaled vertices per shape, its id in shape_source_ptr, scaled local AABB (builder.py:11533-11686)
    ids, loc_lo, loc_hi = np.zeros(S, np.int64), np.zeros((S, 3), np.float32), np.zeros((S, 3), np.float32)
    ms, mc = np.asarray(getattr(m, "shape_mesh_start", -np.ones(S)), int), np.asarray(getattr(m, "shape_mesh_count", np.zeros(S)), int)
    for k in range(S):
        if mc[k] > 0:
            v = np.asarray(m.mesh_points, np.float32).reshape(-1, 3)[ms[k]:ms[k] + mc[k]]
            ids[k] = wp.Mesh(points=arr(v, wp.vec3)).id
            sc = np.asarray(m.shape_scale, np.float32)[k]
            loc_lo[k], loc_hi[k] = v.min(axis=0) * sc, v.max(axis=0) * sc
    src = wp.to_array(ids, int)
    zero3_lo, zero3_hi = arr(loc_lo, wp.vec3), arr(loc_hi, wp.vec3)
    radius = arr(getattr(m, "shape_collision_radius", np.zeros(S, np.float32)), float)
    wp.launch(collide.compute_shape_aabbs, dim=S,
              inputs=[bq, arr(m.shape_transform, wp.transform), shape_body, shape_type, arr(m.shape_scale, wp.vec3), radius, src,
                      shape_margin, shape_gap, zero3_lo, zero3_hi, wp.zeros(1, dtype=int), wp.zeros(1, dtype=int), wp.zeros(1, dtype=int), 1],
              outputs=[lo, hi, geom_data, geom_xform])
    P = len(pairs)
    cand = arr(np.asarray(pairs, np.int32).reshape(-1, 2), wp.vec2i)
    count = arr(np.array([P]), int)
    w = collide.ContactWriterData()
    w.contact_max = cmax
    w.body_q, w.shape_body, w.shape_gap = bq, shape_body, shape_gap
    w.contact_count = wp.zeros(1, dtype=int)
    w.out_shape0, w.out_shape1 = wp.full(cmax, -1, dtype=int), wp.full(cmax, -1, dtype=int)
    for k in ("out_point0", "out_point1", "out_offset0", "out_offset1", "out_normal"):
        setattr(w, k, wp.zeros(cmax, dtype=wp.vec3))
    w.out_margin0, w.out_margin1, w.out_tids = wp.zeros(cmax, dtype=float), wp.zeros(cmax, dtype=float), wp.zeros(cmax, dtype=int)
    w.out_stiffness = w.out_damping = w.out_friction = wp.zeros(0, dtype=float)
    w.out_sort_key = wp.zeros(0, dtype=int)
    w.shape_transform, w.shape_linear_velocity, w.shape_angular_velocity = geom_xform, wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=wp.vec3)
    w.collision_update_dt = w.max_speculative_extension = wp.f32(0.0)
    gjk_pairs, gjk_count = wp.zeros(max(P, 1), dtype=wp.vec2i), wp.zeros(1, dtype=int)
    empty_pairs = lambda: wp.zeros(0, dtype=wp.vec2i)  # noqa: E731
    one = lambda: wp.zeros(1, dtype=int)  # noqa: E731
    wp.launch(PRIMITIVE, dim=max(P, 1),
              inputs=[cand, count, shape_type, geom_data, geom_xform, wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=wp.vec3), 0.0, 0.0,
                      src, shape_gap, arr(m.shape_flags, int), wp.full(S, -1, dtype=int), wp.zeros(S, dtype=wp.vec2i), w, max(P, 1)],
              outputs=[gjk_pairs, gjk_count, empty_pairs(), one(), empty_pairs(), wp.zeros(0, dtype=int), one(), one(), empty_pairs(),
                       one(), empty_pairs(), one()])
    n_analytic = int(w.contact_count[0])
    G = int(gjk_count[0])
    if G:
        wp.launch(GJK_MPR, dim=G,
                  inputs=[gjk_pairs, gjk_count, shape_type, geom_data, geom_xform, src, shape_gap, radius, lo, hi, zero3_lo, zero3_hi, w, G])
    n = int(w.contact_count[0])
    v3 = lambda a: np.array([[float(c) for c in x] for x in a[:n]], np.float32).reshape(n, 3)  # noqa: E731
    return {"count": np.array([n]), "count_analytic": np.array([n_analytic]), "gjk_pairs": np.array([[p[0], p[1]] for p in gjk_pairs[:G]], np.int32).reshape(G, 2),
            "shape0": np.array(w.out_shape0[:n], np.int32), "shape1": np.array(w.out_shape1[:n], np.int32),
            "point0": v3(w.out_point0), "point1": v3(w.out_point1), "offset0": v3(w.out_offset0), "offset1": v3(w.out_offset1),
            "normal": v3(w.out_normal), "margin0": np.array(w.out_margin0[:n], np.float32), "margin1": np.array(w.out_margin1[:n], np.float32),
            "aabb_lower": np.array([[float(c) for c in x] for x in lo], np.float32), "aabb_upper": np.array([[float(c) for c in x] for x in hi], np.float32)}


def main():
    import collide_cases as cc
    import oracle_bridge as ob

    blob = {}
    barrel = "--barrel" in sys.argv  # the barrel-cylinder scenes live in their own record (collide_barrel_reference_vectors.npz)
    for name, make in (cc.barrel_cases() if barrel else cc.cases()).items():
        model, body_q = make()
        orc = ob.Oracle(model)
        ct = orc.contacts()
        pairs, lo, hi = orc.collide(body_q, ct)
        res = reference_collide(model, body_q, pairs, ct.max)
        print(name, "pairs", len(pairs), "reference contacts", int(res["count"][0]), "(analytic", int(res["count_analytic"][0]), ") checker contacts",
              int(ct.count[0]), flush=True)
        blob[f"{name}/pairs"] = np.asarray(pairs, np.int32)
        blob[f"{name}/body_q"] = np.asarray(body_q, np.float32)
        for k, v in res.items():
            blob[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "collide_barrel_reference_vectors.npz" if barrel else "collide_reference_vectors.npz"), **blob)
    print("wrote", len(blob), "arrays")


if __name__ == "__main__":
    main()
