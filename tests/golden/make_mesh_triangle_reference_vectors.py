#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/mesh_triangle_reference_vectors.npz by EXECUTING, in this container (see
tests/golden/refshim), the reference's mesh-vs-convex leg on the scenes of tests/golden/mesh_triangle_cases.py:

  * `narrow_phase_find_mesh_triangle_overlaps_kernel` (newton/_src/geometry/narrow_phase.py:1455-1568 -> collision_core.py:996-1180
    mesh_vs_convex_midphase: the convex shape's support-function AABB in the unscaled mesh frame, widened by margin + gap, the mesh
    query, the front-face test) -- record "<case>/tri_pairs": (mesh, convex, triangle), sorted.  The BVH itself is Warp-native: the
    stand-in tests every triangle's float32 bounds against the query box (same set, index order);
  * `mesh_triangle_contacts_to_reducer_kernel` (contact_reduction_global.py:2299-2403: get_triangle_shape_from_mesh, back-face
    culling, GJK / MPR + manifold through create_compute_gjk_mpr_contacts with post_process_triangle_contact, written with
    write_contact_to_reducer) -- record "<case>/buffered_*": the UNREDUCED list sorted by (mesh, convex, fingerprint),
    fingerprint = (((triangle << 1) | 1) << 3) | manifold index;
  * `reduce_contact_in_hashtable` (:1246-1346, beta = 1e-4) in buffer order AND in reverse order, then
    `export_reduced_contacts_kernel` (:2133-2290) with a recording writer: the surviving contacts sorted by (shape a, shape b,
    fingerprint) (record "<case>/pair|fp|pos|normal|depth|misc"); both orders must agree.
Run from the repo root:  python tests/golden/make_mesh_triangle_reference_vectors.py"""
import importlib
import os
import sys
import warnings

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

# Warp bakes `wp.static(<python float>)` into the kernel as a float32 literal; the stand-in's static() hands the double through, so
# products of two such constants (multicontact.py:848-849 `c * SIN_TILT_ANGLE`: the tilted support directions of the manifold) would
# be rounded once from the double product instead of as a float32 product.  Box / hull / triangle supports are vertices and do not
# see the difference; the curved partners of this leg (cylinder, cone) do, by one ulp of the lateral support point.  This generator
# gives static() Warp's float32 before the reference modules are imported.
_static = wp.static
wp.static = lambda x: np.float32(x) if isinstance(x, float) else _static(x)

warnings.filterwarnings("ignore", category=RuntimeWarning)
g = importlib.import_module("newton._src.geometry.contact_reduction_global")
narrow = importlib.import_module("newton._src.geometry.narrow_phase")

# Warp structs are VALUE types: `contact_data = contact_template` (multicontact.py:761,938, collision_convex.py:197) copies.  The
# Python stand-in aliases the object instead, so the template's sort_sub_key -- the only field read back from the template after
# the alias -- compounds from one manifold contact to the next: k_0 = T << 3, k_i = (k_{i-1} << 3) | i instead of (T << 3) | i.
# The buffer's entry point undoes exactly that (i = k & 7, T = k >> 3 (i + 1)); everything downstream sees Warp's fingerprints.
_export_to_buffer = g.export_contact_to_buffer


def _export_with_warp_fingerprint(shape_a=None, shape_b=None, position=None, normal=None, depth=None, fingerprint=None,
                                  reducer_data=None):
    k = int(fingerprint)
    i = k & 7
    return _export_to_buffer(shape_a=shape_a, shape_b=shape_b, position=position, normal=normal, depth=depth,
                             fingerprint=((k >> (3 * (i + 1))) << 3) | i, reducer_data=reducer_data)


g.export_contact_to_buffer = _export_with_warp_fingerprint
_captured = []


@wp.func
def recording_writer(contact_data, writer_data, output_index):
    c = contact_data
    _captured.append((int(c.shape_a), int(c.shape_b), int(c.sort_sub_key), [float(x) for x in c.contact_point_center],
                      [float(x) for x in c.contact_normal_a_to_b], float(c.contact_distance), float(c.margin_a), float(c.margin_b),
                      float(c.radius_eff_a), float(c.radius_eff_b), float(c.gap_sum)))


@wp.kernel
def register(order, reducer_data, shape_transform, aabb_lo, aabb_hi, res):
    i = order[wp.tid()]
    g.reduce_contact_in_hashtable(i + 1, reducer_data, wp.static(g.BETA_THRESHOLD), shape_transform, aabb_lo, aabb_hi, res)


def run(s, reverse):
    A = wp.to_array
    S = len(s["shape_gap"])
    source = np.zeros(S, np.uint64)
    for k in range(S):
        if s["tri_count"][k] > 0:
            v0, nv, t0, nt = int(s["vertex_start"][k]), int(s["vertex_count"][k]), int(s["tri_start"][k]), int(s["tri_count"][k])
            source[k] = wp.Mesh(points=A(s["vertices"][v0:v0 + nv], wp.vec3),
                                indices=A(s["indices"][t0:t0 + nt].reshape(-1), wp.int32)).id
        elif s["hull_count"][k] > 0:  # CONVEX_MESH partner: wp.Mesh.points of the hull
            h0, nh = int(s["hull_start"][k]), int(s["hull_count"][k])
            source[k] = wp.Mesh(points=A(s["hull_points"][h0:h0 + nh], wp.vec3)).id
    P = len(s["pairs"])
    max_tri = (int(s["tri_count"].sum()) + int(sum(2 * (r - 1) * (c - 1) for _, r, c, *_ in s["hf_table"]))) * max(P, 1) + 8
    xf = A(s["shape_transform"], wp.transform)
    types, data, gap = A(s["shape_type"], wp.int32), A(s["shape_data"], wp.vec4), A(s["shape_gap"], wp.float32)
    src = A(source, wp.uint64)
    lo, hi, res = A(s["aabb_lo"], wp.vec3), A(s["aabb_hi"], wp.vec3), A(s["res"], wp.vec3i)
    hf = importlib.import_module("newton._src.utils.heightfield")
    hf_index, hf_data, hf_elev = A(s["hf_index"], wp.int32), wp.Array(), A(s["hf_elev"], wp.float32)
    for off, nrow, ncol, hx, hy, zlo, zhi in s["hf_table"]:
        d = hf.HeightfieldData()
        d.data_offset, d.nrow, d.ncol = int(off), int(nrow), int(ncol)
        d.hx, d.hy, d.min_z, d.max_z = (np.float32(v) for v in (hx, hy, zlo, zhi))
        hf_data.append(d)
    tri_pairs = wp.zeros(max_tri, dtype=wp.vec3i)
    tri_count = A(np.zeros(1, np.int32), wp.int32)
    wp.launch(narrow.narrow_phase_find_mesh_triangle_overlaps_kernel, dim=[P, 1],
              inputs=[types, xf, src, gap, data, A(np.zeros(S, np.float32), wp.float32), lo, hi, hf_index, hf_data,
                      A(s["pairs"], wp.vec2i), A(np.array([P], np.int32), wp.int32), P],
              outputs=[tri_pairs, tri_count])
    n_tri = int(tri_count[0])
    assert n_tri <= max_tri
    triples = np.array([[int(c) for c in tri_pairs[i]] for i in range(n_tri)], np.int32).reshape(-1, 3)
    triples = triples[np.lexsort((triples[:, 2], triples[:, 1], triples[:, 0]))] if n_tri else triples
    reducer = g.GlobalContactReducer(capacity=max(8 * n_tri, 64), device="cpu", deterministic=True)
    rdata = reducer.get_data_struct()
    threads = max(n_tri, 1)
    wp.launch(g.mesh_triangle_contacts_to_reducer_kernel, dim=threads,
              inputs=[types, data, xf, src, gap, hf_index, hf_data, hf_elev, tri_pairs, tri_count, rdata, threads])
    n = int(reducer.contact_count.numpy()[0])
    pd = reducer.position_depth.numpy().reshape(-1, 4)[1:n + 1]  # contact ids start at 1 (0 = empty slot)
    pair = reducer.shape_pairs.numpy().reshape(-1, 2)[1:n + 1].astype(np.int32)
    fp = reducer.contact_fingerprints.numpy()[1:n + 1].astype(np.int32)
    octs = np.array([[float(v[0]), float(v[1])] for v in reducer.normal.numpy()[1:n + 1]], np.float32).reshape(-1, 2)
    srt = np.lexsort((fp, pair[:, 1], pair[:, 0])) if n else np.zeros(0, np.int64)
    buffered = dict(pair=pair[srt], fp=fp[srt], pos=pd[srt, :3].astype(np.float32), depth=pd[srt, 3].astype(np.float32), oct=octs[srt])
    order = np.arange(n, dtype=np.int32)
    if n:
        wp.launch(register, dim=n, inputs=[A(order[::-1].copy() if reverse else order, wp.int32), rdata, xf, lo, hi, res])
    assert int(reducer.ht_insert_failures.numpy()[0]) == 0
    kernel = g.create_export_reduced_contacts_kernel(recording_writer)
    del _captured[:]
    blocks = 4
    wp.launch(kernel, dim=(blocks, g.EXPORT_REDUCED_CONTACTS_BLOCK_DIM),
              inputs=[rdata.ht_keys, rdata.ht_values, rdata.ht_active_slots, rdata.position_depth, rdata.normal, rdata.shape_pairs,
                      rdata.contact_fingerprints, rdata.exported_flags, types, data, gap, None, blocks, 0, 1])
    return triples, buffered, sorted(_captured, key=lambda r: (r[0], r[1], r[2]))


def main():
    import mesh_triangle_cases as mc

    rec = {}
    for name in mc.CASES + mc.HF_CASES:
        s = mc.scene(name)
        triples, buffered, fwd = run(s, False)
        _, _, rev = run(s, True)
        assert fwd == rev, name
        print(f"{name}: {int(s['tri_count'].sum()) + int(sum(2 * (r - 1) * (c - 1) for _, r, c, *_ in s['hf_table']))} triangles, {len(triples)} triangle pairs, {len(buffered['fp'])} buffered contacts, "
              f"{len(fwd)} after the reduction")
        rec[f"{name}/tri_pairs"] = triples
        for k, v in buffered.items():
            rec[f"{name}/buffered_{k}"] = v
        rec[f"{name}/pair"] = np.array([[r[0], r[1]] for r in fwd], np.int32).reshape(-1, 2)
        rec[f"{name}/fp"] = np.array([r[2] for r in fwd], np.int32)
        rec[f"{name}/pos"] = np.array([r[3] for r in fwd], np.float32).reshape(-1, 3)
        rec[f"{name}/normal"] = np.array([r[4] for r in fwd], np.float32).reshape(-1, 3)
        rec[f"{name}/depth"] = np.array([r[5] for r in fwd], np.float32)
        rec[f"{name}/misc"] = np.array([r[6:] for r in fwd], np.float32).reshape(-1, 5)
    path = os.path.join(HERE, "mesh_triangle_reference_vectors.npz")
    np.savez_compressed(path, **rec)
    print("wrote", len(rec), "arrays to", path)


if __name__ == "__main__":
    main()
