"""TEST INFRASTRUCTURE ONLY (never imported by newton_amd/ or by bench.py's timed path).

The reference's mesh-vs-convex leg (a triangle mesh WITHOUT the SDF route against a primitive or a convex hull) on the CPU:

  * midphase + contact generation   liboracle.so `o_mesh_triangle_contacts` (oracle/oracle_convex.cpp): the C++ restatement of
                                    collision_core.py:996-1180 (query AABB from the support function, front-face test),
                                    contact_reduction_global.py:2299-2403 (triangle shape, back-face culling, GJK / MPR + manifold
                                    with the TRIANGLE support map / Minkowski seed of support_function.py:174-191,467-538)
  * buffering                       contact_reduction_global.py:2059-2096 (write_contact_to_reducer: no gap test)
  * reduction + export              oracle_reduce.reduce_buffered_contacts (reduce_contact_in_hashtable :1246-1346 + the export)

Pinned by tests/golden/mesh_triangle_reference_vectors.npz, the record of the reference's own kernels executed on eight scenes
(tests/golden/make_mesh_triangle_reference_vectors.py).  Warp's BVH is native code and not restated: the triangle set of a query is
every triangle whose float32 bounds touch the query box (parity of the SET is what the record pins)."""
import ctypes as C
import os

import numpy as np

import oracle_reduce as orr

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle.so"))
        _LIB.o_mesh_triangle_contacts.restype = C.c_int
    return _LIB


def triangle_contacts(s):
    """Scene (tests/golden/mesh_triangle_cases.scene) -> (triples [n][3] sorted (mesh, convex, triangle), buffered contact dict sorted
    by (mesh, convex, fingerprint): pair, fp, pos, normal, depth + per contact shape a's transform / local AABB / voxel resolution)."""
    lib = _lib()
    keep = []

    def p(a, dt):
        x = np.ascontiguousarray(a, dtype=dt)
        keep.append(x)
        return x.ctypes.data_as(C.c_void_p)

    pairs = np.asarray(s["pairs"], np.int32).reshape(-1, 2)
    hf_tris = int(sum(2 * (int(r) - 1) * (int(c) - 1) for _, r, c, *_ in s.get("hf_table", ())))
    tri_cap = (int(s["tri_count"].sum()) + hf_tris) * max(len(pairs), 1) + 8
    cap = 8 * tri_cap
    tri_out, out, n_tri = np.zeros((tri_cap, 3), np.int32), np.zeros((cap, 10), np.float32), C.c_int(0)
    n = lib.o_mesh_triangle_contacts(len(pairs), p(pairs, np.int32), p(s["shape_type"], np.int32), p(s["shape_transform"], np.float32),
                                     p(s["shape_data"], np.float32), p(s["shape_gap"], np.float32), p(s["vertex_start"], np.int32),
                                     p(s["tri_start"], np.int32), p(s["tri_count"], np.int32), p(s["vertices"], np.float32),
                                     p(s["indices"], np.int32),
                                     p(s.get("hull_start", np.zeros(len(s["shape_gap"]), np.int32)), np.int32),
                                     p(s.get("hull_count", np.zeros(len(s["shape_gap"]), np.int32)), np.int32),
                                     p(s["hull_points"] if len(s.get("hull_points", ())) else np.zeros((1, 3), np.float32), np.float32),
                                     p(s.get("hf_index", np.full(len(s["shape_gap"]), -1, np.int32)), np.int32),
                                     p(s["hf_table"] if len(s.get("hf_table", ())) else np.zeros((1, 7), np.float32), np.float32),
                                     p(s["hf_elev"] if len(s.get("hf_elev", ())) else np.zeros(1, np.float32), np.float32),
                                     p(s["aabb_lo"], np.float32), p(s["aabb_hi"], np.float32),
                                     tri_out.ctypes.data_as(C.c_void_p), tri_cap, C.byref(n_tri),
                                     out.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap and n_tri.value <= tri_cap
    triples = tri_out[: n_tri.value]
    triples = triples[np.lexsort((triples[:, 2], triples[:, 1], triples[:, 0]))] if len(triples) else triples
    rows = out[:n]
    pair, fp = rows[:, 0:2].astype(np.int32), rows[:, 2].astype(np.int32)
    srt = np.lexsort((fp, pair[:, 1], pair[:, 0])) if n else np.zeros(0, np.int64)
    rows, pair, fp = rows[srt], pair[srt], fp[srt]
    a = pair[:, 0] if n else np.zeros(0, np.int32)
    c = dict(pair=pair.reshape(-1, 2), fp=fp, pos=rows[:, 3:6].copy(), normal=rows[:, 6:9].copy(), depth=rows[:, 9].copy(),
             xform_a=np.asarray(s["shape_transform"], np.float32)[a].reshape(-1, 7),
             aabb_lo=np.asarray(s["aabb_lo"], np.float32)[a].reshape(-1, 3), aabb_hi=np.asarray(s["aabb_hi"], np.float32)[a].reshape(-1, 3),
             res=np.asarray(s["res"], np.int32)[a].reshape(-1, 3))
    return triples, c


def effective_radius(shape_type, shape_data_row):
    """compute_effective_radius (contact_reduction_global.py export path): spheres and capsules carry their radius."""
    return np.float32(shape_data_row[0]) if int(shape_type) in (3, 4) else np.float32(0.0)


def mesh_triangle_rows(s):
    """Scene -> the reduced contacts sorted by (mesh shape, convex shape, fingerprint): dict(pair, fp, pos, normal, depth, margin_a,
    margin_b, radius_a, radius_b) -- what export_reduced_contacts_kernel hands to the contact writer."""
    _, c = triangle_contacts(s)
    out = orr.reduce_buffered_contacts(c)
    out["margin_a"] = np.asarray([s["shape_data"][a][3] for a, _ in out["pair"]], np.float32)
    out["margin_b"] = np.asarray([s["shape_data"][b][3] for _, b in out["pair"]], np.float32)
    out["radius_a"] = np.asarray([effective_radius(s["shape_type"][a], s["shape_data"][a]) for a, _ in out["pair"]], np.float32)
    out["radius_b"] = np.asarray([effective_radius(s["shape_type"][b], s["shape_data"][b]) for _, b in out["pair"]], np.float32)
    return out
