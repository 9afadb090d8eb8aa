#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/reduce_reference_vectors.npz by EXECUTING the reference's global contact
reducer in this container (see tests/golden/refshim): every contact of a case (tests/golden/reduce_cases.py) goes through the
reference's own `export_and_reduce_contact_centered_two_spatial_depths` (newton/_src/geometry/contact_reduction_global.py:
1519-1752, deterministic packing, hashtable from hashtable.py) in the case's arrival order AND in the reverse order, then the
reference's `export_reduced_contacts_kernel` (:2133-2290) hands the survivors to a recording writer.  The record holds the
surviving contacts sorted by (shape a, shape b, fingerprint) -- the order `deterministic=True` sorts them into -- and must be
the same for both arrival orders.
Run from the repo root:  python tests/golden/make_reduce_reference_vectors.py"""
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

lazy_ref.install()
import warp as wp  # noqa: E402  (the stand-in)

warnings.filterwarnings("ignore", category=RuntimeWarning)  # the hash mixer multiplies uint64 with wrap-around
g = importlib.import_module("newton._src.geometry.contact_reduction_global")

_captured = []


@wp.func
def recording_writer(contact_data, writer_data, output_index):
    c = contact_data
    _captured.append((int(c.shape_a), int(c.shape_b), int(c.sort_sub_key), [float(x) for x in c.contact_point_center],
                      [float(x) for x in c.contact_normal_a_to_b], float(c.contact_distance), float(c.margin_a), float(c.margin_b),
                      float(c.radius_eff_a), float(c.radius_eff_b), float(c.gap_sum)))


@wp.kernel
def push_contacts(order, pair, pos, normal, depth, fp, centered, inner, outer, local, aabb_lo, aabb_hi, res, reducer_data, ids):
    i = order[wp.tid()]
    ids[i] = g.export_and_reduce_contact_centered_two_spatial_depths(
        pair[i][0], pair[i][1], pos[i], normal[i], depth[i], fp[i], centered[i], inner[i], outer[i], local[i], aabb_lo[i],
        aabb_hi[i], res[i], reducer_data)


def run(rows_packed, order, n_shapes):
    p = rows_packed
    n = len(p["fp"])
    reducer = g.GlobalContactReducer(capacity=max(2 * n, 64), device="cpu", deterministic=True)
    data = reducer.get_data_struct()
    A = wp.to_array
    ids = wp.zeros(n, dtype=wp.int32)
    wp.launch(push_contacts, dim=n, inputs=[A(order, wp.int32), A(p["pair"], wp.vec2i), A(p["pos"], wp.vec3),
                                            A(p["normal"], wp.vec3), A(p["depth"], wp.float32), A(p["fp"], wp.int32),
                                            A(p["centered"], wp.vec3), A(p["inner"], wp.float32), A(p["outer"], wp.float32),
                                            A(p["local"], wp.vec3), A(p["aabb_lo"], wp.vec3), A(p["aabb_hi"], wp.vec3),
                                            A(p["res"], wp.vec3i), data, ids])
    assert int(reducer.ht_insert_failures.numpy()[0]) == 0
    kernel = g.create_export_reduced_contacts_kernel(recording_writer)
    # meshes: type MESH, no effective radius; margin in shape_data[3]; a small per-shape gap
    from newton._src.geometry.types import GeoType

    shape_types = A(np.full(n_shapes, int(GeoType.MESH), np.int32), wp.int32)
    shape_data = A(np.tile(np.array([1.0, 1.0, 1.0, 0.0005], np.float32), (n_shapes, 1)), wp.vec4)
    shape_gap = A(np.full(n_shapes, 0.004, np.float32), wp.float32)
    del _captured[:]
    blocks = 4
    wp.launch(kernel, dim=(blocks, g.EXPORT_REDUCED_CONTACTS_BLOCK_DIM),
              inputs=[data.ht_keys, data.ht_values, data.ht_active_slots, data.position_depth, data.normal, data.shape_pairs,
                      data.contact_fingerprints, data.exported_flags, shape_types, shape_data, shape_gap, None, blocks, 0, 1])
    out = sorted(_captured, key=lambda r: (r[0], r[1], r[2]))
    stored = int(sum(1 for i in ids.numpy() if i >= 0))
    return out, stored


def main():
    import reduce_cases as rc

    rec = {}
    for name in rc.CASES:
        p = rc.pack(rc.contacts(name))
        n = len(p["fp"])
        fwd, stored = run(p, np.arange(n, dtype=np.int32), 16)
        rev, stored_rev = run(p, np.arange(n, dtype=np.int32)[::-1].copy(), 16)
        assert [(r[0], r[1], r[2]) for r in fwd] == [(r[0], r[1], r[2]) for r in rev], name
        assert fwd == rev, name
        print(f"{name}: {n} contacts in, {len(fwd)} out ({stored} / {stored_rev} buffered in arrival / reverse order)")
        rec[f"{name}/pair"] = np.array([[r[0], r[1]] for r in fwd], np.int32).reshape(-1, 2)
        rec[f"{name}/fp"] = np.array([r[2] for r in fwd], np.int32)
        rec[f"{name}/pos"] = np.array([r[3] for r in fwd], np.float32).reshape(-1, 3)
        rec[f"{name}/normal"] = np.array([r[4] for r in fwd], np.float32).reshape(-1, 3)
        rec[f"{name}/depth"] = np.array([r[5] for r in fwd], np.float32)
        rec[f"{name}/misc"] = np.array([r[6:] for r in fwd], np.float32).reshape(-1, 5)
    path = os.path.join(HERE, "reduce_reference_vectors.npz")
    np.savez_compressed(path, **rec)
    print("wrote", len(rec), "arrays to", path)


if __name__ == "__main__":
    main()
