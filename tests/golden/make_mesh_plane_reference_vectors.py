#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/mesh_plane_reference_vectors.npz by EXECUTING, in this container (see
tests/golden/refshim), the reference's mesh-vs-infinite-plane leg on the scenes of tests/golden/mesh_plane_cases.py:

  * `narrow_phase_process_mesh_plane_contacts_reduce_kernel` (newton/_src/geometry/narrow_phase.py:1866-1990) with the writer the
    pipeline gives it under reduce_contacts=True, `write_contact_to_reducer` (contact_reduction_global.py:2059-2096): every mesh
    vertex within margin + gap of the plane becomes a buffered contact (record "<case>/buffered_*": the UNREDUCED list, in buffer
    order = vertex order per pair);
  * `reduce_contact_in_hashtable` (:1246-1346, what reduce_buffered_contacts_kernel calls per buffered contact, beta = 1e-4) in
    buffer order AND in reverse order, then `export_reduced_contacts_kernel` (:2133-2290) with a recording writer: the surviving
    contacts sorted by (shape a, shape b, fingerprint) (record "<case>/pair|fp|pos|normal|depth|misc"); both orders must agree.
Run from the repo root:  python tests/golden/make_mesh_plane_reference_vectors.py"""
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

warnings.filterwarnings("ignore", category=RuntimeWarning)
g = importlib.import_module("newton._src.geometry.contact_reduction_global")
narrow = importlib.import_module("newton._src.geometry.narrow_phase")

MESH_PLANE = narrow.create_narrow_phase_process_mesh_plane_contacts_kernel(g.write_contact_to_reducer, reduce_contacts=True)
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
    from newton._src.geometry.types import GeoType

    A = wp.to_array
    S = len(s["shape_gap"])
    total = int(s["vertex_count"].sum())
    reducer = g.GlobalContactReducer(capacity=max(2 * total, 64), device="cpu", deterministic=True)
    data = reducer.get_data_struct()
    source = np.zeros(S, np.uint64)
    for k in range(S):
        if s["vertex_count"][k] > 0:
            v0, n = int(s["vertex_start"][k]), int(s["vertex_count"][k])
            source[k] = wp.Mesh(points=A(s["vertices"][v0:v0 + n], wp.vec3)).id
    P = len(s["pairs"])
    xf = A(s["shape_transform"], wp.transform)
    lo, hi, res = A(s["aabb_lo"], wp.vec3), A(s["aabb_hi"], wp.vec3), A(s["res"], wp.vec3i)
    wp.launch(MESH_PLANE, dim=(P, 1),
              inputs=[A(s["shape_data"], wp.vec4), xf, A(source, wp.uint64), A(s["shape_gap"], wp.float32), lo, hi, res,
                      A(s["pairs"], wp.vec2i), A(np.array([P], np.int32), wp.int32),
                      A(np.arange(P + 1, dtype=np.int32), wp.int32),  # block_offsets: one block per pair
                      data, P])
    n = int(reducer.contact_count.numpy()[0])
    pd = reducer.position_depth.numpy().reshape(-1, 4)[1:n + 1]  # contact ids start at 1 (0 = empty slot)
    buffered = dict(pair=reducer.shape_pairs.numpy().reshape(-1, 2)[1:n + 1].astype(np.int32),
                    fp=reducer.contact_fingerprints.numpy()[1:n + 1].astype(np.int32), pos=pd[:, :3].astype(np.float32),
                    depth=pd[:, 3].astype(np.float32),
                    oct=np.array([[float(v[0]), float(v[1])] for v in reducer.normal.numpy()[1:n + 1]], np.float32).reshape(-1, 2))
    order = np.arange(n, dtype=np.int32)
    if n:
        wp.launch(register, dim=n, inputs=[A(order[::-1].copy() if reverse else order, wp.int32), data, xf, lo, hi, res])
    assert int(reducer.ht_insert_failures.numpy()[0]) == 0
    kernel = g.create_export_reduced_contacts_kernel(recording_writer)
    shape_types = A(np.where(s["vertex_count"] > 0, int(GeoType.MESH), int(GeoType.PLANE)).astype(np.int32), wp.int32)
    del _captured[:]
    blocks = 4
    wp.launch(kernel, dim=(blocks, g.EXPORT_REDUCED_CONTACTS_BLOCK_DIM),
              inputs=[data.ht_keys, data.ht_values, data.ht_active_slots, data.position_depth, data.normal, data.shape_pairs,
                      data.contact_fingerprints, data.exported_flags, shape_types, A(s["shape_data"], wp.vec4),
                      A(s["shape_gap"], wp.float32), None, blocks, 0, 1])
    return buffered, sorted(_captured, key=lambda r: (r[0], r[1], r[2]))


def main():
    import mesh_plane_cases as mc

    rec = {}
    for name in mc.CASES:
        s = mc.scene(name)
        buffered, fwd = run(s, False)
        _, rev = run(s, True)
        assert fwd == rev, name
        print(f"{name}: {int(s['vertex_count'].sum())} vertices, {len(buffered['fp'])} buffered contacts, {len(fwd)} after the reduction")
        for k, v in buffered.items():
            rec[f"{name}/buffered_{k}"] = v
        rec[f"{name}/pair"] = np.array([[r[0], r[1]] for r in fwd], np.int32).reshape(-1, 2)
        rec[f"{name}/fp"] = np.array([r[2] for r in fwd], np.int32)
        rec[f"{name}/pos"] = np.array([r[3] for r in fwd], np.float32).reshape(-1, 3)
        rec[f"{name}/normal"] = np.array([r[4] for r in fwd], np.float32).reshape(-1, 3)
        rec[f"{name}/depth"] = np.array([r[5] for r in fwd], np.float32)
        rec[f"{name}/misc"] = np.array([r[6:] for r in fwd], np.float32).reshape(-1, 5)
    path = os.path.join(HERE, "mesh_plane_reference_vectors.npz")
    np.savez_compressed(path, **rec)
    print("wrote", len(rec), "arrays to", path)


if __name__ == "__main__":
    main()
