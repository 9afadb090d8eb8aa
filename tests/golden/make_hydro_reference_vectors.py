#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/hydro_reference_vectors.npz by EXECUTING the reference's hydroelastic pipeline
(newton/_src/geometry/sdf_hydroelastic.py) in this container on the stand-in of tests/golden/refshim, stage by stage as
HydroelasticSDF.launch strings them together (:905-1296), reduce_contacts=False:

    broadphase_collision_pairs_count (:1330)  SAT of the SDF boxes, finer-SDF-is-B normalisation, blocks per pair
    broadphase_collision_pairs_scatter (:1380) block records
    count_iso_voxels_block (:1465) x 4 levels (8, 4, 2, 1 voxels) + scatter_iso_subblock (:1671), prefix sums in between
    generate_contacts_kernel (:1982, pre_prune off) -> the contact buffer of the GlobalContactReducer
    decode_contacts_kernel (:1823)            ContactData per face, collected by a recording writer

The child levels run through the non-cooperative count kernel (n_blocks = 2): the cooperative eight-lanes-per-parent variant
(:1583) evaluates the same test and needs a workgroup of at least eight lanes (the CPU device has one).  Marching-cubes tables:
newton_amd/mc_tables.py (Warp's wp.MarchingCubes tables are not part of /root/reference).
Run from the repo root:  python tests/golden/make_hydro_reference_vectors.py"""
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
                 dummy_modules=("newton._src.sim.builder",),
                 f32_literals=("newton._src.geometry.sdf_hydroelastic", "newton._src.geometry.sdf_texture", "newton._src.geometry.sdf_mc",
                               "newton._src.geometry.collision_core", "newton._src.geometry.contact_reduction_hydroelastic"))
import warp as wp  # noqa: E402  (the stand-in)

warnings.filterwarnings("ignore", category=RuntimeWarning)
import hydro_cases  # noqa: E402
from make_sdf_reference_vectors import texture_sdf_data  # noqa: E402

from newton_amd.mc_tables import tables  # noqa: E402

hy = importlib.import_module("newton._src.geometry.sdf_hydroelastic")
g = importlib.import_module("newton._src.geometry.contact_reduction_global")

_rows = []


@wp.func
def recording_writer(c, writer_data, output_index):
    _rows.append((int(c.shape_a), int(c.shape_b), int(c.sort_sub_key), *[float(x) for x in c.contact_point_center],
                  *[float(x) for x in c.contact_normal_a_to_b], float(c.contact_distance), float(c.contact_stiffness), float(c.gap_sum)))


class _Writer:
    pass


def run(s, margin_contact_area=1.0e-2, edge_clamp_min=0.02):
    A = wp.to_array
    n_shapes, n_pairs = len(s["X"]), len(s["pairs"])
    sdf_data = wp.Array([texture_sdf_data(t) for t in s["sdfs"]])
    X = A(s["X"], wp.transform)
    Xinv = wp.Array([wp.transform_inverse(x) for x in X])
    shape_data, gap = A(s["data"], wp.vec4), A(s["gap"], float)
    pdata = hy.LinearPressureData()
    pdata.shape_kh = A(s["kh"], float)
    pairs, pair_count = A(s["pairs"], wp.vec2i), A(np.array([n_pairs], np.int32), int)
    # ---- broad phase
    nblk, norm = wp.zeros(n_pairs, dtype=int), wp.zeros(n_pairs, dtype=wp.vec2i)
    wp.launch(hy.broadphase_collision_pairs_count, dim=n_pairs, inputs=[X, sdf_data, pairs, pair_count], outputs=[nblk, norm])
    prefix = np.concatenate([[0], np.cumsum(nblk.numpy())]).astype(np.int32)
    total = int(prefix[-1])
    records = wp.zeros(max(total, 1), dtype=wp.vec3ui) if hasattr(wp, "vec3ui") else None
    records = wp.Array([wp.vec3i(0, 0, 0) for _ in range(max(total, 1))])
    wp.launch(hy.broadphase_collision_pairs_scatter, dim=1,
              inputs=[1, A(np.array([total], np.int32), int), A(prefix[:-1] if n_pairs else prefix, int), norm, pair_count, sdf_data, max(total, 1)],
              outputs=[records])
    # ---- octree
    count_int = hy.create_count_iso_voxels_block_kernel(hy.linear_pressure, True, False)
    count_frac = hy.create_count_iso_voxels_block_kernel(hy.linear_pressure, False, False)
    n_in = total
    levels = []
    for size, n_blocks in ((8, 1), (4, 2), (2, 2), (1, 2)):
        cnt, idx = wp.zeros(max(n_in, 1), dtype=int), wp.Array([wp.uint8(0)] * max(n_in, 1))
        kernel = count_int if size % 2 == 0 else count_frac
        wp.launch(kernel, dim=1, inputs=[1, A(np.array([n_in], np.int32), int), sdf_data, shape_data, X, Xinv, pdata, records, norm, gap,
                                         size, n_blocks, max(n_in, 1)], outputs=[cnt, idx])
        pre = np.concatenate([[0], np.cumsum(cnt.numpy()[:n_in])]).astype(np.int32)
        n_out = int(pre[-1])
        out_rec = wp.Array([wp.vec3i(0, 0, 0) for _ in range(max(n_out, 1))])
        wp.launch(hy.scatter_iso_subblock, dim=1, inputs=[1, A(np.array([n_in], np.int32), int), A(pre[:-1] if n_in else pre, int), idx,
                                                          records, size, max(n_in, 1), max(n_out, 1)], outputs=[out_rec])
        levels.append(n_out)
        records, n_in = out_rec, n_out
    vox = np.array([[int(c) for c in hy.unpack_hydro_voxel_coords(records[i])] + [int(records[i][2])] for i in range(n_in)], np.int32).reshape(-1, 4)
    # ---- generate + decode
    tri_range, flat = tables()
    reducer = g.GlobalContactReducer(capacity=max(16 * n_in, 64), device="cpu", store_hydroelastic_data=True, deterministic=True)
    rd = reducer.get_data_struct()
    gen = hy.get_generate_contacts_kernel(False, pre_prune=False, deterministic_reduction=True, pressure_func=hy.linear_pressure,
                                          mc_edge_clamp_min=edge_clamp_min, paired_samples=False)
    flat_tab = wp.Array([wp.vec2i(int(a), int(b)) for a, b in np.asarray(flat).reshape(-1, 2)])
    wp.launch(gen, dim=1, inputs=[1, A(np.array([n_in], np.int32), int), sdf_data, shape_data, X, Xinv, pdata, records, norm,
                                  A(np.asarray(tri_range, np.int32), int), flat_tab, gap, max(n_in, 1), rd, wp.zeros(0, dtype=wp.vec3),
                                  wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=wp.vec3i)],
              outputs=[wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=float), wp.zeros(0, dtype=wp.vec2i)])
    dec = hy.get_decode_contacts_kernel(margin_contact_area, recording_writer)
    w = _Writer()
    w.contact_count, w.contact_max = wp.zeros(1, dtype=int), 1 << 30
    del _rows[:]
    wp.launch(dec, dim=1, inputs=[1, rd.contact_count, pdata.shape_kh, X, gap, rd.position_depth, rd.normal, rd.shape_pairs,
                                  rd.contact_fingerprints, rd.contact_area, rd.contact_pressure, reducer.capacity], outputs=[w])
    rows = np.asarray(_rows, np.float64).reshape(-1, 12)
    reduced = {}
    cr = importlib.import_module("newton._src.geometry.contact_reduction_hydroelastic")
    lo, hi = A(s["aabb_lo"], wp.vec3), A(s["aabb_hi"], wp.vec3)
    res = wp.Array([wp.vec3i(int(a), int(b), int(c)) for a, b, c in s["res"]])
    for tag, pre_prune, normal_matching, anchor, moment in (("prune_nm", True, True, False, False), ("full_nm", False, True, False, False),
                                                            ("prune_plain", True, False, False, False),
                                                            ("prune_anchor", True, True, True, False), ("prune_moment", True, True, False, True),
                                                            ("full_moment_plain", False, False, False, True)):
        # generate (aggregates per normal bin, optional local-first pruning) -> reduce -> export, the non-deterministic variant
        # executed sequentially: contact ids, hashtable entries and float sums in thread order
        red = cr.HydroelasticContactReduction(capacity=max(16 * n_in, 64), device="cpu", writer_func=recording_writer_reduced,
                                              config=cr.HydroelasticReductionConfig(normal_matching=normal_matching, anchor_contact=anchor,
                                                                                     moment_matching=moment,
                                                                                     margin_contact_area=margin_contact_area),
                                              deterministic=False)
        red.clear()
        rd2 = red.get_data_struct()
        gen2 = hy.get_generate_contacts_kernel(False, pre_prune=pre_prune, deterministic_reduction=False, pressure_func=hy.linear_pressure,
                                               mc_edge_clamp_min=edge_clamp_min, paired_samples=False)
        wp.launch(gen2, dim=1, inputs=[1, A(np.array([n_in], np.int32), int), sdf_data, shape_data, X, Xinv, pdata, records, norm,
                                       A(np.asarray(tri_range, np.int32), int), flat_tab, gap, max(n_in, 1), rd2, lo, hi, res],
                  outputs=[wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=float), wp.zeros(0, dtype=wp.vec2i)])
        n_buf = int(rd2.contact_count.numpy()[0])
        red.reduce(pdata.shape_kh, X, lo, hi, res, 1)
        w2 = _Writer()
        w2.contact_count, w2.contact_max = wp.zeros(1, dtype=int), 1 << 30
        del _rows[:]
        red.export(pdata.shape_kh, gap, X, w2, 1)
        r2 = np.asarray(_rows, np.float64).reshape(-1, 13)
        reduced[f"reduced_{tag}/rows"] = r2.astype(np.float32)
        reduced[f"reduced_{tag}/ids"] = r2[:, :3].astype(np.int64)
        reduced[f"reduced_{tag}/buffered"] = np.array([n_buf], np.int32)
    return dict(**reduced, normalized=np.array([[int(p[0]), int(p[1])] for p in norm], np.int32).reshape(-1, 2), blocks=nblk.numpy().astype(np.int32),
                levels=np.asarray(levels, np.int32), voxels=vox, rows=rows.astype(np.float32), ids=rows[:, :3].astype(np.int64))


@wp.func
def recording_writer_reduced(c, writer_data, output_index):
    _rows.append((int(c.shape_a), int(c.shape_b), int(c.sort_sub_key), *[float(x) for x in c.contact_point_center],
                  *[float(x) for x in c.contact_normal_a_to_b], float(c.contact_distance), float(c.contact_stiffness), float(c.gap_sum),
                  float(c.contact_friction_scale)))


def main():
    out = {}
    for name, s in hydro_cases.scenes().items():
        r = run(s)
        for k, v in r.items():
            out[f"{name}/{k}"] = v
        print(name, "blocks", r["blocks"], "levels", r["levels"], "rows", len(r["rows"]), flush=True)
    path = os.path.join(HERE, "hydro_reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
