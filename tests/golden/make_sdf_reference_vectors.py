#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/sdf_reference_vectors.npz by EXECUTING, in this container (tests/golden/refshim:
the reference's own source on a pure-Python stand-in for warp-lang), the texture-SDF samplers and the mesh-vs-SDF narrow phase
of /root/reference on the cases of tests/golden/sdf_cases.py:

  * sdf_texture.py: texture_sample_sdf (software trilinear, :1130), texture_sample_sdf_grad (:1560), texture_sample_sdf_at_voxel
    (:1220), texture_sample_sdf_hw (:1533), _texture_sample_sdf_hw_pair (:1385), texture_sample_sdf_grad_only_hw (:1736);
  * sdf_contact.py: do_edge_sdf_collision (:704-938, texture-only variant: golden pair + 3 Brent steps + endpoint checks) and the
    whole mesh_sdf_collision_kernel (:1098-1515: culling, Brent search, inner-cull consistency, corner ownership, gradient,
    scale_sdf_result_to_world, ContactData) with use_texture_sdf_only / precomputed edge data, contacts collected by a writer.

The TextureSDFData handed to them wraps the arrays of newton_amd.sdf.TextureSDF (scalar storage, paired_samples = False).  What
stays restated is Warp's native texture sampler (refshim/warp: Texture3D / texture_sample) and its vector builtins.
Run from the repo root:  python tests/golden/make_sdf_reference_vectors.py"""
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
                 dummy_modules=("newton._src.sim.builder",),
                 # kernel code whose scalar locals start from literals: float32 like Warp compiles them (lazy_ref._F32Literals)
                 f32_literals=("newton._src.geometry.sdf_contact", "newton._src.geometry.sdf_texture"))
import warp as wp  # noqa: E402  (the stand-in)

import sdf_cases  # noqa: E402

st = importlib.import_module("newton._src.geometry.sdf_texture")
sc = importlib.import_module("newton._src.geometry.sdf_contact")
cd = importlib.import_module("newton._src.geometry.contact_data")
hf = importlib.import_module("newton._src.utils.heightfield")


def texture_sdf_data(t):
    """TextureSDFData (sdf_texture.py:126-160) over the arrays of a newton_amd.sdf.TextureSDF."""
    d = st.TextureSDFData()
    d.coarse_texture = wp.Texture3D(np.asarray(t.coarse, np.float32))
    d.subgrid_texture = wp.Texture3D(np.asarray(t.subgrid))
    d.subgrid_start_slots = wp.to_array3d(np.asarray(t.slots, np.uint32))
    d.sdf_box_lower, d.sdf_box_upper = wp.vec3(*t.box_lower), wp.vec3(*t.box_upper)
    d.inv_sdf_dx = wp.vec3(*t.inv_dx)
    d.subgrid_size = int(t.subgrid_size)
    d.subgrid_size_f, d.subgrid_samples_f = wp.f32(t.subgrid_size), wp.f32(t.subgrid_size + 1)
    d.fine_to_coarse = wp.f32(1.0) / wp.f32(t.subgrid_size)
    d.voxel_size, d.voxel_radius = wp.vec3(*t.voxel_size), wp.f32(t.voxel_radius)
    d.subgrids_min_sdf_value, d.subgrids_sdf_value_range = wp.f32(t.min_value), wp.f32(t.value_range)
    d.paired_samples, d.scale_baked = False, bool(t.scale_baked)
    return d


def v3(x):
    return np.array([float(c) for c in x], np.float32)


def main():
    out = {}
    do_edge = sc._create_sdf_contact_funcs(False, True, st.texture_sample_sdf_hw, st._texture_sample_sdf_hw_pair)
    for name, t in sdf_cases.sdfs().items():
        d = texture_sdf_data(t)
        pts = sdf_cases.query_points(t)
        out[f"sample/{name}/points"] = pts
        out[f"sample/{name}/value"] = np.array([st.texture_sample_sdf(d, wp.vec3(*p)) for p in pts], np.float32)
        vg = [st.texture_sample_sdf_grad(d, wp.vec3(*p)) for p in pts]
        out[f"sample/{name}/grad_value"] = np.array([v for v, _ in vg], np.float32)
        out[f"sample/{name}/grad"] = np.array([v3(g) for _, g in vg], np.float32)
        out[f"sample/{name}/value_hw"] = np.array([st.texture_sample_sdf_hw(d, wp.vec3(*p)) for p in pts], np.float32)
        out[f"sample/{name}/grad_hw"] = np.array([v3(st.texture_sample_sdf_grad_only_hw(d, wp.vec3(*p))) for p in pts], np.float32)
        pair = [st._texture_sample_sdf_hw_pair(d, wp.vec3(*pts[i]), wp.vec3(*pts[-1 - i])) for i in range(len(pts) // 2)]
        out[f"sample/{name}/pair_hw"] = np.array([[float(v[0]), float(v[1])] for v in pair], np.float32)
        vox = sdf_cases.voxels(t)
        out[f"sample/{name}/voxels"] = vox
        out[f"sample/{name}/at_voxel"] = np.array([st.texture_sample_sdf_at_voxel(d, int(i), int(j), int(k)) for i, j, k in vox], np.float32)
        v0, v1, prec = sdf_cases.edges(t)
        res = []
        for a, b, p in zip(v0, v1, prec):
            mid = st.texture_sample_sdf_hw(d, wp.vec3(*((a + b) * np.float32(0.5))))
            dist, point, endpoint = do_edge(d, wp.uint64(0), wp.vec3(*a), wp.vec3(*b), mid, False, 0, False, hf.HeightfieldData(),
                                            wp.zeros(0, dtype=float), wp.f32(p))
            res.append([float(mid), float(dist), *v3(point), float(endpoint)])
        out[f"edge/{name}/v0"], out[f"edge/{name}/v1"], out[f"edge/{name}/precision"] = v0, v1, prec
        out[f"edge/{name}/result"] = np.asarray(res, np.float32)  # midpoint value, distance, point[3], endpoint code
        print(name, "sampled", flush=True)

    rows = []

    @wp.func
    def collect(c, writer_data, idx):
        rows.append((int(c.shape_a), int(c.shape_b), int(c.sort_sub_key), *v3(c.contact_point_center), *v3(c.contact_normal_a_to_b),
                     float(c.contact_distance), float(c.margin_a), float(c.margin_b), float(c.gap_sum)))

    kernel = sc.create_narrow_phase_process_mesh_mesh_contacts_kernel(collect, enable_heightfields=False, reduce_contacts=False,
                                                                     use_precomputed_edge_data=True, use_texture_sdf_only=True)
    A = wp.to_array
    for name, s in sdf_cases.pair_scenes().items():
        rows.clear()
        S_ = len(s["X"])
        table = wp.Array([texture_sdf_data(t) for t in s["sdfs"]])
        blocks = len(s["pairs"])
        wp.launch(kernel, dim=(blocks, 1), inputs=[
            A(s["data"], wp.vec4), A(s["X"], wp.transform), wp.Array([wp.uint64(1)] * S_), table, A(s["sdf_index"], int),
            A(np.zeros(S_, np.int32), int), A(s["gap"], float), A(s["gap"], float), wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=wp.vec3),
            wp.f32(0.0), wp.f32(0.0), wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=wp.vec3), wp.zeros(0, dtype=wp.vec3i),
            A(s["pairs"], wp.vec2i), A(np.array([blocks], np.int32), int), A(np.full(S_, -1, np.int32), int), wp.Array([]),
            wp.zeros(0, dtype=float), wp.zeros(0, dtype=wp.vec2i), A(s["ec"], wp.vec4), A(s["eh"], wp.vec4), A(s["er"], wp.vec2i),
            None, blocks])
        r = np.asarray(rows, np.float64).reshape(-1, 13)
        out[f"pair/{name}/rows"] = r.astype(np.float32)   # shape_a, shape_b, key, centre[3], normal[3], distance, margins, gap
        out[f"pair/{name}/ids"] = r[:, :3].astype(np.int32)
        print(name, len(r), "contacts", flush=True)
    path = os.path.join(HERE, "sdf_reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
