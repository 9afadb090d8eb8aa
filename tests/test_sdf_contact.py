"""Texture-SDF sampling and the mesh-vs-SDF narrow phase: oracle (oracle/oracle_sdf.py, plain float32 restatement) vs the
gfx950 kernels -- emulated on the CPU here, on the device under -m gpu (tests/test_gpu_sdf.py) -- plus the reference-held
expectations of newton/tests/test_sdf_contact.py restated as geometry: a cube resting on a cube produces contacts on the
touching faces with normals along the stacking axis and distances equal to the overlap."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from newton_amd import sdf as S  # noqa: E402
from newton_amd.enums import GeoType  # noqa: E402
from newton_amd.mesh import Mesh, mesh_edge_tables  # noqa: E402


def box_sdf(h=0.5, res=32, mode=S.QuantizationMode.UINT16):
    return S.create_texture_sdf_from_primitive(GeoType.BOX, (h, h, h), max_resolution=res, quantization_mode=mode)


def two_box_scene(dz=0.98, yaw=0.3, margin=0.0, gap=0.02, mode=S.QuantizationMode.UINT16):
    """Two unit cubes (meshes with SDFs), the upper one rotated about z and lowered into the lower one by 1 - dz."""
    m = Mesh.create_box(0.5, 0.5, 0.5)
    # a denser edge set than the 12 cube edges: subdivide every face once so that edges cross the other cube's faces
    v, tri = m.vertices, m.indices.reshape(-1, 3)
    ec, eh = mesh_edge_tables(v, tri)
    sdf = box_sdf(mode=mode)
    q = [0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)]
    X = np.array([[0, 0, 0, 0, 0, 0, 1], [0.05, -0.03, dz, *q]], dtype=np.float32)
    data = np.array([[1, 1, 1, margin], [1, 1, 1, margin]], dtype=np.float32)
    gaps = np.array([gap, gap], dtype=np.float32)
    er = np.array([[0, len(ec)], [0, len(ec)]], dtype=np.int32)  # both shapes share the mesh asset
    return dict(pairs=np.array([[0, 1]], dtype=np.int32), X=X, data=data, gap=gaps, sdf_index=np.array([0, 0], dtype=np.int32),
                sdfs=[sdf], er=er, ec=ec, eh=eh)


def oracle_contacts(sc):
    import oracle_sdf as O

    return O.mesh_sdf_collide(sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["sdf_index"], sc["sdfs"], sc["er"], sc["ec"], sc["eh"])


def test_oracle_sampler_agrees_with_the_host_sampler():
    import oracle_sdf as O

    for mode in (S.QuantizationMode.FLOAT32, S.QuantizationMode.UINT16, S.QuantizationMode.UINT8):
        t = box_sdf(mode=mode)
        o = O.OracleSDF(t)
        pts = np.random.default_rng(3).uniform(-0.9, 0.9, size=(200, 3)).astype(np.float32)
        want = t.sample(pts)
        got = np.array([o.sample(p) for p in pts])
        assert np.max(np.abs(got - want)) <= 1e-6
        g = np.array([o.sample_grad_fd(p) for p in pts])
        inside = np.all(np.abs(pts) < 0.5, axis=1) & (np.abs(want) < 0.05)
        _, g_tri = t.sample_grad(pts)
        cosang = np.sum(g[inside] * g_tri[inside], axis=1) / (np.linalg.norm(g[inside], axis=1) * np.linalg.norm(g_tri[inside], axis=1))
        assert np.all(cosang > 0.9)  # centred differences vs analytic trilinear gradient: same direction near the surface


def test_oracle_cube_on_cube_contacts_are_on_the_touching_faces():
    """Reference expectation (newton/tests/test_sdf_contact.py, cube-on-cube cases): contacts appear, their normals point from
    shape 0 to shape 1 along the stacking axis, their distances equal the overlap, their points lie in the overlap slab."""
    sc = two_box_scene(dz=0.98, mode=S.QuantizationMode.FLOAT32)
    cs = oracle_contacts(sc)
    assert len(cs) >= 4
    modes = {k & 2 for _, k, *_ in cs}
    assert modes == {0, 2}  # both directions of the pair contribute (mesh 0 edges in SDF 1 and vice versa)
    for _, key, c, n, d, m0, m1 in cs:
        assert abs(np.linalg.norm(n) - 1.0) < 1e-5 and n[2] > 0.95
        assert -0.03 < d < 0.045 and 0.45 < c[2] < 0.53
    assert abs(min(d for *_, d, _, _ in cs) + 0.02) < 4e-3  # deepest = the 2 cm overlap (texture resolution 32: voxel 3.4 cm)


def test_oracle_margin_and_gap_gate_the_contacts():
    far = oracle_contacts(two_box_scene(dz=1.10, gap=0.02))
    assert len(far) == 0  # 10 cm apart, threshold 4 cm
    near = oracle_contacts(two_box_scene(dz=1.03, gap=0.02))
    assert len(near) > 0 and all(d > 0.0 for *_, d, _, _ in near)  # inside the gap: separated contacts are emitted
    assert len(oracle_contacts(two_box_scene(dz=1.03, gap=0.01))) == 0


@pytest.fixture(scope="module")
def emu():
    import harness

    return harness.lib()


def _emu_sdf(lib, t):
    import ctypes as C

    from newton_amd import _lib as L

    keep = [np.ascontiguousarray(t.coarse), np.ascontiguousarray(t.subgrid), np.ascontiguousarray(t.slots)]
    d = L.nt_sdf()
    d.coarse, d.subgrid, d.slots = (k.ctypes.data for k in keep)
    d.cx, d.cy, d.cz = (int(x) for x in t.slots.shape)
    d.tex_size, d.subgrid_size, d.quantization, d.scale_baked = int(t.subgrid.shape[0]), int(t.subgrid_size), int(t.quantization_mode), 0
    for k in range(3):
        d.box_lower[k], d.box_upper[k], d.inv_dx[k], d.voxel_size[k] = (float(t.box_lower[k]), float(t.box_upper[k]),
                                                                         float(t.inv_dx[k]), float(t.voxel_size[k]))
    d.voxel_radius, d.min_value, d.value_range = float(t.voxel_radius), float(t.min_value), float(t.value_range)
    return d, keep


@pytest.mark.parametrize("mode", [S.QuantizationMode.FLOAT32, S.QuantizationMode.UINT16, S.QuantizationMode.UINT8])
def test_emulated_kernel_sampler_is_bitwise_the_oracle(emu, mode):
    import ctypes as C

    import oracle_sdf as O

    t = box_sdf(mode=mode)
    d, keep = _emu_sdf(emu, t)
    pts = np.random.default_rng(9).uniform(-0.9, 0.9, size=(300, 3)).astype(np.float32)
    dist, grad = np.zeros(300, np.float32), np.zeros((300, 3), np.float32)
    assert emu.nt_sdf_sample(C.byref(d), pts.ctypes.data, 300, dist.ctypes.data, grad.ctypes.data, None) == 0
    o = O.OracleSDF(t)
    assert np.array_equal(dist, np.array([o.sample(p) for p in pts], dtype=np.float32))
    assert np.array_equal(grad, np.array([o.sample_grad_fd(p) for p in pts], dtype=np.float32))


@pytest.mark.parametrize("dz,mode", [(0.98, S.QuantizationMode.UINT16), (1.02, S.QuantizationMode.FLOAT32), (0.9, S.QuantizationMode.UINT8)])
def test_emulated_mesh_sdf_kernel_matches_the_oracle(emu, dz, mode):
    import ctypes as C

    from newton_amd import _lib as L

    sc = two_box_scene(dz=dz, margin=0.005, mode=mode)
    want = oracle_contacts(sc)
    d, keep = _emu_sdf(emu, sc["sdfs"][0])
    table = (L.nt_sdf * 1)(d)
    cap = 256
    count, o_pair, o_key, o_data = np.zeros(1, np.int32), np.full(cap, -1, np.int32), np.zeros(cap, np.int32), np.zeros((cap, 9), np.float32)
    a = L.nt_mesh_sdf_args()
    a.pairs, a.pair_count = sc["pairs"].ctypes.data, 1
    a.shape_transform, a.shape_data, a.shape_gap = sc["X"].ctypes.data, sc["data"].ctypes.data, sc["gap"].ctypes.data
    a.shape_sdf_index, a.sdf_table, a.sdf_count = sc["sdf_index"].ctypes.data, C.addressof(table), 1
    a.shape_edge_range, a.edge_centers, a.edge_halves = sc["er"].ctypes.data, sc["ec"].ctypes.data, sc["eh"].ctypes.data
    a.out_count, a.out_pair, a.out_key, a.out_data, a.capacity = count.ctypes.data, o_pair.ctypes.data, o_key.ctypes.data, o_data.ctypes.data, cap
    assert emu.nt_mesh_sdf_collide(C.byref(a), None) == 0
    n = int(count[0])
    assert n == len(want) > 0
    order = np.lexsort((o_key[:n], o_pair[:n]))
    keys_want = sorted((p, k) for p, k, *_ in want)
    assert [(int(o_pair[i]), int(o_key[i])) for i in order] == keys_want  # the contact SET is bit-exact
    by_key = {(p, k): (c, nn, dd) for p, k, c, nn, dd, _, _ in want}
    for i in order:
        c, nn, dd = by_key[(int(o_pair[i]), int(o_key[i]))]
        assert np.array_equal(o_data[i, 0:3], c) and o_data[i, 6] == dd  # points and distances bit for bit
        assert np.max(np.abs(o_data[i, 3:6] - nn)) <= 2e-7             # normals: numpy's dot / cross sum in another order


# ------------------------------------------------------------------------------------------------ reduced contacts (a24)
def sphere_on_box_scene(dz=0.985, gap=0.03, margin=0.002, lat=14, lon=18):
    """A UV-sphere mesh (radius 0.5, ~700 edges, SDF of the sphere primitive) resting in a unit cube's top face (SDF of the box
    primitive), plus a second, tilted sphere beside it: two shape pairs, hundreds of unreduced contacts in the gap band."""
    sph, box = Mesh.create_sphere(0.5, lat, lon), Mesh.create_box(0.5, 0.5, 0.5)
    ec_s, eh_s = mesh_edge_tables(sph.vertices, sph.indices.reshape(-1, 3))
    ec_b, eh_b = mesh_edge_tables(box.vertices, box.indices.reshape(-1, 3))
    sdf_s = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.5, 0.5, 0.5), max_resolution=32)
    sdf_b = box_sdf()
    q = [0.0, np.sin(0.2), 0.0, np.cos(0.2)]
    X = np.array([[0, 0, 0, 0, 0, 0, 1], [0.03, -0.02, dz, 0, 0, 0, 1], [0.6, 0.1, 0.93, *q]], dtype=np.float32)
    data = np.array([[1, 1, 1, margin]] * 3, dtype=np.float32)
    gaps = np.full(3, gap, dtype=np.float32)
    er = np.array([[0, len(ec_b)], [len(ec_b), len(ec_s)], [len(ec_b), len(ec_s)]], dtype=np.int32)
    lo, hi, res = S.mesh_reduction_tables([box.vertices, sph.vertices, sph.vertices], [(1, 1, 1)] * 3)
    return dict(pairs=np.array([[0, 1], [0, 2], [1, 2]], dtype=np.int32), X=X, data=data, gap=gaps,
                sdf_index=np.array([0, 1, 1], dtype=np.int32), sdfs=[sdf_b, sdf_s], er=er, ec=np.concatenate([ec_b, ec_s]),
                eh=np.concatenate([eh_b, eh_s]), aabb_lo=lo, aabb_hi=hi, res=res)


def _emu_mesh_sdf(emu, sc, reduced, cap=8192, threads=0):
    import ctypes as C

    from newton_amd import _lib as L

    descs = [_emu_sdf(emu, t) for t in sc["sdfs"]]
    table = (L.nt_sdf * len(descs))(*[d for d, _ in descs])
    count, o_pair, o_key, o_data = np.zeros(1, np.int32), np.full(cap, -1, np.int32), np.zeros(cap, np.int32), np.zeros((cap, 9), np.float32)
    a = L.nt_mesh_sdf_args()
    a.pairs, a.pair_count = sc["pairs"].ctypes.data, len(sc["pairs"])
    a.shape_transform, a.shape_data, a.shape_gap = sc["X"].ctypes.data, sc["data"].ctypes.data, sc["gap"].ctypes.data
    a.shape_sdf_index, a.sdf_table, a.sdf_count = sc["sdf_index"].ctypes.data, C.addressof(table), len(descs)
    a.shape_edge_range, a.edge_centers, a.edge_halves = sc["er"].ctypes.data, sc["ec"].ctypes.data, sc["eh"].ctypes.data
    a.out_count, a.out_pair, a.out_key, a.out_data, a.capacity = count.ctypes.data, o_pair.ctypes.data, o_key.ctypes.data, o_data.ctypes.data, cap
    if reduced:
        r = L.nt_contact_reduce_shapes()
        r.shape_aabb_lower, r.shape_aabb_upper, r.shape_voxel_res = sc["aabb_lo"].ctypes.data, sc["aabb_hi"].ctypes.data, sc["res"].ctypes.data
        r.threads = threads
        assert emu.nt_mesh_sdf_collide_reduced(C.byref(a), C.byref(r), None) == 0
    else:
        assert emu.nt_mesh_sdf_collide(C.byref(a), None) == 0
    n = int(count[0])
    assert n <= cap
    return o_pair[:n].copy(), o_key[:n].copy(), o_data[:n].copy()


def check_reduced_against_unreduced(sc, unreduced, reduced):
    """The fused reduced kernel against the checker's reduction of the unreduced kernel's own rows: same survivors, points,
    distances and exported normals bit for bit; blocks contiguous per pair with ascending fingerprints."""
    import oracle_reduce as R

    u_pair, u_key, u_data = unreduced
    r_pair, r_key, r_data = reduced
    rows = [(int(u_pair[i]), int(u_key[i]), u_data[i, 0:3], u_data[i, 3:6], u_data[i, 6]) for i in range(len(u_key))]
    c = R.reduce_inputs_from_mesh_sdf_contacts(rows, sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["sdf_index"], sc["sdfs"],
                                               sc["aabb_lo"], sc["aabb_hi"], sc["res"])
    want = R.reduce_contacts(c)
    assert 0 < len(want["fp"]) < len(u_key)  # something was reduced away
    shape_pair = sc["pairs"][r_pair]
    o = np.lexsort((r_key, shape_pair[:, 1], shape_pair[:, 0]))
    assert np.array_equal(shape_pair[o], want["pair"]) and np.array_equal(r_key[o], want["fp"])
    assert np.array_equal(r_data[o, 0:3], want["pos"]) and np.array_equal(r_data[o, 6], want["depth"])
    assert np.array_equal(r_data[o, 3:6], want["normal"])
    for p in set(r_pair.tolist()):
        rows_p = np.flatnonzero(r_pair == p)
        assert np.array_equal(rows_p, np.arange(rows_p[0], rows_p[-1] + 1)) and np.all(np.diff(r_key[rows_p]) > 0)
    assert np.all(r_data[:, 7] == sc["data"][sc["pairs"][r_pair, 0], 3]) and np.all(r_data[:, 8] == sc["data"][sc["pairs"][r_pair, 1], 3])
    return len(u_key), len(r_key)


@pytest.mark.parametrize("threads", [0, 64, 128])
def test_emulated_reduced_mesh_sdf_kernel_is_the_reduction_of_the_unreduced_one(emu, threads):
    sc = sphere_on_box_scene()
    n_in, n_out = check_reduced_against_unreduced(sc, _emu_mesh_sdf(emu, sc, False), _emu_mesh_sdf(emu, sc, True, threads=threads))
    assert n_in > n_out + 50  # 167 unreduced contacts -> 87


def test_checker_chain_agrees_with_the_emulated_reduced_kernel(emu):
    """oracle_sdf.mesh_sdf_collide -> oracle_reduce: the float32 checker end to end (its normals differ from the kernel's in
    the last bit -- numpy sums dot / cross in another order -- so the survivor set is compared, then the geometry to 2e-7)."""
    import oracle_reduce as R

    sc = sphere_on_box_scene()
    rows = oracle_contacts(sc)
    c = R.reduce_inputs_from_mesh_sdf_contacts(rows, sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["sdf_index"], sc["sdfs"],
                                               sc["aabb_lo"], sc["aabb_hi"], sc["res"])
    want = R.reduce_contacts(c)
    r_pair, r_key, r_data = _emu_mesh_sdf(emu, sc, True)
    shape_pair = sc["pairs"][r_pair]
    o = np.lexsort((r_key, shape_pair[:, 1], shape_pair[:, 0]))
    assert np.array_equal(shape_pair[o], want["pair"]) and np.array_equal(r_key[o], want["fp"])
    assert np.array_equal(r_data[o, 0:3], want["pos"]) and np.array_equal(r_data[o, 6], want["depth"])
    assert np.abs(r_data[o, 3:6] - want["normal"]).max() <= 3e-7
