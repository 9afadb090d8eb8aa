"""Hydroelastic contacts: the reference-held tables of newton/tests/test_hydroelastic.py
(triangle coverage :135-175, contact bands :236-249, corner numbering :197-203) on the oracle; the physics of the iso-pressure
patch (area, force = sum of stiffness * depth, normal) against the closed form for a sphere pressed into a slab; the generated
marching-cubes tables (closed surfaces, inward normals); and the gfx950 kernel (emulated here, on the device in
tests/test_gpu_sdf.py) against the oracle: identical face set, geometry within 1e-5."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from newton_amd import mc_tables  # noqa: E402
from newton_amd import sdf as S  # noqa: E402
from newton_amd.enums import GeoType  # noqa: E402


def sphere_on_slab(delta=0.005, res=24, kh=(1e6, 1e6), gap=0.002, margin=0.0):
    kw = dict(max_resolution=res, margin=0.02, narrow_band_range=(-0.03, 0.03), quantization_mode=S.QuantizationMode.FLOAT32)
    slab = S.create_texture_sdf_from_primitive(GeoType.BOX, (0.2, 0.2, 0.05), **kw)
    ball = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.1, 0.0, 0.0), **kw)
    X = np.array([[0, 0, 0, 0, 0, 0, 1], [0.01, 0.02, 0.05 + 0.1 - delta, 0, 0, 0, 1]], dtype=np.float32)
    return dict(pairs=np.array([[0, 1]], dtype=np.int32), X=X, data=np.array([[1, 1, 1, margin]] * 2, dtype=np.float32),
                gap=np.array([gap, gap], dtype=np.float32), kh=np.array(kh, dtype=np.float32), sdfs=[slab, ball])


def oracle_faces(sc):
    import oracle_hydro as H

    return H.hydro_collide(sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["kh"], sc["sdfs"], mc_tables.tables())


def test_reference_tables_triangle_fraction_bands_and_corners():
    import oracle_hydro as H

    d = np.array([[1, 2, 3], [-1, -2, -3], [-1, 2, 3], [2, -1, 3], [2, 3, -1], [1, -2, -3], [-2, 1, -3], [-2, -3, 1]], dtype=np.float32)
    got = [H.triangle_fraction(d[i], n) for i, n in enumerate([0, 3, 1, 1, 1, 2, 2, 2])]
    np.testing.assert_allclose(got, [0, 1, 1 / 12, 1 / 12, 1 / 12, 11 / 12, 11 / 12, 11 / 12], rtol=1e-6, atol=0)
    assert [H.classify(np.float32(x), np.float32(0.1)) for x in (-0.01, 0.0, 0.05, 0.1, 0.1001)] == [-1, 0, 0, 0, 1]
    assert [tuple(c[i] for c in H.CORNER) for i in range(8)] == [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1),
                                                                  (1, 1, 1), (0, 1, 1)]
    assert H.effective_stiffness(np.float32(2.0), np.float32(2.0)) == 1.0 and H.effective_stiffness(np.float32(0), np.float32(0)) == 0.0


def test_marching_cubes_tables_describe_closed_inward_oriented_surfaces():
    from collections import Counter

    r, f = mc_tables.tables()
    assert r[0] == 0 and r[1] == 0 and r[256] == r[255] and len(f) == r[256] and max(np.diff(r)) == 15  # <= 5 faces per voxel
    n = 8
    g = np.stack(np.meshgrid(*[np.arange(n + 1)] * 3, indexing="ij"), -1).astype(float)
    c, R = np.array([4.1, 3.9, 4.2]), 2.7
    vals = np.linalg.norm(g - c, axis=-1) - R
    tris = []
    for x in range(n):
        for y in range(n):
            for z in range(n):
                cv = [vals[x + q[0], y + q[1], z + q[2]] for q in mc_tables.CORNERS]
                case = sum(1 << i for i in range(8) if cv[i] < 0)
                for k in range(r[case], r[case + 1], 3):
                    tris.append([mc_tables.CORNERS[a] + (0 - cv[a]) / (cv[b] - cv[a]) * (mc_tables.CORNERS[b] - mc_tables.CORNERS[a]) +
                                 np.array([x, y, z]) for a, b in f[k:k + 3]])
    T = np.array(tris)
    edges = Counter()
    for tri in T:
        for a, b in ((0, 1), (1, 2), (2, 0)):
            edges[tuple(sorted((tuple(np.round(tri[a], 6)), tuple(np.round(tri[b], 6)))))] += 1
    assert set(edges.values()) == {2}  # watertight
    nrm, cen = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]), T.mean(axis=1)
    assert np.all(np.einsum("ij,ij->i", nrm, c - cen) > 0)  # normals point to the inside (value < 0) corners
    area = 0.5 * np.linalg.norm(nrm, axis=1).sum()
    assert abs(area - 4 * np.pi * R * R) / (4 * np.pi * R * R) < 0.05


def test_oracle_patch_of_a_sphere_pressed_into_a_slab():
    """Equal stiffness: the iso-pressure surface is the mid surface, a disc of radius^2 = 2 R delta carrying
    p(r) = kh (delta / 2)(1 - r^2 / a^2), so area = 2 pi R delta and force = kh pi R delta^2 / 2."""
    R, delta, kh = 0.1, 0.005, 1e6
    cs = oracle_faces(sphere_on_slab(delta=delta))
    pen = [c for c in cs if c[6] < 0]
    assert len(pen) > 20 and all(c[2] == 0 and c[3] == 1 for c in cs)  # the slab (coarser voxels) stays shape a
    area = sum(c[8] for c in pen)
    force = sum(c[7] * -c[6] for c in pen)  # what eval_body_contact applies: stiffness * depth
    assert abs(area - 2 * np.pi * R * delta) / (2 * np.pi * R * delta) < 0.15
    assert abs(force - kh * np.pi * R * delta ** 2 / 2) / (kh * np.pi * R * delta ** 2 / 2) < 0.15
    assert np.allclose(force, sum(c[8] * c[9] for c in pen), rtol=1e-5)  # == sum of area * pressure (Elandt et al.)
    n = np.sum([c[5] * c[8] * c[9] for c in pen], axis=0)
    assert np.allclose(n / np.linalg.norm(n), [0, 0, 1], atol=1e-3)  # a -> b: from the slab up into the ball
    spec = [c for c in cs if c[6] >= 0]
    assert spec and all(0 <= c[6] <= 0.004 + 1e-7 and abs(c[7] - 1e-2 * 5e5) < 1e-3 for c in spec)  # gap band: k_a k_b / (k_a + k_b)


def test_oracle_stiffness_ratio_moves_the_patch_and_gap_band_vanishes_without_gap():
    soft_ball = oracle_faces(sphere_on_slab(kh=(1e7, 1e5)))  # stiff slab: the patch hugs the slab's surface
    z = np.mean([c[4][2] for c in soft_ball if c[6] < 0])
    assert abs(z - 0.05) < 1.5e-3
    assert not [c for c in oracle_faces(sphere_on_slab(gap=0.0)) if c[6] > 0]


@pytest.fixture(scope="module")
def emu():
    import harness

    return harness.lib()


@pytest.mark.parametrize("kh,margin", [((1e6, 1e6), 0.0), ((3e6, 5e5), 0.001)])
def test_emulated_hydro_kernel_matches_the_oracle(emu, kh, margin):
    from test_sdf_contact import _emu_sdf

    from newton_amd import _lib as L

    sc = sphere_on_slab(res=16, kh=kh, margin=margin)
    want = oracle_faces(sc)
    descs = [_emu_sdf(emu, t) for t in sc["sdfs"]]
    table = (L.nt_sdf * 2)(descs[0][0], descs[1][0])
    tri_range, flat = mc_tables.tables()
    cap = 4096
    count, o_pair, o_key = np.zeros(1, np.int32), np.full(cap, -1, np.int32), np.zeros(cap, np.int32)
    o_shapes, o_data = np.zeros((cap, 2), np.int32), np.zeros((cap, 10), np.float32)
    idx = np.array([0, 1], dtype=np.int32)
    a = L.nt_hydro_args()
    a.pairs, a.pair_count = sc["pairs"].ctypes.data, 1
    a.shape_transform, a.shape_data, a.shape_gap, a.shape_kh = sc["X"].ctypes.data, sc["data"].ctypes.data, sc["gap"].ctypes.data, sc["kh"].ctypes.data
    a.shape_sdf_index, a.sdf_table, a.sdf_count = idx.ctypes.data, C.addressof(table), 2
    a.tri_range, a.flat_edge_verts, a.margin_contact_area, a.edge_clamp_min = tri_range.ctypes.data, flat.ctypes.data, 1e-2, 0.02
    a.out_count, a.out_pair, a.out_key, a.out_shapes, a.out_data, a.capacity = (count.ctypes.data, o_pair.ctypes.data, o_key.ctypes.data,
                                                                                 o_shapes.ctypes.data, o_data.ctypes.data, cap)
    assert emu.nt_hydro_collide(C.byref(a), None) == 0
    n = int(count[0])
    assert n == len(want) > 10
    order = np.lexsort((o_key[:n], o_pair[:n]))
    assert [(int(o_pair[i]), int(o_key[i])) for i in order] == [(p, k) for p, k, *_ in want]  # the face SET is exact
    for i, w in zip(order, want):
        assert (int(o_shapes[i, 0]), int(o_shapes[i, 1])) == (w[2], w[3])
        assert np.max(np.abs(o_data[i, 0:3] - w[4])) <= 1e-6 and np.max(np.abs(o_data[i, 3:6] - w[5])) <= 1e-5
        assert abs(o_data[i, 6] - w[6]) <= 1e-7 and abs(o_data[i, 7] - w[7]) <= 1e-4 * max(1.0, abs(w[7]))
        assert abs(o_data[i, 8] - w[8]) <= 1e-9 + 1e-5 * w[8] and abs(o_data[i, 9] - w[9]) <= 1e-3


def test_emulated_eval_body_contact_consumes_per_contact_stiffness(emu):
    """Contacts.rigid_contact_stiffness / _damping / _friction (what hydroelastic faces carry) override the shape materials in
    eval_body_contact (kernels_contact.py:452-459): emulated SemiImplicit kernel vs the oracle with random per-contact values,
    and the closed form: a sphere resting on per-contact springs sinks by m g / k."""
    import harness as H
    from oracle_bridge import Oracle, OracleState
    from scenes import mixed_primitive_scene

    model = mixed_primitive_scene(5)
    model.body_q[:, 2] -= 0.03  # press the bodies into the ground
    em = H.EmuModel(model)
    s0, s1, ct = H.EmuState(em), H.EmuState(em), H.EmuContacts(em)
    H.collide(em, s0, ct)
    t = model.env
    rng = np.random.default_rng(5)
    ns = t.np * t.cpp
    ct.prop = np.zeros((3, ns, t.env_stride), dtype=np.float32)
    ct.prop[0, :, : t.env_count] = rng.choice([0.0, 2.0e4, 7.5e3], size=(ns, t.env_count))
    ct.prop[1, :, : t.env_count] = rng.choice([0.0, 30.0], size=(ns, t.env_count))
    ct.prop[2, :, : t.env_count] = rng.choice([0.0, 0.5, 2.0], size=(ns, t.env_count))
    H.semi_implicit_step(em, s0, s1, H.EmuControl(em), ct, 1e-3)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    o.collide(os0.body_q, oc)
    # the oracle's flat arrays are in append order (env, pair, k): pick each contact's slot values
    e = ct.export()
    n = int(oc.count[0])
    assert int(e["count"][0]) == n > 0
    # export order: every env's analytic contacts in (env, slot) order, then every env's convex ones (nt_contacts_export)
    nas = t.np_analytic * t.cpp
    live = [(env, slot) for env in range(t.env_count) for slot in range(nas) if ct.shape0[slot, env] >= 0] + \
           [(env, slot) for env in range(t.env_count) for slot in range(nas, ns) if ct.shape0[slot, env] >= 0]
    assert len(live) == n and np.array_equal([ct.shape0[s_, env] for env, s_ in live], e["shape0"][:n])
    oc.set_properties([ct.prop[0, s, env] for env, s in live], [ct.prop[1, s, env] for env, s in live],
                      [ct.prop[2, s, env] for env, s in live])
    o.semi_implicit_step(os0, os1, o.control(), oc, 1e-3)
    assert np.max(np.abs(s1.aos("body_qd") - os1.body_qd)) <= 2e-4 * max(1.0, np.abs(os1.body_qd).max())
    base = H.EmuState(em)
    ct.prop = None
    H.semi_implicit_step(em, s0, base, H.EmuControl(em), ct, 1e-3)
    assert np.max(np.abs(base.aos("body_qd") - s1.aos("body_qd"))) > 1e-3  # the overrides really changed the forces
