"""Pin the oracle's convex path (MPR -> GJK -> manifold) against the reference's own tests:
newton/tests/test_gjk.py:160-235, test_mpr.py:175-197, test_narrow_phase.py:707-800,1677-1960,2739-3150 and the
aligned box-stack invariant of test_solver_xpbd.py:1791-1840.  Tolerances are the reference's."""
import ctypes as C

import numpy as np
import pytest

import newton_amd as nt
from newton_amd.enums import GeoType
from oracle_bridge import Oracle, OracleState
from pair_scenes import CONVEX_CASES, decode_world, pair_model

F = C.POINTER(C.c_float)


def _probe(fn, ta, sa, xa, tb, sb, xb):
    out = np.zeros(8, dtype=np.float32)
    xa, xb = np.asarray(xa, dtype=np.float32), np.asarray(xb, dtype=np.float32)
    sa, sb = np.asarray(sa, dtype=np.float32), np.asarray(sb, dtype=np.float32)
    fn(int(ta), int(tb), xa.ctypes.data_as(F), xb.ctypes.data_as(F), sa.ctypes.data_as(F), sb.ctypes.data_as(F),
       out.ctypes.data_as(F))
    return bool(out[0]), float(out[1]), out[2:5].copy(), out[5:8].copy()


def test_gjk_known_answers(oracle_lib):
    g = oracle_lib.o_probe_gjk
    I = [0, 0, 0, 1]
    col, dist, _, _ = _probe(g, GeoType.SPHERE, [1, 0, 0], [-1.5, 0, 0, *I], GeoType.SPHERE, [1, 0, 0], [1.5, 0, 0, *I])
    assert abs(dist - 1.0) < 1e-5 and not col
    col, dist, _, _ = _probe(g, GeoType.SPHERE, [1, 0, 0], [-1, 0, 0, *I], GeoType.SPHERE, [1, 0, 0], [1, 0, 0, *I])
    assert abs(dist) < 1e-5
    col, dist, _, _ = _probe(g, GeoType.SPHERE, [3, 0, 0], [-1, 0, 0, *I], GeoType.SPHERE, [3, 0, 0], [3, 0, 0, *I])
    assert abs(dist) < 1e-5 and col
    col, dist, _, n = _probe(g, GeoType.BOX, [1, 1, 1], [-2, 0, 0, *I], GeoType.BOX, [1, 1, 1], [2.5, 0, 0, *I])
    assert abs(dist - 2.5) < 1e-5 and not col
    assert np.allclose(n, [1, 0, 0], atol=1e-5)


def test_mpr_box_support_tie_boundary(oracle_lib):
    """test_mpr.py:175-197: witnesses stay valid just below / above the centred-tie threshold."""
    for k in (0.5, 2.0):
        ang = np.float32(1.0e-6 * k)
        q = [np.sin(0.5 * ang), 0, 0, np.cos(0.5 * ang)]
        col, sd, _, n = _probe(oracle_lib.o_probe_mpr, GeoType.BOX, [0.5] * 3, [0, 0, 0, 0, 0, 0, 1], GeoType.BOX, [0.5] * 3,
                               [0, 0.999, 0, *q])
        expected = 0.5 + 0.5 * (np.cos(ang) + np.sin(ang)) - 0.999
        assert col
        assert abs(np.linalg.norm(n) - 1.0) < 1e-6
        assert np.allclose(n, [0, 1, 0], atol=1e-5)
        assert abs(-sd - expected) < 2e-5


def _contacts(name):
    model = pair_model(CONVEX_CASES[name])
    o = Oracle(model)
    ct = o.contacts()
    o.collide(model.body_q, ct)
    n = int(ct.count[0])
    return model, decode_world(model, model.body_q, ct.shape0[:n], ct.shape1[:n], ct.point0[:n], ct.point1[:n], ct.normal[:n],
                               ct.margin0[:n], ct.margin1[:n])


def _sd_box(p, center, half):
    q = np.abs(np.asarray(p) - center) - half
    return np.linalg.norm(np.maximum(q, 0.0)) + min(q.max(), 0.0)


def test_box_box_face(oracle_lib):
    _, cs = _contacts("box_box_face")
    assert len(cs) == 4
    for c, n, d in cs:
        assert abs(np.linalg.norm(n) - 1.0) < 1e-5 and n[0] > 0.9
        assert abs(d + 0.2) < 1e-4
        assert abs(_sd_box(c - n * d / 2, [0, 0, 0], 1.0)) < 5e-5 and abs(_sd_box(c + n * d / 2, [1.8, 0, 0], 1.0)) < 5e-5


def test_box_box_edge(oracle_lib):
    _, cs = _contacts("box_box_edge")
    assert len(cs) > 0
    assert abs(np.linalg.norm(cs[0][1]) - 1.0) < 1e-5
    assert abs(min(d for _, _, d in cs) - (1.2 - 0.5 - np.sqrt(0.5))) < 1e-4


@pytest.mark.parametrize("name,expected", [("box_box_overlap_0p01", -0.01), ("box_box_touching", 0.0),
                                           ("box_box_small_thickness", -(0.01 + 5e-5)), ("box_box_large_thickness", -0.02)])
def test_box_box_penetration_accuracy(oracle_lib, name, expected):
    """test_narrow_phase.py:2739-2807,2869-2951: deepest penetration within 5e-5 for each `enlarge` branch."""
    _, cs = _contacts(name)
    assert len(cs) > 0
    assert abs(min(d for _, _, d in cs) - expected) < 5e-5


def test_box_box_contact_point_on_surface(oracle_lib):
    _, cs = _contacts("box_box_overlap_0p05")
    ok = 0
    for c, n, d in cs:
        if d >= 0:
            continue
        assert abs(_sd_box(c - n * d / 2, [0, 0, 0], 0.5)) < 5e-5
        assert abs(_sd_box(c + n * d / 2, [0, 0, 0.95], 0.5)) < 5e-5
        ok += 1
    assert ok > 0


def test_ellipsoid_family(oracle_lib):
    _, cs = _contacts("ell_ell_separated")
    assert len(cs) == 0 or cs[0][2] > 0.0
    _, cs = _contacts("ell_ell_penetrating")
    assert len(cs) == 1 and cs[0][2] < 0 and abs(np.linalg.norm(cs[0][1]) - 1) < 1e-5 and cs[0][1][0] > 0
    assert abs(cs[0][2] + 0.2) < 1e-3
    # type sorting puts the sphere first (SPHERE < ELLIPSOID): normal points sphere -> ellipsoid = -x
    _, cs = _contacts("ell_sphere")
    assert len(cs) == 1 and cs[0][2] < 0 and cs[0][1][0] < -0.9 and abs(cs[0][2] + 0.1) < 1e-3
    _, cs = _contacts("ell_box")
    assert len(cs) == 1 and cs[0][1][0] > 0.9 and abs(cs[0][2] + 0.1) < 1e-3
    _, cs = _contacts("ell_capsule")
    assert len(cs) == 1 and abs(np.linalg.norm(cs[0][1]) - 1) < 1e-5
    _, cs = _contacts("ell_ell_spherelike")
    assert len(cs) == 1 and abs(cs[0][2] + 0.2) < 1e-3 and cs[0][1][0] > 0.99


def test_axial_shapes(oracle_lib):
    """Cylinder / cone / capsule manifolds: counts and depths from plain geometry."""
    _, cs = _contacts("capsule_box")       # capsule lying on the box top: 2 end contacts, depth 0.05
    assert len(cs) >= 2
    assert abs(min(d for _, _, d in cs) + 0.05) < 2e-4
    _, cs = _contacts("cylinder_box_flat")  # cap face on box top: depth 0.02
    assert len(cs) >= 3
    assert all(abs(d + 0.02) < 2e-4 for _, _, d in cs)
    _, cs = _contacts("cylinder_box_rolling")  # cylinder on its side: line contact, depth 0.01
    assert len(cs) >= 2
    assert abs(min(d for _, _, d in cs) + 0.01) < 2e-4
    for c, n, d in cs:  # rolling projection keeps contacts in the plane through the axis
        assert abs(c[0]) < 1e-4
    _, cs = _contacts("cone_box")  # base on the box top, depth 0.01
    assert len(cs) >= 3 and all(abs(d + 0.01) < 2e-4 for _, _, d in cs)
    _, cs = _contacts("sphere_cone")
    assert len(cs) == 1
    _, cs = _contacts("cylinder_cylinder")
    assert len(cs) >= 3 and all(abs(d + 0.02) < 2e-4 for _, _, d in cs)


def test_aligned_box_stack_remains_stable(oracle_lib):
    """test_solver_xpbd.py:1791-1840 (z-up restatement): five aligned boxes stay upright for 180 frames."""
    b = nt.ModelBuilder()
    for k in range(5):
        body = b.add_body(xform=[0, 0, 0.5 + k, 0, 0, 0, 1])
        b.add_shape_box(body, hx=0.5, hy=0.5, hz=0.5)
    b.add_ground_plane()
    m = b.finalize()
    o = Oracle(m)
    ct, c = o.contacts(), o.control()
    s0, s1 = OracleState(m), OracleState(m)
    for _ in range(180 * 4):
        s0.body_f[:] = 0
        o.collide(s0.body_q, ct)
        o.xpbd_step(s0, s1, c, ct, 1.0 / 240.0, iterations=4)
        s0, s1 = s1, s0
    q = s0.body_q
    assert np.all(np.isfinite(q))
    assert np.allclose(q[:, 2], 0.5 + np.arange(5), atol=2e-2)
    assert np.max(np.linalg.norm(q[:, :2], axis=1)) < 1e-2
    assert np.max(np.linalg.norm(q[:, 3:5], axis=1)) < 1e-3
