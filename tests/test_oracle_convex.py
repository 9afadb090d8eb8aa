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


import convex_known_answers as KA


def _cs(name):
    return _contacts(name)[1]


@pytest.mark.parametrize("check", KA.CHECKS, ids=lambda f: f.__name__)
def test_convex_known_answers(oracle_lib, check):
    check(_cs)


@pytest.mark.parametrize("name,expected", KA.PENETRATION_CASES)
def test_box_box_penetration_accuracy(oracle_lib, name, expected):
    KA.check_box_box_penetration_accuracy(_cs, name, expected)


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
