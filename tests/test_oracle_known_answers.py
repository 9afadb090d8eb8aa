"""Pin the CPU oracle against the reference's own hand-computed known-answer tables
(newton/tests/test_collision_primitives.py:429-1528, extracted by tests/golden/make_primitive_known_answers.py).
Tolerances are the reference's: distances `places=5` (plane/sphere family) or +-0.01; contact counts exact."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from newton_amd import _np_math as nm
from newton_amd.enums import GeoType

HERE = os.path.dirname(os.path.abspath(__file__))
TABLES = json.load(open(os.path.join(HERE, "golden", "primitive_known_answers.json")))["tables"]
MAXVAL = 1e10


def _xf(pos, z_axis=None, mat=None):
    if mat is not None:
        R = np.asarray(mat, dtype=np.float64).reshape(3, 3)
        assert np.allclose(R, np.eye(3)), "tables only use identity rotations"
        q = nm.quat_identity()
    elif z_axis is not None:
        q = nm.quat_between_vectors([0.0, 0.0, 1.0], z_axis)
    else:
        q = nm.quat_identity()
    return np.array([*pos, *q], dtype=np.float32)


def _probe(lib, ta, tb, xa, xb, sa, sb, margin=0.0):
    f = C.POINTER(C.c_float)
    d4 = np.zeros(4, dtype=np.float32)
    p12 = np.zeros(12, dtype=np.float32)
    n3 = np.zeros(3, dtype=np.float32)
    sa = np.asarray(sa, dtype=np.float32)
    sb = np.asarray(sb, dtype=np.float32)
    cnt = lib.o_probe_primitive(int(ta), int(tb), xa.ctypes.data_as(f), xb.ctypes.data_as(f), sa.ctypes.data_as(f),
                                sb.ctypes.data_as(f), C.c_float(margin), d4.ctypes.data_as(f), p12.ctypes.data_as(f),
                                n3.ctypes.data_as(f))
    return cnt, d4, p12.reshape(4, 3), n3


def test_plane_sphere(oracle_lib):
    for n, pp, sp, r, want in TABLES["test_plane_sphere"]:
        cnt, d, p, nrm = _probe(oracle_lib, GeoType.PLANE, GeoType.SPHERE, _xf(pp, z_axis=n), _xf(sp), [0, 0, 0], [r, 0, 0])
        assert cnt == 1 and abs(d[0] - want) < 1e-5
        assert np.allclose(nrm, n, atol=1e-6)
        if want < 0:  # contact position between sphere centre and plane (reference's geometric checks)
            assert np.linalg.norm(p[0] - sp) < r + 0.01
            assert np.dot(p[0] - pp, n) <= np.dot(np.asarray(sp) - pp, n) + 0.01


def test_sphere_sphere(oracle_lib):
    for p1, r1, p2, r2, want in TABLES["test_sphere_sphere"]:
        cnt, d, _, nrm = _probe(oracle_lib, GeoType.SPHERE, GeoType.SPHERE, _xf(p1), _xf(p2), [r1, 0, 0], [r2, 0, 0])
        assert cnt == 1 and abs(d[0] - want) < 1e-5
        assert abs(np.linalg.norm(nrm) - 1.0) < 1e-5


def test_sphere_capsule(oracle_lib):
    for sp, sr, cp, axis, cr, ch, want in TABLES["test_sphere_capsule"]:
        cnt, d, _, _ = _probe(oracle_lib, GeoType.SPHERE, GeoType.CAPSULE, _xf(sp), _xf(cp, z_axis=axis), [sr, 0, 0], [cr, ch, 0])
        assert cnt == 1 and abs(d[0] - want) < 1e-4


def test_capsule_capsule(oracle_lib):
    for p1, a1, r1, h1, p2, a2, r2, h2, want in TABLES["test_capsule_capsule"]:
        cnt, d, _, _ = _probe(oracle_lib, GeoType.CAPSULE, GeoType.CAPSULE, _xf(p1, z_axis=a1), _xf(p2, z_axis=a2),
                              [r1, h1, 0], [r2, h2, 0])
        assert cnt >= 1 and abs(d[0] - want) < 1e-4


def test_plane_ellipsoid(oracle_lib):
    for n, pp, ep, rot, size, want in TABLES["test_plane_ellipsoid"]:
        cnt, d, _, _ = _probe(oracle_lib, GeoType.PLANE, GeoType.ELLIPSOID, _xf(pp, z_axis=n), _xf(ep, mat=rot), [0, 0, 0], size)
        assert cnt == 1 and abs(d[0] - want) < 1e-4


def test_sphere_cylinder(oracle_lib):
    for sp, sr, cp, axis, cr, ch, want in TABLES["test_sphere_cylinder"]:
        cnt, d, _, _ = _probe(oracle_lib, GeoType.SPHERE, GeoType.CYLINDER, _xf(sp), _xf(cp, z_axis=axis), [sr, 0, 0], [cr, ch, 0])
        assert cnt == 1 and abs(d[0] - want) < 1e-4


def test_sphere_box(oracle_lib):
    for sp, sr, bp, rot, size, want in TABLES["test_sphere_box"]:
        cnt, d, _, _ = _probe(oracle_lib, GeoType.SPHERE, GeoType.BOX, _xf(sp), _xf(bp, mat=rot), [sr, 0, 0], size)
        assert cnt == 1 and abs(d[0] - want) < 1e-4


def test_plane_capsule(oracle_lib):
    for n, pp, cp, axis, r, h, want in TABLES["test_plane_capsule"]:
        cnt, d, _, _ = _probe(oracle_lib, GeoType.PLANE, GeoType.CAPSULE, _xf(pp, z_axis=n), _xf(cp, z_axis=axis), [0, 0, 0], [r, h, 0])
        assert cnt == 2 and abs(min(d[0], d[1]) - want) < 1e-4


def test_plane_box(oracle_lib):
    for n, pp, bp, rot, size, want_count, want in TABLES["test_plane_box"]:
        cnt, d, _, nrm = _probe(oracle_lib, GeoType.PLANE, GeoType.BOX, _xf(pp, z_axis=n), _xf(bp, mat=rot), [0, 0, 0], size, 0.0)
        assert cnt == want_count
        for j in range(4):
            if d[j] < MAXVAL * 0.99:
                assert abs(d[j] - want) < 0.01
        if want_count:
            assert np.dot(nrm, n) > 0.99


def test_plane_cylinder(oracle_lib):
    for n, pp, cp, axis, r, h, want in TABLES["test_plane_cylinder"]:
        cnt, d, _, nrm = _probe(oracle_lib, GeoType.PLANE, GeoType.CYLINDER, _xf(pp, z_axis=n), _xf(cp, z_axis=axis), [0, 0, 0], [r, h, 0])
        valid = d[d < MAXVAL * 0.99]
        if want <= 0.0:
            assert len(valid) > 0
        if len(valid):
            assert abs(valid.min() - want) < 0.01
        if want <= 0.0:
            assert np.dot(nrm, n) > 0.99
