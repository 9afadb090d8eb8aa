"""XPBD joint projection invariants restated from newton/tests/test_solver_xpbd.py:151-330 (zero gravity, one step):
distance-joint bounds from separated and coincident anchors, ball- and prismatic-joint recovery from large anchor
separations.  Oracle on the CPU, HIP path on the GPU."""
import numpy as np
import pytest

import newton_amd as nt
from newton_amd import _np_math as nm

I4 = [0.0, 0.0, 0.0, 1.0]


def _step(model, backend, body_q, iterations, dt):
    if backend == "oracle":
        from oracle_bridge import Oracle, OracleState

        o = Oracle(model)
        s0, s1 = OracleState(model, body_q=body_q), OracleState(model)
        o.xpbd_step(s0, s1, o.control(), None, dt, iterations=iterations)
        return s1.body_q
    s0, s1 = model.state(), model.state()
    s0.body_q = body_q
    nt.solvers.SolverXPBD(model, iterations=iterations).step(s0, s1, None, None, dt)
    return s1.body_q.cpu().numpy()


def _distance(backend, device, initial_distance, min_distance, max_distance):
    b = nt.ModelBuilder(gravity=0.0)
    body = b.add_link(xform=[initial_distance, 0.0, 0.0, *I4])
    b.add_shape_sphere(body, radius=0.1)
    j = b.add_joint_distance(-1, body, parent_xform=[0, 0, 0, *nm.quat_from_axis_angle([0.0, 0.0, 1.0], np.pi * 0.5)],
                             min_distance=min_distance, max_distance=max_distance)
    b.add_articulation([j])
    model = b.finalize(device=device)
    return _step(model, backend, model.body_q, 10, 1.0 / 60.0)[0, :3]


def _check_distance(backend, device=None):
    assert np.linalg.norm(_distance(backend, device, 0.25, 1.0, -1.0)) >= 0.99
    assert np.allclose(_distance(backend, device, 0.0, 1.0, -1.0), (0.0, 1.0, 0.0), atol=0.01)
    assert np.linalg.norm(_distance(backend, device, 2.0, -1.0, 1.0)) <= 1.01


def _capsule_pair(kind, device):
    r, hh = 0.0625, 0.25
    ext = r + hh
    b = nt.ModelBuilder(gravity=0.0)
    shape_xf = [0, 0, 0, *nm.quat_from_axis_angle([0.0, 1.0, 0.0], 0.5 * np.pi)]
    parent = b.add_link()
    b.add_shape_capsule(parent, xform=shape_xf, radius=r, half_height=hh)
    child = b.add_link(xform=[2.0 * ext, 0.0, 0.0, *I4] if kind == "ball" else [0.0, 0.0, 0.0, *I4])
    b.add_shape_capsule(child, xform=shape_xf, radius=r, half_height=hh)
    root = b.add_joint_free(parent)
    if kind == "ball":
        j = b.add_joint_ball(parent, child, parent_xform=[ext, 0, 0, *I4], child_xform=[-ext, 0, 0, *I4])
    else:
        j = b.add_joint_prismatic(parent, child, axis=(1, 0, 0), limit_lower=-2.0, limit_upper=2.0)
    b.add_articulation([root, j])
    return b.finalize(device=device), parent, child, ext


def _check_ball(backend, device=None):
    model, parent, child, ext = _capsule_pair("ball", device)
    bq = np.array(model.body_q, dtype=np.float32).copy()
    bq[child, :3] += np.array((1.0, 1.0, 0.0), dtype=np.float32)
    out = _step(model, backend, bq, 2, 1.0 / 240.0)
    gap = np.linalg.norm(nm.transform_point(out[child], [-ext, 0, 0]) - nm.transform_point(out[parent], [ext, 0, 0]))
    assert gap < 0.5


def _check_prismatic(backend, device=None):
    model, parent, child, _ = _capsule_pair("prismatic", device)
    bq = np.array(model.body_q, dtype=np.float32).copy()
    bq[child, :3] += np.array((0.5, 1.0, 1.0), dtype=np.float32)
    out = _step(model, backend, bq, 2, 1.0 / 240.0)
    rel = nm.transform_point(nm.transform_inverse(out[parent]), out[child, :3])
    assert np.hypot(rel[1], rel[2]) < 0.5
    assert -2.0 <= rel[0] <= 2.0


CHECKS = {"distance": _check_distance, "ball": _check_ball, "prismatic": _check_prismatic}


@pytest.mark.parametrize("name", sorted(CHECKS))
def test_joint_recovery_oracle(oracle_lib, name):
    CHECKS[name]("oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CHECKS))
def test_joint_recovery_hip(name):
    CHECKS[name]("hip", device="cuda:0")
