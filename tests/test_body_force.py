"""Wrench application on a floating body, restated from newton/tests/test_body_force.py:31-103 (zero gravity, rotated
body, 1000 N / N m for one 0.1 s step): the expected velocity is F/m * dt (or tau/I * dt) within 5 %, all other twist
components stay below 1e-3 -- for every solver, through body_f and through Control.joint_f, on the oracle (CPU) and on the
HIP path (GPU)."""
import numpy as np
import pytest

import newton_amd as nt
from newton_amd import _np_math as nm

SOLVERS = ["xpbd", "semi_implicit", "featherstone"]


def _model(device=None, worlds=0):
    env = nt.ModelBuilder(up_axis=1, gravity=0.0)
    rot = nm.quat_from_axis_angle([1.0, 0.0, 0.0], np.pi * 0.5)
    body = env.add_body(xform=[1.0, 2.0, 3.0, *rot])
    env.add_shape_box(body, hx=0.5, hy=0.5, hz=0.5)
    if worlds == 0:
        return env.finalize(device=device)
    scene = nt.ModelBuilder(up_axis=1, gravity=0.0)
    scene.replicate(env, worlds)
    return scene.finalize(device=device)


def _expected(model, index, magnitude, dt):
    if index >= 3:
        inertia = np.asarray(model.body_inertia[0]).reshape(3, 3)
        assert abs(inertia[0, 0] - inertia[1, 1]) < 1e-3 and abs(inertia[1, 1] - inertia[2, 2]) < 1e-3
        return magnitude / inertia[2, 2] * dt
    return magnitude / float(model.body_mass[0]) * dt


def _check(qd, index, expected):
    assert abs(qd[index] - expected) < 5e-2 * abs(expected)
    for i in range(6):
        if i != index:
            assert abs(qd[i]) < 1e-3


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("index", [1, 5])
@pytest.mark.parametrize("use_control", [False, True])
def test_floating_body_oracle(oracle_lib, solver, index, use_control):
    from oracle_bridge import Oracle, OracleState

    model = _model()
    o = Oracle(model)
    wrench = np.zeros(6, dtype=np.float32)
    wrench[index] = 1000.0
    s0, s1 = OracleState(model, body_f=None if use_control else wrench.reshape(1, 6)), OracleState(model)
    c = o.control(joint_f=wrench if use_control else None)
    dt = 0.1
    if solver == "xpbd":
        o.xpbd_step(s0, s1, c, None, dt)
    elif solver == "semi_implicit":
        o.semi_implicit_step(s0, s1, c, None, dt, angular_damping=0.0)
    else:
        o.featherstone_step(s0, s1, c, None, dt)
    _check(s1.body_qd[0], index, _expected(model, index, 1000.0, dt))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("index", [1, 5])
@pytest.mark.parametrize("use_control", [False, True])
def test_floating_body_hip(solver, index, use_control):
    model = _model(device="cuda:0", worlds=5)
    wrench = np.zeros(6, dtype=np.float32)
    wrench[index] = 1000.0
    s0, s1 = model.state(), model.state()
    ctrl = model.control()
    if use_control:
        ctrl.joint_f = np.tile(wrench, 5)
    else:
        s0.body_f = np.tile(wrench, (5, 1))
    if solver == "xpbd":
        sol = nt.solvers.SolverXPBD(model)
    elif solver == "semi_implicit":
        sol = nt.solvers.SolverSemiImplicit(model, angular_damping=0.0)
    else:
        sol = nt.solvers.SolverFeatherstone(model)
    sol.step(s0, s1, ctrl, None, 0.1)
    qd = s1.body_qd.cpu().numpy()
    for e in range(5):
        _check(qd[e], index, _expected(model, index, 1000.0, 0.1))
