"""body_qd convention known answers, restated from newton/tests/test_body_velocity.py:124-372,846-960: body_qd's linear part is
the COM velocity, so with an off-origin COM a pure spin leaves the COM where it is, a pure translation moves the COM by v t,
and both together superpose -- for XPBD and SemiImplicit (maximal coordinates, tolerance 1e-4) and Featherstone (FREE-joint
joint_qd, tolerance 1e-3).  Oracle on the CPU, HIP on the GPU."""
import numpy as np
import pytest

import newton_amd as nt
from newton_amd import _np_math as nm

I4 = [0.0, 0.0, 0.0, 1.0]
BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]
SOLVERS = {"xpbd": (False, 1e-4), "semi_implicit": (False, 1e-4), "featherstone": (True, 1e-3)}
COM_OFFSETS = [(0.5, 0.0, 0.0), (0.0, 0.3, 0.0), (0.0, 0.0, 0.4), (0.2, 0.3, 0.1)]
DT, STEPS = 0.01, 10


def _run(backend, solver, com_offset, velocity, initial_pos):
    b = nt.ModelBuilder(gravity=0.0)
    body = b.add_body(xform=[*initial_pos, *I4])
    b.add_shape_box(body, hx=0.1, hy=0.1, hz=0.1)
    b.body_com[body] = np.asarray(com_offset, dtype=np.float64)
    model = b.finalize(device="cuda:0" if backend == "hip" else None)
    generalized = SOLVERS[solver][0]
    velocity = np.asarray(velocity, dtype=np.float32)
    com = np.asarray(com_offset, dtype=np.float64)

    if backend == "oracle":
        from oracle_bridge import Oracle, OracleState

        o = Oracle(model)
        s0, s1 = OracleState(model), OracleState(model)
        if generalized:
            s0.joint_qd[:6] = velocity
            bq, bqd = o.eval_fk(s0.joint_q, s0.joint_qd)
            s0.body_q[:], s0.body_qd[:] = bq, bqd
        else:
            s0.body_qd[0] = velocity
        q_initial = s0.body_q[0].copy()
        for _ in range(STEPS):
            if solver == "xpbd":
                o.xpbd_step(s0, s1, o.control(), None, DT, angular_damping=0.0)
            elif solver == "semi_implicit":
                o.semi_implicit_step(s0, s1, o.control(), None, DT, angular_damping=0.0)
            else:
                o.featherstone_step(s0, s1, o.control(), None, DT)
            s0, s1 = s1, s0
        q_final = s0.body_q[0].copy()
    else:
        cls = {"xpbd": nt.solvers.SolverXPBD, "semi_implicit": nt.solvers.SolverSemiImplicit,
               "featherstone": nt.solvers.SolverFeatherstone}[solver]
        sol = cls(model) if solver == "featherstone" else cls(model, angular_damping=0.0)
        s0, s1 = model.state(), model.state()
        if generalized:
            s0.joint_qd = velocity
            nt.eval_fk(model, s0.joint_q, s0.joint_qd, s0)
        else:
            s0.body_qd = velocity.reshape(1, 6)
        q_initial = s0.body_q.cpu().numpy()[0].copy()
        for _ in range(STEPS):
            sol.step(s0, s1, None, None, DT)
            s0, s1 = s1, s0
        q_final = s0.body_q.cpu().numpy()[0].copy()
    com_initial = nm.transform_point(q_initial.astype(np.float64), com)
    com_final = nm.transform_point(q_final.astype(np.float64), com)
    return q_initial, q_final, com_initial, com_final


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("solver", list(SOLVERS))
def test_spin_keeps_the_com_stationary(oracle_lib, backend, solver):
    tol = SOLVERS[solver][1]
    for com_offset in COM_OFFSETS:
        for w in ((0.0, 0.0, 1.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)):
            q0, q1, c0, c1 = _run(backend, solver, com_offset, (0.0, 0.0, 0.0, *w), (1.0, 2.0, 3.0))
            assert np.linalg.norm(c1 - c0) < tol, (com_offset, w)
            assert abs(float(np.dot(q0[3:], q1[3:]))) < 0.9999  # it did rotate


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("solver", list(SOLVERS))
def test_translation_moves_the_com_by_v_t(oracle_lib, backend, solver):
    tol = SOLVERS[solver][1]
    for com_offset in COM_OFFSETS:
        for v in ((0.7, 0.0, 0.0), (0.0, 0.7, 0.0), (0.0, 0.0, 0.7)):
            _, _, c0, c1 = _run(backend, solver, com_offset, (*v, 0.0, 0.0, 0.0), (0.0, 0.0, 1.0))
            assert np.linalg.norm((c1 - c0) - np.asarray(v) * DT * STEPS) < tol, (com_offset, v)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("solver", list(SOLVERS))
def test_spin_and_translation_superpose(oracle_lib, backend, solver):
    tol = SOLVERS[solver][1]
    for com_offset in COM_OFFSETS:
        q0, q1, c0, c1 = _run(backend, solver, com_offset, (0.1, 0.0, 0.0, 0.0, 0.0, 1.0), (0.0, 0.0, 1.0))
        assert np.linalg.norm((c1 - c0) - np.array([0.1, 0.0, 0.0]) * DT * STEPS) < tol, com_offset
        assert abs(float(np.dot(q0[3:], q1[3:]))) < 0.9999
