"""GPU parity on a chain with every supported joint type (revolute, prismatic, ball, fixed, D6 + FREE / fixed root, with
limits, drives, armature): XPBD, SolverSemiImplicit and SolverFeatherstone, HIP vs the CPU oracle, step by step."""
import numpy as np
import pytest

from test_gpu_parity_xpbd import _rel, _setup

pytestmark = pytest.mark.gpu


def _stepwise(solver_name, free_root, n_env, steps, dt, tol_q, tol_qd):
    from oracle_bridge import OracleState
    from scenes import joint_zoo_scene

    nt, model, o = _setup(joint_zoo_scene, n_env, free_root=free_root)
    rng = np.random.default_rng(9)
    jf = rng.normal(0, 0.5, size=model.joint_dof_count).astype(np.float32)
    s0, s1 = model.state(), model.state()
    ctrl = model.control()
    ctrl.joint_f = jf
    if solver_name == "xpbd":
        solver = nt.solvers.SolverXPBD(model, iterations=3)
    elif solver_name == "semi":
        solver = nt.solvers.SolverSemiImplicit(model)
    else:
        solver = nt.solvers.SolverFeatherstone(model)
    os0, os1 = OracleState(model), OracleState(model)
    c = o.control(joint_f=jf)
    for _ in range(steps):
        s0.body_q, s0.body_qd = os0.body_q, os0.body_qd
        s0.joint_q, s0.joint_qd = os0.joint_q, os0.joint_qd
        s0.clear_forces()
        solver.step(s0, s1, ctrl, None, dt)
        os0.body_f[:] = 0
        if solver_name == "xpbd":
            o.xpbd_step(os0, os1, c, None, dt, iterations=3)
        elif solver_name == "semi":
            o.semi_implicit_step(os0, os1, c, None, dt)
        else:
            o.featherstone_step(os0, os1, c, None, dt)
        assert np.all(np.isfinite(os1.body_q))
        assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= tol_q
        assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= tol_qd
        if solver_name == "fs":
            assert _rel(s1.joint_q.cpu().numpy(), os1.joint_q) <= tol_q
            assert _rel(s1.joint_qd.cpu().numpy(), os1.joint_qd) <= tol_qd
        os0, os1 = os1, os0


@pytest.mark.parametrize("free_root", [False, True])
def test_xpbd_joint_zoo(free_root):
    # XPBD velocities are position corrections / dt: 1e-6-level position differences (the documented world-frame
    # inverse-inertia formulation) show up as ~1e-3 in |qd| ~ 10 m/s
    _stepwise("xpbd", free_root, 11, 25, 1e-3, 1e-5, 1e-3)


@pytest.mark.parametrize("free_root", [False, True])
def test_semi_implicit_joint_zoo(free_root):
    # penalty joints (ke = 1e4) on 2 kg links need a small step to stay inside the explicit stability limit
    _stepwise("semi", free_root, 11, 25, 1e-4, 1e-5, 3e-4)


@pytest.mark.parametrize("free_root", [False, True])
def test_featherstone_joint_zoo(free_root):
    _stepwise("fs", free_root, 11, 25, 1e-3, 1e-5, 3e-4)
