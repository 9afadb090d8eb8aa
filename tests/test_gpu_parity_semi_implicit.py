"""GPU parity: HIP SolverSemiImplicit (through the C ABI) vs the CPU oracle (config C1 + a contact-rich scene)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b, floor=1.0):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def test_pendulum_single_step_and_rollout():
    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState
    from scenes import pendulum_scene

    model = pendulum_scene(37, device="cuda:0", seed=11)
    o = Oracle(model)
    solver = nt.solvers.SolverSemiImplicit(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s0, s1 = model.state(), model.state()
    os0, os1 = OracleState(model), OracleState(model)
    oc, c = o.contacts(), o.control()
    dt = 1e-3
    for step in range(100):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, dt)
        s0, s1 = s1, s0
        os0.body_f[:] = 0
        o.collide(os0.body_q, oc)
        o.semi_implicit_step(os0, os1, c, oc, dt)
        os0, os1 = os1, os0
        if step == 0:
            assert _rel(s0.body_q.cpu().numpy(), os0.body_q) <= 1e-5
            assert _rel(s0.body_qd.cpu().numpy(), os0.body_qd) <= 1e-5
    assert _rel(s0.body_q.cpu().numpy(), os0.body_q) <= 1e-4
    assert _rel(s0.body_qd.cpu().numpy(), os0.body_qd, floor=1.0) <= 1e-3


def test_quadruped_semi_implicit_with_contacts():
    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState
    from scenes import quadruped_scene

    model = quadruped_scene(21, device="cuda:0")
    E = model.world_count
    model.joint_q.reshape(E, -1)[:, 2] -= 0.26
    bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    model.body_q, model.body_qd = bq, bqd
    rng = np.random.default_rng(3)
    model.body_qd = (model.body_qd + rng.normal(0, 0.3, size=model.body_qd.shape)).astype(np.float32)
    o = Oracle(model)
    solver = nt.solvers.SolverSemiImplicit(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s0, s1 = model.state(), model.state()
    ctrl = model.control()
    jf = rng.normal(0, 1.0, size=model.joint_dof_count).astype(np.float32)
    ctrl.joint_f = jf
    pipe.collide(s0, contacts)
    solver.step(s0, s1, ctrl, contacts, 1e-4)
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    o.collide(os0.body_q, oc)
    assert oc.count[0] > 0
    o.semi_implicit_step(os0, os1, o.control(joint_f=jf), oc, 1e-4)
    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd, floor=1.0) <= 1e-4
