"""PD joint drives on a single revolute joint, restated from newton/tests/test_joint_controllers.py:39-107,640-664
(zero gravity, unit inertia, 100 steps of 1/60 s): the position drive (ke 2000, kd 500) settles at the target pi/2, the
velocity drive (kd 500) reaches pi/2 rad/s, both within 1e-2 -- SolverFeatherstone and SolverXPBD(iterations=5), on the
oracle (CPU) and on the HIP path (GPU)."""
import numpy as np
import pytest

import newton_amd as nt

CASES = [("position", np.pi / 2.0, 0.0, np.pi / 2.0, 0.0, 2000.0, 500.0), ("velocity", 0.0, np.pi / 2.0, None, np.pi / 2.0, 0.0, 500.0)]


def _model(pos_target, vel_target, ke, kd, device=None, worlds=0):
    env = nt.ModelBuilder(up_axis=1, gravity=0.0)
    b = env.add_link(inertia=np.eye(3), mass=1.0)
    env.add_shape_box(b, hx=0.2, hy=0.2, hz=0.2, cfg=nt.ModelBuilder.ShapeConfig(density=1.0))
    j = env.add_joint_revolute(-1, b, parent_xform=[0.0, 2.0, 0.0, 0.0, 0.0, 0.0, 1.0], child_xform=[0.0, 2.0, 0.0, 0.0, 0.0, 0.0, 1.0],
                               axis=[0.0, 0.0, 1.0], target_pos=pos_target, target_vel=vel_target, armature=0.0, limit_ke=0.0,
                               limit_kd=0.0, target_ke=ke, target_kd=kd)
    env.add_articulation([j])
    if worlds == 0:
        return env.finalize(device=device)
    scene = nt.ModelBuilder(up_axis=1, gravity=0.0)
    scene.replicate(env, worlds)
    return scene.finalize(device=device)


def _angle_and_rate(body_q, body_qd):
    """Joint angle / rate of the z-revolute recovered from the maximal state (what eval_ik returns for this joint)."""
    return 2.0 * np.arctan2(body_q[5], body_q[6]), body_qd[5]


@pytest.mark.parametrize("solver", ["featherstone", "xpbd"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_revolute_controller_oracle(oracle_lib, solver, case):
    from oracle_bridge import Oracle, OracleState

    _, pos_t, vel_t, exp_pos, exp_vel, ke, kd = case
    model = _model(pos_t, vel_t, ke, kd)
    o = Oracle(model)
    c = o.control(joint_target_q=np.array([pos_t], dtype=np.float32), joint_target_qd=np.array([vel_t], dtype=np.float32))
    s0, s1 = OracleState(model), OracleState(model)
    for _ in range(100):
        s0.body_f[:] = 0
        if solver == "xpbd":
            o.xpbd_step(s0, s1, c, None, 1.0 / 60.0, iterations=5)
        else:
            o.featherstone_step(s0, s1, c, None, 1.0 / 60.0)
        s0, s1 = s1, s0
    if solver == "featherstone":
        q, qd = float(s0.joint_q[0]), float(s0.joint_qd[0])
    else:
        q, qd = _angle_and_rate(s0.body_q[0], s0.body_qd[0])
    if exp_pos is not None:
        assert abs(q - exp_pos) < 1e-2
    if exp_vel is not None:
        assert abs(qd - exp_vel) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["featherstone", "xpbd"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_revolute_controller_hip(solver, case):
    _, pos_t, vel_t, exp_pos, exp_vel, ke, kd = case
    model = _model(pos_t, vel_t, ke, kd, device="cuda:0", worlds=3)
    ctrl = model.control()
    ctrl.joint_target_q = np.full(3, pos_t, dtype=np.float32)
    ctrl.joint_target_qd = np.full(3, vel_t, dtype=np.float32)
    sol = nt.solvers.SolverXPBD(model, iterations=5) if solver == "xpbd" else nt.solvers.SolverFeatherstone(model, angular_damping=0.0)
    s0, s1 = model.state(), model.state()
    for _ in range(100):
        s0.clear_forces()
        sol.step(s0, s1, ctrl, None, 1.0 / 60.0)
        s0, s1 = s1, s0
    bq, bqd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
    for e in range(3):
        if solver == "featherstone":
            q, qd = float(s0.joint_q.cpu().numpy()[e]), float(s0.joint_qd.cpu().numpy()[e])
        else:
            q, qd = _angle_and_rate(bq[e], bqd[e])
        if exp_pos is not None:
            assert abs(q - exp_pos) < 1e-2
        if exp_vel is not None:
            assert abs(qd - exp_vel) < 1e-2


def test_semi_implicit_d6_three_angular_axes_pd_targets(oracle_lib):
    """D6 joint with three driven angular axes under SolverSemiImplicit (kernels_body.py:371-515): the relative rotation is
    decomposed into the intrinsic X-Y'-Z'' angles, each axis gets its PD drive, and the body settles at the target angles."""
    from scipy.spatial.transform import Rotation

    import newton_amd as nt
    from newton_amd import _np_math as nm
    from oracle_bridge import Oracle, OracleState

    D = nt.ModelBuilder.JointDofConfig
    targets = [0.3, -0.2, 0.4]
    b = nt.ModelBuilder(gravity=0.0)
    link = b.add_link()
    b.add_shape_box(link, hx=0.1, hy=0.15, hz=0.2)
    j = b.add_joint_d6(-1, link, linear_axes=[], angular_axes=[D(axis=a, target_ke=50.0, target_kd=5.0) for a in range(3)],
                       parent_xform=[0.0, 0.0, 1.0, *nm.quat_rpy(0.2, 0.1, -0.3)])
    b.add_articulation([j])
    model = b.finalize()
    o = Oracle(model)
    s0, s1 = OracleState(model), OracleState(model)
    ctrl = o.control(joint_target_q=np.array(targets, dtype=np.float32))
    for _ in range(30000):
        s0.body_f[:] = 0
        o.semi_implicit_step(s0, s1, ctrl, None, 1e-4, angular_damping=0.0)
        s0, s1 = s1, s0
    X_wp = np.asarray(model.joint_X_p)[0].astype(np.float64)
    rel = nm.quat_mul(nm.quat_inverse(X_wp[3:]), s0.body_q[0, 3:].astype(np.float64))
    assert np.allclose(Rotation.from_quat(rel).as_euler("XYZ"), targets, atol=1e-3)
    assert np.abs(s0.body_qd[0, 3:]).max() < 1e-2
