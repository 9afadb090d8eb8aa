"""GPU parity: SolverFeatherstone (one gfx950 kernel through nt_featherstone_step) vs the CPU oracle (config C3).
Single step <= 1e-5 rel on body_q / joint_q, velocities within the dt-amplified bound; 100-step rollout <= 1e-4 rel."""
import numpy as np
import pytest

from test_gpu_parity_xpbd import _lower_quadrupeds, _rel, _setup

pytestmark = pytest.mark.gpu


def _step_both(nt, model, o, n_steps, dt, epb=0, with_contacts=True, jf=None):
    from oracle_bridge import OracleState

    s0, s1 = model.state(), model.state()
    ctrl = model.control()
    if jf is not None:
        ctrl.joint_f = jf
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts() if with_contacts else None
    solver = nt.solvers.SolverFeatherstone(model, envs_per_block=epb)
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts() if with_contacts else None
    c = o.control(joint_f=jf)
    for _ in range(n_steps):
        s0.clear_forces()
        if with_contacts:
            pipe.collide(s0, contacts)
        solver.step(s0, s1, ctrl, contacts, dt)
        s0, s1 = s1, s0
        os0.body_f[:] = 0
        if with_contacts:
            o.collide(os0.body_q, oc)
        o.featherstone_step(os0, os1, c, oc, dt)
        os0, os1 = os1, os0
    return s0, os0, oc


@pytest.mark.parametrize("n_env,epb", [(1, 0), (5, 4), (67, 8), (130, 0)])
def test_quadruped_single_step(n_env, epb):
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, n_env)
    _lower_quadrupeds(nt, model, 0.26)
    rng = np.random.default_rng(11)
    model.joint_qd = (model.joint_qd + rng.normal(0, 0.3, size=model.joint_qd.shape)).astype(np.float32)
    jf = rng.normal(0, 2.0, size=model.joint_dof_count).astype(np.float32)
    s, os_, oc = _step_both(nt, model, o, 1, 1e-3, epb=epb, jf=jf)
    assert oc.count[0] > 0
    assert _rel(s.joint_q.cpu().numpy(), os_.joint_q) <= 1e-5
    assert _rel(s.body_q.cpu().numpy(), os_.body_q) <= 1e-5
    assert _rel(s.joint_qd.cpu().numpy(), os_.joint_qd) <= 2e-4
    assert _rel(s.body_qd.cpu().numpy(), os_.body_qd) <= 2e-4


def test_quadruped_rollout_100_steps():
    """100 steps of free flight + PD hold (contacts are emitted through the 0.1 gap but stay inactive)."""
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, 24)
    _lower_quadrupeds(nt, model, 0.08)
    s, os_, oc = _step_both(nt, model, o, 100, 1e-3)
    assert oc.count[0] > 0
    assert _rel(s.joint_q.cpu().numpy(), os_.joint_q) <= 1e-4
    assert _rel(s.body_q.cpu().numpy(), os_.body_q) <= 1e-4
    assert _rel(s.joint_qd.cpu().numpy(), os_.joint_qd) <= 1e-4
    assert _rel(s.body_qd.cpu().numpy(), os_.body_qd) <= 1e-4


def test_quadruped_impact_phase_stepwise():
    """Through touchdown the penalty contacts (ke ~ 1e4 on light feet, explicit integration) amplify rounding noise by
    ~3-10x per step, so trajectories are compared step by step from the oracle's state: 120 consecutive single steps
    covering free flight, impact and rebound, each <= 1e-5 / dt-amplified velocity bound."""
    from oracle_bridge import OracleState
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, 6)
    _lower_quadrupeds(nt, model, 0.22)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverFeatherstone(model)
    os0, os1 = OracleState(model), OracleState(model)
    oc, c = o.contacts(), o.control()
    deepest = 0.0
    for _ in range(120):
        s0.joint_q, s0.joint_qd, s0.body_q, s0.body_qd = os0.joint_q, os0.joint_qd, os0.body_q, os0.body_qd
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, 1e-3)
        os0.body_f[:] = 0
        o.collide(os0.body_q, oc)
        o.featherstone_step(os0, os1, c, oc, 1e-3)
        assert _rel(s1.joint_q.cpu().numpy(), os1.joint_q) <= 1e-5
        assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
        assert _rel(s1.joint_qd.cpu().numpy(), os1.joint_qd) <= 5e-4
        assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 5e-4
        deepest = max(deepest, float(np.max(np.abs(os1.joint_qd))))
        os0, os1 = os1, os0
    assert deepest > 1.0  # the impact really happened inside the window


def test_free_bodies_single_step():
    """8 free boxes per env (one articulation each): exercises the FREE-root branch and na > 1."""
    from scenes import box_stack_scene

    nt, model, o = _setup(box_stack_scene, 19)
    rng = np.random.default_rng(5)
    model.joint_qd = rng.normal(0, 0.5, size=model.joint_qd.shape).astype(np.float32)
    s, os_, _ = _step_both(nt, model, o, 3, 1e-3, with_contacts=False)
    assert _rel(s.joint_q.cpu().numpy(), os_.joint_q) <= 1e-5
    assert _rel(s.body_q.cpu().numpy(), os_.body_q) <= 1e-5
    assert _rel(s.joint_qd.cpu().numpy(), os_.joint_qd) <= 1e-4
    assert _rel(s.body_qd.cpu().numpy(), os_.body_qd) <= 1e-4


def test_pendulum_period_and_energy():
    """test_physics_verification.py:112-186 through the HIP path (revolute pendulum, gravity -10, 64 envs)."""
    import newton_amd as nt

    g, L, a0, dt = -10.0, 1.0, 0.05, 1e-3
    env = nt.ModelBuilder(up_axis=1, gravity=g)
    link = env.add_link()
    env.add_shape_sphere(link, radius=0.01)
    j = env.add_joint_revolute(-1, link, axis=(0, 0, 1), parent_xform=[0, 0, 0, 0, 0, 0, 1], child_xform=[0, L, 0, 0, 0, 0, 1],
                               armature=0.0)
    env.add_articulation([j])
    scene = nt.ModelBuilder(up_axis=1, gravity=g)
    scene.replicate(env, 64)
    model = scene.finalize(device="cuda:0")
    model.joint_q[:] = a0
    mass = float(model.body_mass[0])
    I_pivot = float(np.asarray(model.body_inertia[0]).reshape(3, 3)[2, 2]) + mass * L * L
    T = 2.0 * np.pi * np.sqrt(I_pivot / (mass * abs(g) * L))
    s0, s1 = model.state(), model.state()
    solver = nt.solvers.SolverFeatherstone(model, angular_damping=0.0)
    n = int(1.5 * T / dt)
    angles = []
    for _ in range(n):
        solver.step(s0, s1, None, None, dt)
        s0, s1 = s1, s0
        angles.append(s0.joint_q.cpu().numpy().copy())
    angles = np.array(angles)
    t = np.arange(1, n + 1) * dt
    err = np.mean(np.abs(angles - (a0 * np.cos(2.0 * np.pi / T * t))[:, None])) / a0
    assert err < 0.01
    assert np.max(np.abs(angles - angles[:, :1])) == 0.0  # identical envs stay bit-identical


def test_eval_fk_device_matches_oracle():
    """newton.eval_fk on the device (nt_eval_fk) vs the oracle's eval_fk restatement, random joint state."""
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, 37)
    rng = np.random.default_rng(2)
    E = model.world_count
    jq = model.joint_q.copy()
    jq.reshape(E, -1)[:, 7:] += rng.normal(0, 0.4, size=(E, 12)).astype(np.float32)
    q = rng.normal(size=(E, 4))
    jq.reshape(E, -1)[:, 3:7] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    jqd = rng.normal(0, 1.0, size=model.joint_qd.shape).astype(np.float32)
    state = model.state()
    nt.eval_fk(model, jq, jqd, state)
    bq, bqd = o.eval_fk(jq, jqd)
    assert _rel(state.body_q.cpu().numpy(), bq) <= 1e-5
    assert _rel(state.body_qd.cpu().numpy(), bqd) <= 1e-5


def test_rollout_bitwise_equals_api_loop():
    """nt_featherstone_rollout == the per-call loop {clear_forces; collide; step; swap}, bit for bit (20 steps, contacts on)."""
    from scenes import quadruped_scene

    nt, model, _ = _setup(quadruped_scene, 21)
    _lower_quadrupeds(nt, model, 0.23)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverFeatherstone(model)
    s0, s1 = model.state(), model.state()
    for _ in range(21):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, 1e-3)
        s0, s1 = s1, s0
    r0, r1 = model.state(), model.state()
    res = solver.rollout(r0, r1, None, contacts, 1e-3, 21)
    assert res is r1
    for name in ("joint_q", "joint_qd", "body_q", "body_qd"):
        assert np.array_equal(getattr(res, name).cpu().numpy(), getattr(s0, name).cpu().numpy()), name


def test_uniform_tile_of_16_is_bitwise_the_tiles_of_4_at_full_size():
    """Round 6: from 2 049 uniform-parameter environments up the rollout keeps 16 environments per workgroup (block-shared parameters,
    tree-mode solve region, 32 lanes per environment).  At BASELINE.json's size the automatic choice, the explicit request and the
    per-environment tiles of 4 give the same bits, frame after frame, with all feet in the ground."""
    import ctypes as C

    from scenes import quadruped_scene

    nt, model, _ = _setup(quadruped_scene, 4096)
    _lower_quadrupeds(nt, model, 0.24)
    results = []
    for epb in (4, 0, 16):
        pipe = nt.CollisionPipeline(model)
        contacts = pipe.contacts()
        solver = nt.solvers.SolverFeatherstone(model, envs_per_block=epb)
        r0, r1 = model.state(), model.state()
        for _ in range(3):
            res = solver.rollout(r0, r1, None, contacts, 1e-3, 10)
            assert res is r0
        results.append({k: getattr(r0, k).cpu().numpy().copy() for k in ("joint_q", "joint_qd", "body_q", "body_qd")})
        assert int(contacts.rigid_contact_count.cpu().numpy()[0]) >= 4096 * 4
    for other in results[1:]:
        for k, v in results[0].items():
            assert np.array_equal(v, other[k]), k
    assert np.all(np.isfinite(results[0]["body_q"]))


def test_body_parent_f_matches_oracle_step_and_rollout():
    """State.body_parent_f (compute_body_parent_f, featherstone/kernels.py:2371-2416) of a contact-loaded quadruped step vs
    the oracle (<= 1e-4 of the largest wrench); the fused rollout reports the last substep's wrenches bit-identically."""
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, 37)
    model.request_state_attributes("body_parent_f")
    _lower_quadrupeds(nt, model, 0.26)
    rng = np.random.default_rng(5)
    model.joint_qd = (model.joint_qd + rng.normal(0, 0.3, size=model.joint_qd.shape)).astype(np.float32)
    jf = rng.normal(0, 2.0, size=model.joint_dof_count).astype(np.float32)
    s, os_, oc = _step_both(nt, model, o, 1, 1e-3, jf=jf)
    assert oc.count[0] > 0
    want = os_.body_parent_f
    got = s.body_parent_f.cpu().numpy()
    assert np.abs(want).max() > 1.0
    assert np.max(np.abs(got - want)) <= 1e-4 * np.abs(want).max()

    solver = nt.solvers.SolverFeatherstone(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    r0, r1 = model.state(), model.state()
    out = solver.rollout(r0, r1, None, contacts, 1e-3, 3)
    t0, t1 = model.state(), model.state()
    for _ in range(3):
        t0.clear_forces()
        pipe.collide(t0, contacts)
        solver.step(t0, t1, None, contacts, 1e-3)
        t0, t1 = t1, t0
    assert np.array_equal(out.body_q.cpu().numpy(), t0.body_q.cpu().numpy())
    assert np.array_equal(out.body_parent_f.cpu().numpy(), t0.body_parent_f.cpu().numpy())
