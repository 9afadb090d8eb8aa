"""GPU parity: HIP collide + XPBD (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances: single step max-rel-err <= 1e-5 on body_q / body_qd; broad-phase pair sets and
per-env contact counts bit-exact; contact geometry <= 1e-5; 100-substep rollouts <= 1e-4 rel.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b, floor=1.0):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def _setup(model_fn, n, **kw):
    import newton_amd as nt
    from oracle_bridge import Oracle

    model = model_fn(n, device="cuda:0", **kw)
    return nt, model, Oracle(model)


def _lower_quadrupeds(nt, model, dz):
    E = model.world_count
    model.joint_q.reshape(E, -1)[:, 2] -= dz
    bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    model.body_q, model.body_qd = bq, bqd


def _compare_contacts(model, contacts, oc, pairs_oracle):
    """candidate pair set and per-env contact counts bit-exact; geometry within 1e-5."""
    t = model.env
    E = t.env_count
    n_gpu = int(contacts.rigid_contact_count.cpu().numpy()[0])
    n_or = int(oc.count[0])
    assert n_gpu == n_or
    # candidate pairs
    mask = contacts.candidate_pair_mask.cpu().numpy()
    all_pairs = np.asarray(model.shape_contact_pairs).reshape(E, t.np, 2)
    got = {tuple(p) for p in all_pairs[mask]}
    want = {tuple(p) for p in pairs_oracle}
    assert got == want
    # per env counts
    s0 = oc.shape0[:n_or]
    s1 = oc.shape1[:n_or]
    L0, nloc = t.shape_local0, max(t.ns, 1)
    loc0 = (s0 >= L0) & (s0 < L0 + E * t.ns)
    env_of = np.where(loc0, (s0 - L0) // nloc, (s1 - L0) // nloc)
    want_counts = np.bincount(env_of, minlength=E)
    assert np.array_equal(contacts.rigid_contact_count_per_env.cpu().numpy(), want_counts)
    # flat arrays in the reference's append order
    assert np.array_equal(contacts.rigid_contact_shape0.cpu().numpy()[:n_or], s0)
    assert np.array_equal(contacts.rigid_contact_shape1.cpu().numpy()[:n_or], s1)
    for name in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
        g = getattr(contacts, "rigid_contact_" + name).cpu().numpy()[:n_or]
        w = getattr(oc, name)[:n_or]
        assert n_or == 0 or np.max(np.abs(g - w)) <= 1e-5, name


@pytest.mark.parametrize("n_env,epb", [(1, 0), (5, 16), (64, 8), (130, 16), (257, 0)])
def test_quadruped_single_step(n_env, epb):
    from oracle_bridge import OracleState
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, n_env)
    _lower_quadrupeds(nt, model, 0.26)
    rng = np.random.default_rng(7)
    model.body_qd = (model.body_qd + rng.normal(0, 0.2, size=model.body_qd.shape)).astype(np.float32)
    s0, s1 = model.state(), model.state()
    ctrl = model.control()
    jf = rng.normal(0, 2.0, size=model.joint_dof_count).astype(np.float32)
    ctrl.joint_f = jf
    pipe = nt.CollisionPipeline(model, envs_per_block=epb)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, envs_per_block=epb)
    s0.clear_forces()
    pipe.collide(s0, contacts)
    solver.step(s0, s1, ctrl, contacts, 1e-3)

    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    pairs, _, _ = o.collide(os0.body_q, oc)
    assert oc.count[0] > 0
    o.xpbd_step(os0, os1, o.control(joint_f=jf), oc, 1e-3)
    _compare_contacts(model, contacts, oc, pairs)
    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 1e-5 * 20  # velocities are O(1e-1); dt-amplified


def test_quadruped_rollout_100_substeps():
    """100 substeps: per-call API loop and the fused rollout must both track the oracle to 1e-4 rel."""
    from oracle_bridge import OracleState
    from scenes import quadruped_scene

    n_env = 48
    nt, model, o = _setup(quadruped_scene, n_env)
    _lower_quadrupeds(nt, model, 0.22)
    dt = 1e-3
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model)
    ctrl = model.control()
    # (a) API loop
    s0, s1 = model.state(), model.state()
    for _ in range(100):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, ctrl, contacts, dt)
        s0, s1 = s1, s0
    loop_q, loop_qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
    # (b) fused rollout
    r0, r1 = model.state(), model.state()
    res = solver.rollout(r0, r1, ctrl, contacts, dt, 100)
    roll_q, roll_qd = res.body_q.cpu().numpy(), res.body_qd.cpu().numpy()
    assert np.array_equal(loop_q, roll_q) and np.array_equal(loop_qd, roll_qd), "fused rollout != API loop (bitwise)"
    # (c) oracle
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    c = o.control()
    for _ in range(100):
        os0.body_f[:] = 0
        o.collide(os0.body_q, oc)
        o.xpbd_step(os0, os1, c, oc, dt)
        os0, os1 = os1, os0
    assert _rel(loop_q, os0.body_q) <= 1e-4
    assert _rel(loop_qd, os0.body_qd, floor=1.0) <= 2e-3


def test_mixed_primitives_collide_and_step():
    from oracle_bridge import OracleState
    from scenes import mixed_primitive_scene

    nt, model, o = _setup(mixed_primitive_scene, 33)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=3)
    pipe.collide(s0, contacts)
    solver.step(s0, s1, None, contacts, 1.0 / 240.0)
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    pairs, _, _ = o.collide(os0.body_q, oc)
    assert oc.count[0] > 0
    o.xpbd_step(os0, os1, o.control(), oc, 1.0 / 240.0, iterations=3)
    _compare_contacts(model, contacts, oc, pairs)
    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 2e-4


def test_quadruped_settles_like_reference_example():
    """Invariant of newton/examples/basic/example_basic_urdf.py:145-162 (test_final), 200 frames x 10 substeps:
    root height 0.46 +- 0.01 and all |qd| < 0.15 -- through the fused rollout."""
    from scenes import quadruped_scene

    nt, model, _ = _setup(quadruped_scene, 16, seed=None)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model)
    for _ in range(200):
        res = solver.rollout(s0, s1, None, contacts, 1e-3, 10)
        assert res is s0
    q, qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd))
    assert np.max(np.abs(qd)) < 0.15
    root_z = q.reshape(16, 13, 7)[:, 0, 2]
    assert np.all(np.abs(root_z - 0.46) < 0.01)


def test_product_path_has_no_cpu_fallback():
    import newton_amd as nt
    from newton_amd._lib import NewtonHipError
    from scenes import quadruped_scene

    model = quadruped_scene(2)  # host model
    with pytest.raises(NewtonHipError):
        nt.solvers.SolverXPBD(model)
    with pytest.raises(NewtonHipError):
        nt.CollisionPipeline(model)


def _bounce_scene(world_count, device=None, e=0.8):
    import newton_amd as nt

    cfg = nt.ModelBuilder.ShapeConfig(mu=0.3, restitution=e, margin=0.001, gap=0.02)
    env = nt.ModelBuilder()
    b = env.add_body(xform=[0.0, 0.0, 0.06, 0.0, 0.0, 0.0, 1.0])
    env.add_shape_sphere(b, radius=0.05, cfg=cfg)
    b = env.add_body(xform=[0.3, 0.0, 0.11, 0.1, 0.05, 0.0, 0.99373])
    env.add_shape_box(b, hx=0.1, hy=0.08, hz=0.1, cfg=cfg)
    b = env.add_body(xform=[0.02, 0.01, 0.20, 0.0, 0.0, 0.0, 1.0])
    env.add_shape_sphere(b, radius=0.07, cfg=cfg)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    scene.add_ground_plane(cfg=cfg)
    return scene.finalize(device=device)


def test_restitution_single_steps_match_oracle():
    """enable_restitution=True (apply_rigid_restitution, xpbd/kernels.py:2583-2728): approaching bodies in contact,
    20 consecutive steps compared step by step from the oracle's state."""
    from oracle_bridge import OracleState

    nt, model, o = _setup(_bounce_scene, 9)
    rng = np.random.default_rng(3)
    model.body_qd[:, 2] = -1.0 + rng.normal(0, 0.1, size=model.body_count).astype(np.float32)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2, enable_restitution=True)
    os0, os1 = OracleState(model), OracleState(model)
    oc, c = o.contacts(), o.control()
    bounced = False
    for _ in range(20):
        s0.body_q, s0.body_qd = os0.body_q, os0.body_qd
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, 2e-3)
        o.collide(os0.body_q, oc)
        o.xpbd_step(os0, os1, c, oc, 2e-3, iterations=2, enable_restitution=True)
        assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
        assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 2e-4
        bounced = bounced or bool(np.any(os1.body_qd[:, 2] > 0.3))
        os0, os1 = os1, os0
    assert bounced  # the restitution impulse really fired


def test_restitution_rebound_height():
    """test_physics_verification.py:612-690 through the fused rollout: rebound height = e^2 * drop height within 1 %."""
    import newton_amd as nt

    g, h_drop, radius = -10.0, 1.0, 0.05
    got = {}
    for e in (0.5, 0.8):
        cfg = nt.ModelBuilder.ShapeConfig(mu=0.0, restitution=e, ke=1e4, kd=100.0, kf=0.0, margin=0.001, gap=0.0)
        env = nt.ModelBuilder(up_axis=1, gravity=g)
        b = env.add_body(xform=[0.0, radius + h_drop, 0.0, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_sphere(b, radius=radius, cfg=cfg)
        scene = nt.ModelBuilder(up_axis=1, gravity=g)
        scene.replicate(env, 64)
        scene.add_ground_plane(cfg=cfg)
        model = scene.finalize(device="cuda:0")
        pipe = nt.CollisionPipeline(model)
        contacts = pipe.contacts()
        solver = nt.solvers.SolverXPBD(model, enable_restitution=True)
        s0, s1 = model.state(), model.state()
        n = int(3.0 * np.sqrt(2.0 * h_drop / abs(g)) / 1e-3)
        ys = []
        for _ in range(n // 2):
            res = solver.rollout(s0, s1, None, contacts, 1e-3, 2)
            ys.append(float(res.body_q[0, 1]))
        y = np.array(ys)
        assert y.min() > -0.01
        impact = next(i for i in range(1, len(y) - 1) if y[i] < y[i - 1] and y[i] <= y[i + 1])
        got[e] = np.max(y[impact:]) - radius
        assert abs(got[e] - e * e * h_drop) < 0.012 * e * e * h_drop
    assert abs(got[0.8] / got[0.5] - 2.56) < 0.02 * 2.56


def test_state_reset_masked_device():
    """nt_state_reset: masked env-column copy of every state array."""
    from scenes import quadruped_scene

    nt, model, _ = _setup(quadruped_scene, 70)
    default = model.state()
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model)
    res = solver.rollout(s0, s1, None, contacts, 1e-3, 20)
    moved = res.body_q.cpu().numpy().reshape(70, -1).copy()
    mask = np.zeros(71, dtype=bool)
    mask[[0, 5, 64, 69]] = True
    res.reset(default, world_mask=mask)
    q = res.body_q.cpu().numpy().reshape(70, -1)
    d = default.body_q.cpu().numpy().reshape(70, -1)
    sel = mask[:70]
    assert np.array_equal(q[sel], d[sel]) and np.array_equal(q[~sel], moved[~sel])
    assert np.array_equal(res.body_qd.cpu().numpy().reshape(70, -1)[sel], default.body_qd.cpu().numpy().reshape(70, -1)[sel])


def test_ground_plane_first_scene_matches_oracle():
    """Global shape in front of the env-local block (add_ground_plane() before replicate): ids, pairs and contacts line up."""
    from oracle_bridge import OracleState

    def scene(n, device=None):
        import newton_amd as nt

        env = nt.ModelBuilder()
        b = env.add_body(xform=[0.0, 0.0, 0.095, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_box(b, hx=0.2, hy=0.15, hz=0.1)
        b = env.add_body(xform=[0.05, 0.02, 0.29, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_sphere(b, radius=0.1)
        b = env.add_body(xform=[0.5, 0.0, 0.14, 0.7071068, 0.0, 0.0, 0.7071068])
        env.add_shape_capsule(b, radius=0.05, half_height=0.1)
        sc = nt.ModelBuilder()
        sc.add_ground_plane()
        sc.replicate(env, n)
        return sc.finalize(device=device)

    nt, model, o = _setup(scene, 19)
    assert model.env.shape_local0 == 1
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    pipe.collide(s0, contacts)
    solver.step(s0, s1, None, contacts, 1.0 / 240.0)
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    pairs, _, _ = o.collide(os0.body_q, oc)
    assert oc.count[0] >= 19 * 5
    o.xpbd_step(os0, os1, o.control(), oc, 1.0 / 240.0, iterations=2)
    _compare_contacts(model, contacts, oc, pairs)
    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 2e-4


def test_kinematic_platform_stepwise():
    """A kinematic (prescribed-motion) platform carrying dynamic bodies: kinematic bodies pass through every kernel
    unchanged and feed the friction rows through their velocity (xpbd/kernels.py:2318-2334); 40 steps vs the oracle."""
    from oracle_bridge import OracleState

    def scene(n, device=None):
        import newton_amd as nt

        env = nt.ModelBuilder()
        plat = env.add_body(xform=[0.0, 0.0, 0.5, 0.0, 0.0, 0.0, 1.0], is_kinematic=True)
        env.add_shape_box(plat, hx=1.0, hy=1.0, hz=0.05)
        b = env.add_body(xform=[0.1, 0.0, 0.649, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_box(b, hx=0.1, hy=0.1, hz=0.1)
        b = env.add_body(xform=[-0.3, 0.2, 0.629, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_sphere(b, radius=0.08)
        sc = nt.ModelBuilder()
        sc.replicate(env, n)
        sc.add_ground_plane()
        return sc.finalize(device=device)

    nt, model, o = _setup(scene, 13)
    E = 13
    qd = model.body_qd.reshape(E, 3, 6)
    qd[:, 0, 0] = 0.5   # the platform slides along +x ...
    qd[:, 0, 5] = 0.3   # ... and yaws
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=3)
    os0, os1 = OracleState(model), OracleState(model)
    oc, c = o.contacts(), o.control()
    dragged = False
    for _ in range(40):
        s0.body_q, s0.body_qd = os0.body_q, os0.body_qd
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, 2e-3)
        os0.body_f[:] = 0
        pairs, _, _ = o.collide(os0.body_q, oc)
        o.xpbd_step(os0, os1, c, oc, 2e-3, iterations=3)
        assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
        assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 3e-4
        # the platform itself is passed through untouched
        assert np.array_equal(os1.body_q.reshape(E, 3, 7)[:, 0], os0.body_q.reshape(E, 3, 7)[:, 0])
        dragged = dragged or bool(np.any(os1.body_qd.reshape(E, 3, 6)[:, 1, 0] > 0.05))
        os0, os1 = os1, os0
    assert dragged  # friction against the moving platform accelerated the box


def test_articulation_view_device():
    """ArticulationView on a device state: getters are strided views of the env-major buffers (no copy), masked setters and
    masked FK only touch the selected worlds, and the result matches the host implementation."""
    import torch
    from newton_amd.selection import ArticulationView
    from scenes import quadruped_scene

    nt, model, _ = _setup(quadruped_scene, 70)
    host = quadruped_scene(70)
    view, hview = ArticulationView(model, "*"), ArticulationView(host, "*")
    s, hs = model.state(), host.state()
    q = view.get_dof_positions(s)
    assert q.shape == (70, 19) and q.data_ptr() == s._soa["joint_q"].data_ptr()  # a view, not a copy
    rng = np.random.default_rng(0)
    new_q = hview.get_dof_positions(hs).copy()
    new_q[:, 7:] += rng.normal(0, 0.2, size=(70, 12)).astype(np.float32)
    mask = rng.random(70) < 0.4
    view.set_dof_positions(s, new_q, mask=mask)
    hview.set_dof_positions(hs, new_q, mask=mask)
    assert np.array_equal(view.get_dof_positions(s).cpu().numpy(), hview.get_dof_positions(hs))
    view.eval_fk(s, mask=mask)
    hview.eval_fk(hs, mask=mask)
    assert np.max(np.abs(s.body_q.cpu().numpy() - hs.body_q)) <= 1e-5
    assert view.get_link_transforms(s).shape == (70, 13, 7)
    assert torch.equal(view.get_root_transforms(s), view.get_dof_positions(s)[:, :7])


@pytest.mark.parametrize("n_env,epb", [(3, 0), (70, 16)])
def test_reported_contact_force_and_parent_force_match_oracle(n_env, epb):
    """contacts.force (update_contacts) and state_out.body_parent_f of one XPBD step vs the oracle, with joint_f drive,
    external body_f and ground contacts in play: <= 1e-4 of the largest wrench."""
    from oracle_bridge import OracleState
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, n_env)
    model.request_state_attributes("body_parent_f")
    model.request_contact_attributes("force")
    _lower_quadrupeds(nt, model, 0.26)
    rng = np.random.default_rng(11)
    model.body_qd = (model.body_qd + rng.normal(0, 0.2, size=model.body_qd.shape)).astype(np.float32)
    joint_f = rng.normal(0, 5.0, size=model.joint_dof_count).astype(np.float32)
    body_f = rng.normal(0, 3.0, size=(model.body_count, 6)).astype(np.float32)
    s0, s1 = model.state(), model.state()
    s0.body_f = body_f
    ctrl = model.control()
    ctrl.joint_f = joint_f
    pipe = nt.CollisionPipeline(model, envs_per_block=epb)
    contacts = pipe.contacts()
    pipe.collide(s0, contacts)
    solver = nt.solvers.SolverXPBD(model, envs_per_block=epb)
    solver.step(s0, s1, ctrl, contacts, 1e-3)
    solver.update_contacts(contacts, s1)

    os0, os1 = OracleState(model, body_f=body_f), OracleState(model)
    oc = o.contacts()
    o.collide(os0.body_q, oc)
    n = int(oc.count[0])
    assert n > 0
    force = np.zeros((oc.max, 6), dtype=np.float32)
    o.xpbd_step(os0, os1, o.control(joint_f=joint_f), oc, 1e-3, contact_force_out=force)

    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    got_f = contacts.force.cpu().numpy()
    assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == n
    assert np.abs(force[:n]).max() > 1.0
    assert np.max(np.abs(got_f[:n] - force[:n])) <= 1e-4 * np.abs(force[:n]).max()
    assert np.all(got_f[n:] == 0.0)
    got_p = s1.body_parent_f.cpu().numpy()
    assert np.abs(os1.body_parent_f).max() > 1.0
    assert np.max(np.abs(got_p - os1.body_parent_f)) <= 1e-4 * np.abs(os1.body_parent_f).max()
    # the fused rollout falls back to the per-substep loop when reporting is requested: same numbers as stepping
    r0, r1 = model.state(), model.state()
    out = solver.rollout(r0, r1, ctrl, contacts, 1e-3, 3)
    t0, t1 = model.state(), model.state()
    for _ in range(3):
        t0.clear_forces()
        pipe.collide(t0, contacts)
        solver.step(t0, t1, ctrl, contacts, 1e-3)
        t0, t1 = t1, t0
    assert np.array_equal(out.body_q.cpu().numpy(), t0.body_q.cpu().numpy())
    assert np.array_equal(out.body_parent_f.cpu().numpy(), t0.body_parent_f.cpu().numpy())
