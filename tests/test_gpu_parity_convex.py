"""GPU parity for the convex narrow phase (MPR -> GJK -> manifold) and config C2 (box stack, XPBD):
HIP path through the C ABI vs the CPU oracle.  Contact counts / candidate pairs bit-exact, geometry <= 1e-5."""
import numpy as np
import pytest

from test_gpu_parity_xpbd import _compare_contacts, _rel, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(set(__import__("pair_scenes").CONVEX_CASES) - set(__import__("pair_scenes").RECENT_CONVEX_CASES)))
def test_convex_pair_contacts(name):
    import newton_amd as nt
    from oracle_bridge import Oracle
    from pair_scenes import CONVEX_CASES, pair_model

    model = pair_model(CONVEX_CASES[name], device="cuda:0")
    o = Oracle(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s0 = model.state()
    pipe.collide(s0, contacts)
    oc = o.contacts()
    pairs, _, _ = o.collide(model.body_q, oc)
    _compare_contacts(model, contacts, oc, pairs)


@pytest.mark.parametrize("n_env,epb", [(1, 0), (37, 8), (130, 0)])
def test_box_stack_single_step(n_env, epb):
    """C2: 8-box stack; collide + one XPBD step (4 iterations) against the oracle."""
    from oracle_bridge import OracleState
    from scenes import box_stack_scene

    nt, model, o = _setup(box_stack_scene, n_env)
    # push the stack 2 mm into itself so every box-box pair penetrates and the manifolds are exercised
    model.body_q[:, 2] -= 0.002 * (np.arange(model.body_count) % 8 + 1)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model, envs_per_block=epb)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4, envs_per_block=epb)
    pipe.collide(s0, contacts)
    solver.step(s0, s1, None, contacts, 1.0 / 240.0)
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    pairs, _, _ = o.collide(os0.body_q, oc)
    assert oc.count[0] >= n_env * (4 + 7 * 4)
    o.xpbd_step(os0, os1, o.control(), oc, 1.0 / 240.0, iterations=4)
    _compare_contacts(model, contacts, oc, pairs)
    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 2e-4


def test_box_stack_rollout_matches_api_loop_and_oracle():
    from oracle_bridge import OracleState
    from scenes import box_stack_scene

    nt, model, o = _setup(box_stack_scene, 9)
    dt = 1.0 / 240.0
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    s0, s1 = model.state(), model.state()
    for _ in range(40):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, dt)
        s0, s1 = s1, s0
    r0, r1 = model.state(), model.state()
    res = solver.rollout(r0, r1, None, contacts, dt, 40)
    assert np.array_equal(res.body_q.cpu().numpy(), s0.body_q.cpu().numpy()), "fused rollout != API loop (bitwise)"
    os0, os1 = OracleState(model), OracleState(model)
    oc, c = o.contacts(), o.control()
    for _ in range(40):
        os0.body_f[:] = 0
        o.collide(os0.body_q, oc)
        o.xpbd_step(os0, os1, c, oc, dt, iterations=4)
        os0, os1 = os1, os0
    assert _rel(s0.body_q.cpu().numpy(), os0.body_q) <= 1e-4
    assert _rel(s0.body_qd.cpu().numpy(), os0.body_qd) <= 2e-3


def test_aligned_box_stack_remains_stable():
    """test_solver_xpbd.py:1791-1840 through the HIP path: 5 aligned boxes, 180 frames x 4 substeps, 4 iterations."""
    import newton_amd as nt

    env = nt.ModelBuilder()
    for k in range(5):
        b = env.add_body(xform=[0, 0, 0.5 + k, 0, 0, 0, 1])
        env.add_shape_box(b, hx=0.5, hy=0.5, hz=0.5)
    scene = nt.ModelBuilder()
    scene.replicate(env, 64)
    scene.add_ground_plane()
    model = scene.finalize(device="cuda:0")
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    s0, s1 = model.state(), model.state()
    for _ in range(180):
        res = solver.rollout(s0, s1, None, contacts, 1.0 / 240.0, 4)
        assert res is s0
    q = s0.body_q.cpu().numpy().reshape(64, 5, 7)
    assert np.all(np.isfinite(q))
    assert np.allclose(q[:, :, 2], 0.5 + np.arange(5), atol=2e-2)
    assert np.max(np.linalg.norm(q[:, :, :2], axis=-1)) < 1e-2
    assert np.max(np.linalg.norm(q[:, :, 3:5], axis=-1)) < 1e-3


def _hull_pile_scene(world_count, device=None):
    """Per env: a hull box, a scaled hull 'pebble', a cone and a primitive box dropped close together on the ground plane:
    plane-hull / plane-cone pairs go through the infinite-plane box proxy, hull-hull / hull-box / cone-* through MPR."""
    import newton_amd as nt

    env = nt.ModelBuilder()
    cube = nt.Mesh.create_box(0.2, 0.15, 0.1)
    pebble = nt.Mesh.create_sphere(0.15, 8, 10)
    b = env.add_body(xform=[0.0, 0.0, 0.095, 0.0, 0.0, 0.0, 1.0])
    env.add_shape_convex_hull(b, mesh=cube)
    b = env.add_body(xform=[0.05, 0.03, 0.30, *nt._np_math.quat_rpy(0.2, -0.1, 0.4)])
    env.add_shape_convex_hull(b, mesh=pebble, scale=(1.0, 0.8, 0.7))
    b = env.add_body(xform=[0.5, 0.0, 0.19, 0.0, 0.0, 0.0, 1.0])
    env.add_shape_cone(b, radius=0.15, half_height=0.2)
    b = env.add_body(xform=[0.3, 0.25, 0.09, *nt._np_math.quat_rpy(0.0, 0.0, 0.5)])
    env.add_shape_box(b, hx=0.15, hy=0.1, hz=0.1)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    scene.add_ground_plane()
    return scene.finalize(device=device)


@pytest.mark.parametrize("n_env", [1, 21])
def test_hull_pile_collide_and_step(n_env):
    from oracle_bridge import OracleState

    nt, model, o = _setup(_hull_pile_scene, n_env)
    rng = np.random.default_rng(4)
    off = rng.uniform(-0.004, 0.004, size=(model.body_count, 3)).astype(np.float32)
    model.body_q[:, :3] += off
    model.joint_q.reshape(-1, 7)[:, :3] += off
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=3)
    pipe.collide(s0, contacts)
    solver.step(s0, s1, None, contacts, 1.0 / 240.0)
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    pairs, _, _ = o.collide(os0.body_q, oc)
    assert oc.count[0] >= n_env * 10
    o.xpbd_step(os0, os1, o.control(), oc, 1.0 / 240.0, iterations=3)
    _compare_contacts(model, contacts, oc, pairs)
    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 2e-4


def test_hull_pile_settles():
    """The pile comes to rest on the plane: finite state, nothing sinks below the ground, speeds decay."""
    nt, model, _ = _setup(_hull_pile_scene, 32)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    for _ in range(120):
        res = solver.rollout(s0, s1, None, contacts, 1.0 / 480.0, 8)
        assert res is s0
    q, qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd))
    assert np.all(q[:, 2] > 0.03)
    assert np.max(np.abs(qd[:, :3])) < 0.5  # the scaled pebble may still be rolling off the box
