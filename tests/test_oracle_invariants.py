"""Pin the CPU oracle (and the host builder / URDF importer / FK feeding it) against behavioural acceptance tests
of the reference -- the reference holds no golden vectors for this path."""
import numpy as np
import pytest

import newton_amd as nt
from oracle_bridge import Oracle, OracleState
from scenes import box_stack_scene, quadruped_scene


def _run(model, frames, substeps, dt, **xpbd):
    o = Oracle(model)
    s0, s1 = OracleState(model), OracleState(model)
    ct, ctrl = o.contacts(), o.control()
    for _ in range(frames * substeps):
        s0.body_f[:] = 0
        o.collide(s0.body_q, ct)
        o.xpbd_step(s0, s1, ctrl, ct, dt, **xpbd)
        s0, s1 = s1, s0
    return s0, ct


def test_quadruped_example_final_state():
    """newton/examples/basic/example_basic_urdf.py:145-162 (`test_final`, run for 200 frames by
    newton/tests/test_examples.py:349-357): all |qd| < 0.15 and root height 0.46 +- 0.01."""
    model = quadruped_scene(2, seed=None)
    s, _ = _run(model, 200, 10, 1e-3)
    assert np.all(np.isfinite(s.body_q)) and np.max(np.abs(s.body_qd)) < 0.15
    assert np.all(np.abs(s.body_q.reshape(2, 13, 7)[:, 0, 2] - 0.46) < 0.01)
    # asv gate (bench_quadruped_xpbd.py:58-66): quaternions stay normalised
    assert np.max(np.abs(np.linalg.norm(s.body_q[:, 3:], axis=1) - 1.0)) < 1e-3


def test_free_fall_matches_analytic():
    """test_physics_verification.py:48-110 style: symplectic Euler free fall, v = g t, z = z0 + g dt^2 n(n+1)/2."""
    b = nt.ModelBuilder()
    body = b.add_body(xform=[0, 0, 10.0, 0, 0, 0, 1])
    b.add_shape_sphere(body, radius=0.1)
    model = b.finalize()
    o = Oracle(model)
    s0, s1 = OracleState(model), OracleState(model)
    n, dt = 100, 1e-3
    for _ in range(n):
        o.xpbd_step(s0, s1, o.control(), None, dt)
        s0, s1 = s1, s0
    assert abs(s0.body_qd[0, 2] - (-9.81 * n * dt)) < 1e-4
    assert abs(s0.body_q[0, 2] - (10.0 - 9.81 * dt * dt * n * (n + 1) / 2)) < 1e-4


@pytest.mark.parametrize("kind,rest_z", [("sphere", 0.1), ("box", 0.1), ("capsule_upright", 0.25), ("cylinder", 0.1)])
def test_shapes_rest_on_plane(kind, rest_z):
    """newton/tests/test_rigid_contact.py:28-236: shapes settle on the ground plane at the expected height."""
    b = nt.ModelBuilder()
    body = b.add_body(xform=[0, 0, rest_z + 0.05, 0, 0, 0, 1])
    if kind == "sphere":
        b.add_shape_sphere(body, radius=0.1)
    elif kind == "box":
        b.add_shape_box(body, hx=0.2, hy=0.15, hz=0.1)
    elif kind == "capsule_upright":
        b.add_shape_capsule(body, radius=0.1, half_height=0.15)
    else:
        b.add_shape_cylinder(body, radius=0.15, half_height=0.1)
    b.add_ground_plane()
    model = b.finalize()
    s, ct = _run(model, 60, 10, 1.0 / 600.0, iterations=4)
    assert abs(s.body_q[0, 2] - rest_z) < 5e-3
    assert np.max(np.abs(s.body_qd)) < 5e-2
    assert ct.count[0] >= 1


def test_articulation_does_not_drift():
    """newton/tests/test_solver_xpbd.py:750-845: contacts are solved before joints; a standing quadruped's joint
    anchors stay together (drift < 1 cm over 3 s is the reference bound; 1 s here)."""
    model = quadruped_scene(1, seed=None)
    s, _ = _run(model, 100, 10, 1e-3)
    X_p, X_c = model.joint_X_p, model.joint_X_c
    from newton_amd import _np_math as nm

    for j in range(1, model.joint_count):
        p, c = model.joint_parent[j], model.joint_child[j]
        wp_ = nm.transform_mul(s.body_q[p].astype(np.float64), X_p[j].astype(np.float64))
        wc_ = nm.transform_mul(s.body_q[c].astype(np.float64), X_c[j].astype(np.float64))
        assert np.linalg.norm(wp_[:3] - wc_[:3]) < 1e-2


def test_plane_box_half_of_stack_scene():
    """C2 scene: the bottom box of each stack gets exactly 4 ground contacts (analytic kernel, appended first), every
    touching box-box face gets a 4-point manifold (MPR kernel, appended after); candidate pairs = ground pair +
    adjacent boxes."""
    model = box_stack_scene(3, n_boxes=3, seed=0)
    o = Oracle(model)
    ct = o.contacts()
    pairs, _, _ = o.collide(model.body_q, ct)
    ground = model.shape_count - 1
    got = {tuple(p) for p in pairs}
    for w in range(3):
        assert (3 * w, ground) in got
        assert (3 * w, 3 * w + 1) in got and (3 * w + 1, 3 * w + 2) in got
    assert ct.count[0] == 3 * 4 + 3 * 2 * 4
    assert np.all(ct.shape0[:12] == ground)
    assert np.allclose(ct.normal[:12], [0, 0, 1])
    assert np.all(ct.shape0[12:36] != ground) and np.all(ct.shape1[12:36] != ground)
    assert np.allclose(ct.normal[12:36], [0, 0, 1], atol=1e-5)


def test_fk_numpy_matches_oracle():
    model = quadruped_scene(3, seed=5)
    rng = np.random.default_rng(0)
    jq = model.joint_q.copy()
    jq.reshape(3, -1)[:, 7:] += rng.normal(0, 0.3, size=(3, 12)).astype(np.float32)
    q = rng.normal(size=(3, 4))
    jq.reshape(3, -1)[:, 3:7] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    jqd = rng.normal(0, 1.0, size=model.joint_qd.shape).astype(np.float32)
    bq, bqd = nt.articulation.eval_fk_numpy(model, jq, jqd)
    obq, obqd = Oracle(model).eval_fk(jq, jqd)
    assert np.max(np.abs(bq - obq)) < 1e-5
    assert np.max(np.abs(bqd - obqd)) < 1e-4


def test_pendulum_example_final_state():
    """newton/examples/basic/example_basic_pendulum.py:113-137 (`test_final`, 100 frames x 10 substeps, SolverSemiImplicit
    defaults): links stay in the swing plane and inside the reachable area, velocities bounded."""
    from scenes import pendulum_scene

    model = pendulum_scene(1)
    o = Oracle(model)
    s0, s1 = OracleState(model), OracleState(model)
    ct, c = o.contacts(), o.control()
    for _ in range(1000):
        s0.body_f[:] = 0
        o.collide(s0.body_q, ct)
        o.semi_implicit_step(s0, s1, c, ct, 1e-3)
        s0, s1 = s1, s0
    for b in range(2):
        q, qd = s0.body_q[b], s0.body_qd[b]
        assert abs(q[0]) < 1e-5 and abs(q[1]) < 1.0 and 0.0 < q[2] < 5.0
        assert abs(qd[0]) < 1e-4 and abs(qd[1]) < 10.0 and abs(qd[2]) < 5.0 and abs(qd[3]) < 10.0 and abs(qd[4]) < 10.0


def test_restitution_rebound_height():
    """test_physics_verification.py:612-690 (SolverXPBD(enable_restitution=True)): a sphere dropped from 1 m rebounds to
    e^2 m within 1 %, and the two restitution values keep the 2.56 ratio."""
    import newton_amd as nt

    g, h_drop, radius = -10.0, 1.0, 0.05
    got = {}
    for e in (0.5, 0.8):
        cfg = nt.ModelBuilder.ShapeConfig(mu=0.0, restitution=e, ke=1e4, kd=100.0, kf=0.0, margin=0.001, gap=0.0)
        b = nt.ModelBuilder(up_axis=1, gravity=g)
        body = b.add_body(xform=[0.0, radius + h_drop, 0.0, 0.0, 0.0, 0.0, 1.0])
        b.add_shape_sphere(body, radius=radius, cfg=cfg)
        b.add_ground_plane(cfg=cfg)
        m = b.finalize()
        o = Oracle(m)
        ct, c = o.contacts(), o.control()
        s0, s1 = OracleState(m), OracleState(m)
        ys = []
        for _ in range(int(3.0 * np.sqrt(2.0 * h_drop / abs(g)) / 1e-3)):
            s0.body_f[:] = 0
            o.collide(s0.body_q, ct)
            o.xpbd_step(s0, s1, c, ct, 1e-3, enable_restitution=True)
            s0, s1 = s1, s0
            ys.append(float(s0.body_q[0, 1]))
        y = np.array(ys)
        assert y.min() > -0.01
        impact = next(i for i in range(1, len(y) - 1) if y[i] < y[i - 1] and y[i] <= y[i + 1])
        got[e] = np.max(y[impact:]) - radius
        assert abs(got[e] - e * e * h_drop) < 0.01 * e * e * h_drop
    assert abs(got[0.8] / got[0.5] - 2.56) < 0.01 * 2.56
