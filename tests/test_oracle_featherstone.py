"""Pin the oracle's SolverFeatherstone restatement against the reference's physics-verification invariants
(newton/tests/test_physics_verification.py:48-537, solver_fn = SolverFeatherstone(angular_damping=0.0)), with the
reference's own tolerances, plus a contact sanity check on the C3 quadruped."""
import numpy as np

import newton_amd as nt
from newton_amd import _np_math as nm
from oracle_bridge import Oracle, OracleState
from scenes import quadruped_scene

I4 = [0.0, 0.0, 0.0, 1.0]


def _run(o, s0, s1, c, n, dt, contacts=None, collide=False, each=None):
    for i in range(1, n + 1):
        s0.body_f[:] = 0
        if collide:
            o.collide(s0.body_q, contacts)
        o.featherstone_step(s0, s1, c, contacts, dt)
        s0, s1 = s1, s0
        if each is not None:
            each(i, s0)
    return s0, s1


def test_free_fall(oracle_lib):
    g, h0, dt = -10.0, 5.0, 1e-3
    b = nt.ModelBuilder(up_axis=1, gravity=g)
    body = b.add_body(xform=[0.0, h0, 0.0, *I4])
    b.add_shape_sphere(body, radius=0.1)
    m = b.finalize()
    o = Oracle(m)
    s0, s1 = OracleState(m), OracleState(m)

    def check(i, s):
        if i % 100:
            return
        t = i * dt
        pos, vel = s.body_q[0, :3], s.body_qd[0, :3]
        assert abs(pos[1] - (h0 + 0.5 * g * t * t)) < max(2.0 * 0.5 * abs(g) * dt * t, 1e-3)
        assert abs(vel[1] - g * t) < max(abs(g) * dt, 1e-3)
        assert abs(pos[0]) < 1e-4 and abs(pos[2]) < 1e-4

    _run(o, s0, s1, o.control(), 500, dt, each=check)


def _pendulum(initial_angle, g=-10.0, L=1.0):
    b = nt.ModelBuilder(up_axis=1, gravity=g)
    link = b.add_link()
    b.add_shape_sphere(link, radius=0.01)
    j = b.add_joint_revolute(-1, link, axis=(0, 0, 1), parent_xform=[0, 0, 0, *I4], child_xform=[0, L, 0, *I4], armature=0.0)
    b.add_articulation([j])
    m = b.finalize()
    m.joint_q[0] = initial_angle
    mass = float(m.body_mass[0])
    I_pivot = float(np.asarray(m.body_inertia[0]).reshape(3, 3)[2, 2]) + mass * L * L
    return m, mass, I_pivot


def test_pendulum_period(oracle_lib):
    g, L, a0, dt = -10.0, 1.0, 0.05, 1e-3
    m, mass, I_pivot = _pendulum(a0, g, L)
    T = 2.0 * np.pi * np.sqrt(I_pivot / (mass * abs(g) * L))
    o = Oracle(m)
    s0, s1 = OracleState(m), OracleState(m)
    n = int(3.5 * T / dt)
    angles = []
    _run(o, s0, s1, o.control(), n, dt, each=lambda i, s: angles.append(float(s.joint_q[0])))
    t = np.arange(1, n + 1) * dt
    err = np.mean(np.abs(np.array(angles) - a0 * np.cos(2.0 * np.pi / T * t))) / a0
    assert err < 0.01


def test_energy_conservation(oracle_lib):
    g, L, dt = -10.0, 1.0, 1e-3
    m, mass, I_pivot = _pendulum(1.0, g, L)
    o = Oracle(m)
    s0, s1 = OracleState(m), OracleState(m)

    def energy(s):
        return 0.5 * I_pivot * float(s.joint_qd[0]) ** 2, mass * abs(g) * (-L * np.cos(float(s.joint_q[0])))

    ke0, pe0 = energy(s0)
    E0 = ke0 + pe0
    kes, es = [], []

    def rec(i, s):
        ke, pe = energy(s)
        kes.append(ke)
        es.append(ke + pe)

    _run(o, s0, s1, o.control(), int(2.0 / dt), dt, each=rec)
    assert min(kes) / abs(E0) < 0.01
    assert np.max(np.abs(np.array(es) - E0)) / abs(E0) < 0.005


def test_momentum_conservation(oracle_lib):
    positions = [(0.0, 0.0, 0.0), (100.0, 0.0, 0.0), (0.0, 100.0, 0.0), (0.0, 0.0, 100.0)]
    velocities = [(1.0, 0.0, 0.0, 0.0, 0.0, 0.5), (0.0, -1.0, 0.0, 0.3, 0.0, 0.0), (0.0, 0.0, 1.5, 0.0, -0.2, 0.0),
                  (-0.5, 0.5, -0.5, 0.0, 0.0, -0.3)]
    b = nt.ModelBuilder(up_axis=1, gravity=0.0)
    for p in positions:
        body = b.add_body(xform=[*p, *I4])
        b.add_shape_box(body, hx=0.5, hy=0.5, hz=0.5)
    m = b.finalize()
    m.joint_qd[:] = np.asarray(velocities, dtype=np.float32).reshape(-1)
    o = Oracle(m)
    bq, bqd = o.eval_fk(m.joint_q, m.joint_qd)
    s0, s1 = OracleState(m, body_q=bq, body_qd=bqd), OracleState(m)

    def momenta(s):
        p, Lm = np.zeros(3), np.zeros(3)
        for i in range(4):
            mass = float(m.body_mass[i])
            v, w, r = s.body_qd[i, :3].astype(np.float64), s.body_qd[i, 3:].astype(np.float64), s.body_q[i, :3].astype(np.float64)
            R = nm.quat_to_matrix(s.body_q[i, 3:7])
            p += mass * v
            Lm += np.cross(r, mass * v) + R @ np.asarray(m.body_inertia[i]).reshape(3, 3) @ R.T @ w
        return p, Lm

    p0, L0 = momenta(s0)
    assert np.linalg.norm(p0) > 0.1 and np.linalg.norm(L0) > 0.1
    s0, _ = _run(o, s0, s1, o.control(), 1000, 1e-3)
    p1, L1 = momenta(s0)
    assert np.linalg.norm(p1 - p0) / np.linalg.norm(p0) < 5e-4
    assert np.linalg.norm(L1 - L0) / np.linalg.norm(L0) < 5e-4
    assert np.linalg.norm(s0.body_q[:4, :3] - np.asarray(positions)) > 0.1


def test_quadruped_contact_drop_stays_sane(oracle_lib):
    """C3: quadruped dropped on the plane under SolverFeatherstone defaults: finite state, penalty contacts hold the
    feet near the ground, joint_q/joint_qd stay consistent with body_q (FK round trip)."""
    m = quadruped_scene(2, seed=1)
    o = Oracle(m)
    ct, c = o.contacts(), o.control()
    s0, s1 = OracleState(m), OracleState(m)
    s0, s1 = _run(o, s0, s1, c, 400, 1e-3, contacts=ct, collide=True)
    assert np.all(np.isfinite(s0.body_q)) and np.all(np.isfinite(s0.joint_q))
    z = s0.body_q.reshape(2, 13, 7)[:, :, 2]
    assert np.all(z > 0.0) and np.all(z < 1.0)
    bq, _ = o.eval_fk(s0.joint_q, s0.joint_qd)
    assert np.max(np.abs(bq - s0.body_q)) < 1e-5


def _probe_H(model, n):
    import ctypes as C

    o = Oracle(model)
    H = np.zeros(n * n, dtype=np.float32)
    o.L.o_featherstone_probe_H(H.ctypes.data_as(C.POINTER(C.c_float)), n * n)
    s0, s1 = OracleState(model), OracleState(model)
    o.featherstone_step(s0, s1, o.control(), None, 1e-3)
    return H.reshape(n, n).astype(np.float64)


def test_mass_matrix_fixed_base_pendulum_closed_form(oracle_lib):
    """test_jacobian_mass_matrix.py:413-439: H = I_zz + m L^2 for a fixed-base z-revolute pendulum."""
    mass, length, izz = 2.0, 0.75, 0.2
    b = nt.ModelBuilder(gravity=0.0)
    body = b.add_link(mass=mass, inertia=np.diag([0.1, 0.15, izz]))
    j = b.add_joint_revolute(-1, body, axis=(0, 0, 1), child_xform=[-length, 0.0, 0.0, *I4], armature=0.0)
    b.add_articulation([j])
    H = _probe_H(b.finalize(), 1)
    assert abs(H[0, 0] - (izz + mass * length ** 2)) < 1e-6 * (izz + mass * length ** 2) + 1e-6


def test_mass_matrix_floating_base_pendulum_closed_form(oracle_lib):
    """test_jacobian_mass_matrix.py:442-470: 7x7 H of a free base with one revolute child at identity pose."""
    base_mass, child_mass, length = 3.0, 2.0, 0.6
    base_inertia, child_inertia = (0.4, 0.5, 0.6), (0.2, 0.25, 0.3)
    b = nt.ModelBuilder(gravity=0.0)
    base = b.add_link(mass=base_mass, inertia=np.diag(base_inertia))
    child = b.add_link(mass=child_mass, inertia=np.diag(child_inertia))
    jf = b.add_joint_free(base)
    jr = b.add_joint_revolute(base, child, axis=(0, 0, 1), child_xform=[-length, 0.0, 0.0, *I4], armature=0.0)
    b.add_articulation([jf, jr])
    H = _probe_H(b.finalize(), 7)
    T = np.zeros((6, 7))
    T[0, 0] = T[1, 1] = T[2, 2] = 1.0
    T[2, 4], T[1, 5], T[1, 6] = -length, length, length
    T[3, 3] = T[4, 4] = T[5, 5] = 1.0
    T[5, 6] = 1.0
    expected = T.T @ np.diag([child_mass] * 3 + list(child_inertia)) @ T
    expected[np.arange(3), np.arange(3)] += base_mass
    expected[np.arange(3, 6), np.arange(3, 6)] += np.asarray(base_inertia)
    assert np.allclose(H, expected, rtol=1e-6, atol=1e-6)


def test_d6_with_three_angular_axes_moves_like_a_ball_joint(oracle_lib):
    """A spinning asymmetric pendulum on a D6 joint with three angular axes follows the same motion as on a BALL joint
    (featherstone/kernels.py:266-335: FK-transported axes as motion subspace + the apparent derivative of that subspace);
    different coordinates, same physics, 0.5 s at dt = 2e-4."""
    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState

    D = nt.ModelBuilder.JointDofConfig
    w0 = np.array([1.5, -0.8, 2.0], dtype=np.float32)
    out = {}
    for kind in ("ball", "d6"):
        b = nt.ModelBuilder(gravity=(0.0, 0.0, -9.81))
        link = b.add_link()
        b.add_shape_box(link, hx=0.05, hy=0.1, hz=0.2, xform=[0.0, 0.0, -0.25, 0.0, 0.0, 0.0, 1.0])
        if kind == "ball":
            j = b.add_joint_ball(-1, link, parent_xform=[0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0])
        else:
            j = b.add_joint_d6(-1, link, linear_axes=[], angular_axes=[D(axis=0), D(axis=1), D(axis=2)],
                               parent_xform=[0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0])
        b.add_articulation([j])
        model = b.finalize()
        o = Oracle(model)
        s0, s1 = OracleState(model), OracleState(model)
        s0.joint_qd[:3] = w0  # at q = 0 the transported axes are X, Y, Z: joint speeds == angular velocity
        for _ in range(2500):
            s0.body_f[:] = 0
            o.featherstone_step(s0, s1, o.control(), None, 2e-4, angular_damping=0.0)
            s0, s1 = s1, s0
        out[kind] = (s0.body_q[0].copy(), s0.body_qd[0].copy())
    qa, qb = out["ball"][0][3:], out["d6"][0][3:]
    assert min(np.abs(qa - qb).max(), np.abs(qa + qb).max()) < 5e-4
    assert np.abs(out["ball"][1][3:] - out["d6"][1][3:]).max() < 3e-3
    assert np.linalg.norm(out["ball"][1][3:]) > 1.0  # still spinning


def test_free_joint_below_a_swinging_link_moves_like_a_free_body(oracle_lib):
    """A FREE joint transmits nothing: the body behind one, hung below a swinging pendulum link, must fly exactly like a
    stand-alone free body with the same initial twist (descendant path of solver_featherstone.py:229-265,1006-1046: internal
    parent-origin coordinates, pose re-integrated from the world COM twist, joint_q rebuilt from the poses)."""
    X = lambda p, q=(0.0, 0.0, 0.0, 1.0): nm.transform(p, q)  # noqa: E731
    cfg = nt.ModelBuilder.ShapeConfig(has_shape_collision=False)

    def build(attached):
        b = nt.ModelBuilder()
        if attached:
            arm = b.add_link(xform=[0.3, 0.0, 2.0, *I4])
            b.add_shape_box(arm, hx=0.3, hy=0.03, hz=0.03, cfg=cfg)
        body = b.add_link(xform=[0.9, 0.1, 1.8, *nm.quat_rpy(0.3, -0.2, 0.5)])
        b.add_shape_box(body, xform=X([0.02, -0.01, 0.03]), hx=0.1, hy=0.06, hz=0.04, cfg=cfg)
        if attached:
            j0 = b.add_joint_revolute(-1, arm, axis=[0.0, 1.0, 0.0], parent_xform=X([0.0, 0.0, 2.0]), child_xform=X([-0.3, 0.0, 0.0]))
            j1 = b.add_joint_free(body, parent=arm, parent_xform=X([0.3, 0.0, 0.0], nm.quat_rpy(0.1, 0.2, -0.3)),
                                  child_xform=X([-0.02, 0.03, 0.0]))
            b.add_articulation([j0, j1])
        else:
            b.add_articulation([b.add_joint_free(body)])
        return b.finalize()

    twist = np.array([0.4, -0.3, 1.0, 0.8, -1.1, 0.6], np.float32)  # COM velocity, angular velocity (world)
    tracks = []
    for attached, dt, n in ((True, 1e-3, 300), (False, 1e-3, 300), (True, 2.5e-4, 1200), (False, 2.5e-4, 1200)):
        m = build(attached)
        k = 1 if attached else 0
        s0, s1 = OracleState(m), OracleState(m)
        if attached:
            # joint_qd of a FREE joint (public convention): the child's COM twist relative to the parent, in the parent anchor frame
            from newton_amd.articulation import _qinv, _qrot

            w_arm, pivot = np.array([0.0, 1.5, 0.0]), np.array([0.0, 0.0, 2.0])  # the arm swings
            q_anchor = np.array(nm.quat_rpy(0.1, 0.2, -0.3))  # arm orientation is the identity at t = 0
            x_com = s0.body_q[k, :3] + _qrot(s0.body_q[k, 3:], np.asarray(m.body_com).reshape(-1, 3)[k])
            s0.joint_qd[0] = w_arm[1]
            s0.joint_qd[1:4] = _qrot(_qinv(q_anchor), twist[:3] - np.cross(w_arm, x_com - pivot))
            s0.joint_qd[4:7] = _qrot(_qinv(q_anchor), twist[3:] - w_arm)
        else:
            s0.joint_qd[:] = twist  # root FREE joint: COM twist in the world frame
        bq, bqd = nt.articulation.eval_fk_numpy(m, s0.joint_q, s0.joint_qd)
        assert np.abs(bq - s0.body_q).max() < 1e-6 and np.abs(bqd[k] - twist).max() < 1e-5
        s0.body_qd[:] = bqd
        o = Oracle(m)
        s0, _ = _run(o, s0, s1, o.control(), n, dt)
        tracks.append((s0.body_q[k].copy(), s0.body_qd[k].copy()))
    # the internal coordinates live in the rotating parent frame and the integrator is first order: the two flights agree to
    # O(dt) (2.0 mm after 0.3 s and 0.15 m of travel at dt = 1 ms, measured) and the gap closes with dt
    gaps = []
    for (q_a, qd_a), (q_f, qd_f) in (tracks[:2], tracks[2:]):
        assert np.abs(q_f[:3] - np.array([0.9, 0.1, 1.8])).max() > 0.1  # it went somewhere
        gaps.append((np.abs(q_a[:3] - q_f[:3]).max(), np.abs(qd_a[:3] - qd_f[:3]).max(),
                     min(np.abs(q_a[3:] - q_f[3:]).max(), np.abs(q_a[3:] + q_f[3:]).max()), np.abs(qd_a[3:] - qd_f[3:]).max()))
    print("free body vs body behind a FREE joint (pos, vel, rot, ang vel) at dt = 1 ms / 0.25 ms:", gaps)
    assert gaps[0][0] < 4e-3 and gaps[0][1] < 2e-2 and gaps[0][2] < 1e-2 and gaps[0][3] < 5e-2
    assert gaps[1][0] < 0.4 * gaps[0][0] and gaps[1][0] < 1e-3
