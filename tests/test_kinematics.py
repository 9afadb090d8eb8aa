"""Forward-kinematics invariants restated from newton/tests/test_kinematics.py:114-258: closed-form link positions of a
2-link planar arm (tol 1e-4) and body twists that agree with finite differences of the poses for a revolute chain with a
COM offset and an off-axis child anchor (tol 5e-3).  Run on the oracle FK, the vectorised host FK, and (GPU) nt_eval_fk."""
import numpy as np
import pytest

import newton_amd as nt
from newton_amd import _np_math as nm

I4 = [0.0, 0.0, 0.0, 1.0]


def _fk(backend, model, q, qd):
    if backend == "oracle":
        from oracle_bridge import Oracle

        return Oracle(model).eval_fk(q, qd)
    if backend == "numpy":
        return nt.articulation.eval_fk_numpy(model, q, qd)
    state = model.state()
    nt.eval_fk(model, q, qd, state)
    return state.body_q.cpu().numpy(), state.body_qd.cpu().numpy()


def _planar_arm(device=None):
    L1, L2 = 1.0, 0.8
    b = nt.ModelBuilder(up_axis=1, gravity=0.0)
    l0, l1 = b.add_link(), b.add_link()
    b.add_shape_sphere(l0, radius=0.01)
    b.add_shape_sphere(l1, radius=0.01)
    j0 = b.add_joint_revolute(-1, l0, axis=(0, 0, 1), parent_xform=[0, 0, 0, *I4], child_xform=[0, L1, 0, *I4])
    j1 = b.add_joint_revolute(l0, l1, axis=(0, 0, 1), parent_xform=[0, 0, 0, *I4], child_xform=[0, L2, 0, *I4])
    b.add_articulation([j0, j1])
    return b.finalize(device=device), L1, L2


def _check_planar(backend, device=None):
    model, L1, L2 = _planar_arm(device)
    for t1, t2 in [(0.0, 0.0), (0.3, 0.0), (0.0, -0.5), (np.pi / 4, np.pi / 4), (0.3, -0.2)]:
        q = np.array([t1, t2], dtype=np.float32)
        bq, _ = _fk(backend, model, q, np.zeros(2, dtype=np.float32))
        want0 = [L1 * np.sin(t1), -L1 * np.cos(t1), 0.0]
        want1 = [L1 * np.sin(t1) + L2 * np.sin(t1 + t2), -L1 * np.cos(t1) - L2 * np.cos(t1 + t2), 0.0]
        assert np.allclose(bq[0][:3], want0, atol=1e-4) and np.allclose(bq[1][:3], want1, atol=1e-4)


def _chain(prismatic, device=None):
    b = nt.ModelBuilder(up_axis=1, gravity=0.0)
    l0, l1 = b.add_link(mass=1.0, inertia=np.eye(3)), b.add_link(mass=1.0, inertia=np.eye(3))
    b.body_com[l0] = np.array([0.35, 0.0, 0.0])
    j0 = b.add_joint_revolute(-1, l0, axis=(0, 0, 1), parent_xform=[0, 0, 0, *I4], child_xform=[0, 0, 0, *I4])
    if prismatic:
        j1 = b.add_joint_prismatic(l0, l1, axis=(1, 0, 0), parent_xform=[1.0, 0, 0, *I4], child_xform=[0.2, 0.0, -0.15, *I4])
    else:
        j1 = b.add_joint_revolute(l0, l1, axis=(0, 0, 1), parent_xform=[1.0, 0, 0, *I4], child_xform=[0.2, 0.0, -0.15, *I4])
    b.add_articulation([j0, j1])
    return b.finalize(device=device)


def _check_fd(backend, prismatic, device=None):
    model = _chain(prismatic, device)
    q = np.array([0.7, -0.35], dtype=np.float32)
    qd = np.array([1.1, -0.45], dtype=np.float32)
    dt = 1.0e-4
    bq, bqd = _fk(backend, model, q, qd)
    bq2, _ = _fk(backend, model, (q + qd * dt).astype(np.float32), qd)
    tip = 1
    fd = (bq2[tip, :3].astype(np.float64) - bq[tip, :3]) / dt
    w = bqd[tip, 3:].astype(np.float64)
    v_origin = bqd[tip, :3] - np.cross(w, nm.quat_rotate(bq[tip, 3:], np.asarray(model.body_com[tip], dtype=np.float64)))
    assert np.allclose(fd, v_origin, atol=5e-3)


@pytest.mark.parametrize("backend", ["oracle", "numpy"])
def test_planar_arm_closed_form(oracle_lib, backend):
    _check_planar(backend)


@pytest.mark.parametrize("backend", ["oracle", "numpy"])
@pytest.mark.parametrize("prismatic", [False, True])
def test_descendant_velocity_matches_finite_difference(oracle_lib, backend, prismatic):
    _check_fd(backend, prismatic)


@pytest.mark.gpu
def test_planar_arm_closed_form_hip():
    _check_planar("hip", device="cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("prismatic", [False, True])
def test_descendant_velocity_matches_finite_difference_hip(prismatic):
    _check_fd("hip", prismatic, device="cuda:0")


def test_d6_with_two_and_three_angular_axes_fk(oracle_lib):
    """D6 joints with several angular axes compose intrinsic rotations about transported axes (compute_2d/3d_rotational_dofs,
    newton/_src/sim/articulation.py:36-83,127-178): for axes X, Y, Z the joint rotation is scipy's intrinsic 'XYZ' Euler
    rotation; body_qd is the time derivative of the pose (finite differences); host FK == oracle FK."""
    from scipy.spatial.transform import Rotation

    import newton_amd as nt
    from newton_amd import _np_math as nm
    from oracle_bridge import Oracle

    D = nt.ModelBuilder.JointDofConfig
    rng = np.random.default_rng(5)
    for ang_axes in ([0, 1], [2, 0], [0, 1, 2], [0, 2, 1]):
        b = nt.ModelBuilder()
        l0 = b.add_link()
        b.add_shape_box(l0, hx=0.1, hy=0.1, hz=0.1)
        l1 = b.add_link()
        b.add_shape_box(l1, hx=0.1, hy=0.1, hz=0.1)
        j0 = b.add_joint_fixed(-1, l0, parent_xform=[0.0, 0.0, 1.0, *nm.quat_rpy(0.3, 0.2, 0.1)])
        j1 = b.add_joint_d6(l0, l1, linear_axes=[D(axis=0)], angular_axes=[D(axis=a) for a in ang_axes],
                            parent_xform=[0.2, 0.0, 0.0, *nm.quat_rpy(0.1, -0.3, 0.2)], child_xform=[-0.1, 0.05, 0.0, 0.0, 0.0, 0.0, 1.0])
        b.add_articulation([j0, j1])
        model = b.finalize()
        n = len(ang_axes)
        q = rng.uniform(-0.8, 0.8, size=1 + n).astype(np.float32)
        qd = rng.normal(size=1 + n).astype(np.float32)
        bq, bqd = nt.articulation.eval_fk_numpy(model, q, qd)
        oq, oqd = Oracle(model).eval_fk(q, qd)
        assert np.max(np.abs(bq - oq)) <= 2e-6 and np.max(np.abs(bqd - oqd)) <= 2e-6
        # joint rotation == intrinsic Euler rotation about the listed axes
        X_wp = nm.transform_mul(bq[0].astype(np.float64), np.asarray(model.joint_X_p)[1].astype(np.float64))
        X_wc = nm.transform_mul(bq[1].astype(np.float64), np.asarray(model.joint_X_c)[1].astype(np.float64))
        rel = nm.transform_mul(nm.transform_inverse(X_wp), X_wc)
        want = Rotation.from_euler("".join("XYZ"[a] for a in ang_axes), q[1:].astype(np.float64)).as_quat()
        assert min(np.abs(rel[3:] - want).max(), np.abs(rel[3:] + want).max()) <= 1e-6
        assert np.allclose(rel[:3], [q[0], 0.0, 0.0], atol=1e-6)
        # velocities: finite difference of the child's COM position and orientation along qd
        h = 1e-3
        bq2, _ = nt.articulation.eval_fk_numpy(model, (q + h * qd).astype(np.float32), qd)
        com = np.asarray(model.body_com)[1].astype(np.float64)
        c0 = nm.transform_point(bq[1].astype(np.float64), com)
        c1 = nm.transform_point(bq2[1].astype(np.float64), com)
        assert np.allclose((c1 - c0) / h, bqd[1, :3], atol=5e-3)
        dq = nm.quat_mul(bq2[1, 3:].astype(np.float64), nm.quat_inverse(bq[1, 3:].astype(np.float64)))
        assert np.allclose(2.0 * dq[:3] / h, bqd[1, 3:], atol=5e-3)


def test_eval_fk_mask_indices_and_body_flag_filter():
    """eval_fk(..., mask= / indices= / body_flag_filter=) (newton/_src/sim/articulation.py:500-573): only the selected
    articulations' bodies, and only bodies whose flags match, are re-posed."""
    import newton_amd as nt

    b = nt.ModelBuilder()
    for k in range(3):  # three single-joint articulations; the middle one is kinematic
        link = b.add_link(is_kinematic=(k == 1), mass=1.0)
        b.add_shape_box(link, hx=0.1, hy=0.1, hz=0.1)
        j = b.add_joint_revolute(-1, link, axis=(0, 0, 1), parent_xform=[float(k), 0.0, 0.0, 0.0, 0.0, 0.0, 1.0],
                                 child_xform=[-0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
        b.add_articulation([j])
    model = b.finalize()
    q = np.array([0.3, -0.4, 0.5], dtype=np.float32)
    qd = np.zeros(3, dtype=np.float32)
    full_q, _ = nt.articulation.eval_fk_numpy(model, q, qd)
    base_q = np.array(model.state().body_q, dtype=np.float32).copy()
    assert not np.allclose(full_q, base_q)

    def run(**kw):
        s = model.state()
        nt.eval_fk(model, q, qd, s, **kw)
        return np.asarray(s.body_q)

    assert np.allclose(run(), full_q)
    got = run(mask=np.array([True, False, True]))
    assert np.allclose(got[[0, 2]], full_q[[0, 2]]) and np.allclose(got[1], base_q[1])
    got = run(indices=[1])
    assert np.allclose(got[1], full_q[1]) and np.allclose(got[[0, 2]], base_q[[0, 2]])
    got = run(body_flag_filter=nt.BodyFlags.KINEMATIC)
    assert np.allclose(got[1], full_q[1]) and np.allclose(got[[0, 2]], base_q[[0, 2]])
    import pytest as _pytest

    with _pytest.raises(ValueError):
        run(mask=np.array([True, False, True]), indices=[0])
