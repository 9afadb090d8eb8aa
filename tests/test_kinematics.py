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
