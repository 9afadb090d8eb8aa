"""State.body_parent_f known answers, restated from newton/tests/test_parent_force.py:25-200 (Featherstone: one-step
static pendulum, centrifugal term, wrench propagation on a 2-link chain) and newton/tests/test_solver_xpbd.py:1137-1300
(XPBD: time-averaged steady state of a single suspended body).  Oracle on the CPU, HIP path on the GPU."""
import numpy as np
import pytest

import newton_amd as nt
from newton_amd import _np_math as nm

I4 = [0.0, 0.0, 0.0, 1.0]
BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


class _Sim:
    """solver.step loop with ping-pong states on either backend; ``parent_f()`` reads state.body_parent_f of the last output."""

    def __init__(self, model, solver, backend, **kw):
        self.model, self.kind, self.backend, self.kw = model, solver, backend, kw
        if backend == "oracle":
            from oracle_bridge import Oracle, OracleState

            self.o = Oracle(model)
            self.s0, self.s1 = OracleState(model), OracleState(model)
            assert self.s0.body_parent_f is not None
            bq, bqd = self.o.eval_fk(model.joint_q, model.joint_qd)
            self.s0.body_q[:], self.s0.body_qd[:] = bq, bqd
        else:
            cls = nt.solvers.SolverFeatherstone if solver == "featherstone" else nt.solvers.SolverXPBD
            self.solver = cls(model, **kw)
            self.s0, self.s1 = model.state(), model.state()
            assert self.s0.body_parent_f is not None
            nt.eval_fk(model, model.joint_q, model.joint_qd, self.s0)

    def set_joint_qd(self, qd):
        qd = np.asarray(qd, dtype=np.float32)
        if self.backend == "oracle":
            self.s0.joint_qd[:] = qd
            bq, bqd = self.o.eval_fk(self.s0.joint_q, qd)
            self.s0.body_q[:], self.s0.body_qd[:] = bq, bqd
        else:
            self.s0.joint_qd = qd
            nt.eval_fk(self.model, self.s0.joint_q, self.s0.joint_qd, self.s0)

    def set_body_f(self, f):
        if self.backend == "oracle":
            self.s0.body_f[:] = f
        else:
            self.s0.body_f = f

    def step(self, dt):
        if self.backend == "oracle":
            if self.kind == "featherstone":
                self.o.featherstone_step(self.s0, self.s1, self.o.control(), None, dt)
            else:
                self.o.xpbd_step(self.s0, self.s1, self.o.control(), None, dt, **self.kw)
        else:
            self.solver.step(self.s0, self.s1, None, None, dt)
        self.s0, self.s1 = self.s1, self.s0

    def parent_f(self):
        f = self.s0.body_parent_f
        return np.array(f if self.backend == "oracle" else f.cpu().numpy(), dtype=np.float64)


def _pendulum(device, joint_axis, child_offset, parent_xform=None):
    b = nt.ModelBuilder(gravity=(0.0, 0.0, -9.81))
    b.request_state_attributes("body_parent_f")
    link = b.add_link()
    b.add_shape_box(link, hx=0.1, hy=0.1, hz=0.1)
    j = b.add_joint_revolute(-1, link, parent_xform=parent_xform, child_xform=[*child_offset, *I4], axis=joint_axis)
    b.add_articulation([j])
    return b.finalize(device=device)


def _device(backend):
    return "cuda:0" if backend == "hip" else None


@pytest.mark.parametrize("backend", BACKENDS)
def test_featherstone_static_pendulum(oracle_lib, backend):
    xforms = [None, [5, 3, -2, *I4], [1, 2, 3, *nm.quat_from_axis_angle([1.0, 0.0, 0.0], np.pi * 0.5)]]
    for xf in xforms:
        model = _pendulum(_device(backend), (0, 1, 0), (0, 0, 1), xf)
        sim = _Sim(model, "featherstone", backend)
        sim.step(5e-3)
        f = sim.parent_f()[0]
        weight = float(np.asarray(model.body_mass)[0]) * 9.81
        np.testing.assert_allclose(f[:3], [0, 0, weight], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(f[3:], 0.0, atol=1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_featherstone_centrifugal(oracle_lib, backend):
    r, omega = 1.0, 5.0
    model = _pendulum(_device(backend), (0, 0, 1), (-r, 0, 0))
    sim = _Sim(model, "featherstone", backend)
    sim.set_joint_qd([omega])
    sim.step(5e-3)
    f = sim.parent_f()[0]
    mass = float(np.asarray(model.body_mass)[0])
    np.testing.assert_allclose(f[:3], [-mass * omega**2 * r, 0.0, mass * 9.81], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(f[3:], 0.0, atol=1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_featherstone_body_f_propagates_up_the_chain(oracle_lib, backend):
    b = nt.ModelBuilder(gravity=(0.0, 0.0, -9.81))
    b.request_state_attributes("body_parent_f")
    l0 = b.add_link()
    b.add_shape_box(l0, hx=0.1, hy=0.1, hz=0.1)
    j0 = b.add_joint_revolute(-1, l0, child_xform=[0, 0, 1, *I4], axis=(0, 1, 0))
    l1 = b.add_link()
    b.add_shape_box(l1, hx=0.1, hy=0.1, hz=0.1)
    j1 = b.add_joint_revolute(l0, l1, parent_xform=[0, 0, -1, *I4], child_xform=[0, 0, 1, *I4], axis=(0, 1, 0))
    b.add_articulation([j0, j1])
    model = b.finalize(device=_device(backend))
    total_weight = float(np.asarray(model.body_mass)[:2].sum()) * 9.81

    for case, idx, mag, lin, tor in (("force", 1, 10.0, [0, -10.0, total_weight], [-20.0, 0, 0]),
                                     ("torque", 3, 5.0, [0, 0, total_weight], [-5.0, 0, 0])):
        sim = _Sim(model, "featherstone", backend)
        body_f = np.zeros((2, 6), dtype=np.float32)
        body_f[1, idx] = mag
        sim.set_body_f(body_f)
        sim.step(5e-3)
        f = sim.parent_f()[0]
        np.testing.assert_allclose(f[:3], lin, rtol=1e-4, atol=1e-3, err_msg=case)
        np.testing.assert_allclose(f[3:], tor, atol=1e-2, err_msg=case)


def _suspended_body(device, joint_kind, parent_kinematic):
    b = nt.ModelBuilder(gravity=(0.0, 0.0, -9.81))
    b.request_state_attributes("body_parent_f")
    parent = -1
    if parent_kinematic:
        parent = b.add_body()
        b.add_shape_box(parent, hx=0.05, hy=0.05, hz=0.05)
        b.body_flags[parent] = int(nt.BodyFlags.KINEMATIC)
    child = b.add_link()
    b.add_shape_box(child, hx=0.1, hy=0.1, hz=0.1)
    kw = dict(child_xform=[0.0, 0.0, 1.0, *I4])
    if joint_kind == "revolute":
        j = b.add_joint_revolute(parent, child, axis=(0.0, 1.0, 0.0), **kw)
    elif joint_kind == "ball":
        j = b.add_joint_ball(parent, child, **kw)
    else:
        j = b.add_joint_fixed(parent, child, **kw)
    b.add_articulation([j])
    return b.finalize(device=device), child


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("joint_kind,parent_kinematic", [("revolute", False), ("revolute", True), ("ball", False),
                                                         ("ball", True), ("fixed", False)])
def test_xpbd_parent_force_steady_state(oracle_lib, backend, joint_kind, parent_kinematic):
    """test_solver_xpbd.py:1247-1300: one dynamic body under a world / kinematic parent: reaction = weight within 1 %."""
    model, child = _suspended_body(_device(backend), joint_kind, parent_kinematic)
    sim = _Sim(model, "xpbd", backend, iterations=8)
    sub_dt, substeps = 1.0 / 60.0 / 8, 8
    for _ in range(60 * substeps):
        sim.step(sub_dt)
    avg = np.zeros(6)
    for _ in range(30):
        for _ in range(substeps):
            sim.step(sub_dt)
        avg += sim.parent_f()[child]
    avg /= 30
    weight = float(np.asarray(model.body_mass)[child]) * 9.81
    np.testing.assert_allclose(avg[2], weight, rtol=0.01)
    np.testing.assert_allclose(avg[:2], 0.0, atol=0.1)
    np.testing.assert_allclose(avg[3:], 0.0, atol=0.1)
    if parent_kinematic:  # roots / bodies without an inbound joint report zero (solver_xpbd.py:736-741)
        assert np.all(sim.parent_f()[0] == 0.0)


def test_state_attribute_requests():
    b = nt.ModelBuilder()
    with pytest.raises(ValueError):
        b.request_state_attributes("no_such_attribute")
    link = b.add_body()
    b.add_shape_sphere(link, radius=0.1)
    m = b.finalize()
    assert m.get_requested_state_attributes() == []
    m.request_state_attributes("body_parent_f")
    m.request_contact_attributes("force")
    assert m.get_requested_state_attributes() == ["body_parent_f"] and m.get_requested_contact_attributes() == {"force"}
