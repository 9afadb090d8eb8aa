"""Contacts.force known answers, restated from newton/tests/test_solver_xpbd.py:845-1135: after settling, the reported
per-contact forces sum to each body's weight (sphere, heavy sphere, 4-contact box, 3-cube pyramid: 1.5 mg under each
bottom cube); zero force in free fall and for a contact pair inside the gap but not touching; update_contacts misuse
raises.  Oracle on the CPU, HIP path on the GPU."""
import numpy as np
import pytest

import newton_amd as nt

I4 = [0.0, 0.0, 0.0, 1.0]
BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


class _Sim:
    """{clear_forces; collide; step; swap} + update_contacts on either backend; ``contacts()`` -> (shape0, shape1, force[n,6])."""

    def __init__(self, model, backend, **kw):
        self.model, self.backend, self.kw = model, backend, kw
        if backend == "oracle":
            from oracle_bridge import Oracle, OracleState

            self.o = Oracle(model)
            self.s0, self.s1 = OracleState(model), OracleState(model)
            self.ct = self.o.contacts()
            self.force = np.zeros((max(self.ct.max, 1), 6), dtype=np.float32)
        else:
            self.solver = nt.solvers.SolverXPBD(model, **kw)
            self.pipe = nt.CollisionPipeline(model)
            self.ct = self.pipe.contacts()
            self.s0, self.s1 = model.state(), model.state()

    def step(self, dt):
        if self.backend == "oracle":
            self.s0.body_f[:] = 0
            self.o.collide(self.s0.body_q, self.ct)
            self.o.xpbd_step(self.s0, self.s1, self.o.control(), self.ct, dt, contact_force_out=self.force, **self.kw)
        else:
            self.s0.clear_forces()
            self.pipe.collide(self.s0, self.ct)
            self.solver.step(self.s0, self.s1, None, self.ct, dt)
        self.s0, self.s1 = self.s1, self.s0

    def contacts(self):
        if self.backend == "oracle":
            n = int(self.ct.count[0])
            return self.ct.shape0[:n].copy(), self.ct.shape1[:n].copy(), self.force[:n].astype(np.float64)
        self.solver.update_contacts(self.ct, self.s0)
        n = int(self.ct.rigid_contact_count.cpu().numpy()[0])
        return (self.ct.rigid_contact_shape0.cpu().numpy()[:n], self.ct.rigid_contact_shape1.cpu().numpy()[:n],
                self.ct.force.cpu().numpy()[:n].astype(np.float64))


def _device(backend):
    return "cuda:0" if backend == "hip" else None


@pytest.mark.parametrize("backend", BACKENDS)
def test_contact_forces_sum_to_weight(oracle_lib, backend):
    g = 9.81
    sphere_r, heavy_r, h = 0.25, 0.5, 0.5
    sphere_mass = 1000.0 * (4.0 / 3.0) * np.pi * sphere_r**3
    heavy_mass = 2000.0 * (4.0 / 3.0) * np.pi * heavy_r**3
    box_mass = cube_mass = 1000.0 * (2.0 * h) ** 3
    b = nt.ModelBuilder()
    ground = b.add_ground_plane()
    b.default_shape_cfg.density = 1000.0
    sphere = b.add_body(xform=[0.0, 0.0, sphere_r, *I4])
    b.add_shape_sphere(sphere, radius=sphere_r)
    b.default_shape_cfg.density = 2000.0
    heavy = b.add_body(xform=[10.0, 0.0, heavy_r, *I4])
    b.add_shape_sphere(heavy, radius=heavy_r)
    b.default_shape_cfg.density = 1000.0
    box = b.add_body(xform=[20.0, 0.0, h, *I4])
    b.add_shape_box(box, hx=h, hy=h, hz=h)
    # The reference places the two bottom cubes EXACTLY face to face (test_solver_xpbd.py:898-902).  In that configuration its MPR / GJK
    # returns, for some states, one contact 0.65 m from the centres of these 0.5 m cubes with a normal tilted by 18 degrees -- a 0.3 m
    # "penetration" that throws the cubes apart at 60 m/s.  Reference behaviour, pinned bit for bit by test_touching_cubes_* below; whether
    # a 260-frame run meets such a state depends on its rounding (the HIP path of round 5 walked past it, round 6's hits it in frame 110).
    # The known answer tested here -- 1.5 m g under each bottom cube -- does not involve that pair (its healthy contacts carry +-50 N of
    # 15 000): the two cubes keep the reference's positions and are excluded from colliding with each other.
    left = b.add_body(xform=[30.0 - h, 0.0, h, *I4])
    s_left = b.add_shape_box(left, hx=h, hy=h, hz=h)
    right = b.add_body(xform=[30.0 + h, 0.0, h, *I4])
    s_right = b.add_shape_box(right, hx=h, hy=h, hz=h)
    b.add_shape_collision_filter_pair(s_left, s_right)
    top = b.add_body(xform=[30.0, 0.0, 3.0 * h, *I4])
    b.add_shape_box(top, hx=h, hy=h, hz=h)
    b.request_contact_attributes("force")
    model = b.finalize(device=_device(backend))
    shape_body = np.asarray(model.shape_body)

    sim = _Sim(model, backend, iterations=32, rigid_contact_con_weighting=True)
    sub_dt, substeps, settle, avg_steps = 1.0 / 60.0 / 8, 8, 200, 60
    for _ in range(settle * substeps):
        sim.step(sub_dt)
    on_ground = {k: np.zeros(3) for k in (sphere, heavy, box, left, right)}
    for _ in range(avg_steps):
        for _ in range(substeps):
            sim.step(sub_dt)
        s0, s1, force = sim.contacts()
        n_box = 0
        for a, c, f in zip(s0, s1, force[:, :3]):
            # contacts.force is the force on shape0's body by shape1's; fold into "force on the ground"
            if a == ground:
                other = c
            elif c == ground:
                other, f = a, -f
            else:
                continue
            body = shape_body[other]
            if body in on_ground:
                on_ground[body] += f
            n_box += body == box
        assert n_box > 1, "the box must rest on several contact points"
    for k in on_ground:
        on_ground[k] /= avg_steps
    np.testing.assert_allclose(on_ground[sphere][2], -sphere_mass * g, rtol=0.05)
    np.testing.assert_allclose(on_ground[sphere][:2], 0.0, atol=0.5)
    np.testing.assert_allclose(on_ground[heavy][2], -heavy_mass * g, rtol=0.05)
    np.testing.assert_allclose(on_ground[heavy][:2], 0.0, atol=0.5)
    np.testing.assert_allclose(on_ground[box][2], -box_mass * g, rtol=0.10)  # mg, not N_contacts * mg
    np.testing.assert_allclose(on_ground[box][:2], 0.0, atol=1.0)
    np.testing.assert_allclose(-on_ground[left][2], 1.5 * cube_mass * g, rtol=0.15)
    np.testing.assert_allclose(-on_ground[right][2], 1.5 * cube_mass * g, rtol=0.15)


@pytest.mark.parametrize("backend", BACKENDS)
def test_no_force_without_penetration(oracle_lib, backend):
    # free fall, no ground
    b = nt.ModelBuilder()
    body = b.add_body(xform=[0.0, 0.0, 5.0, *I4])
    b.add_shape_sphere(body, radius=0.25)
    b.request_contact_attributes("force")
    # a second free body far away so that the model has a (never colliding) pair
    other = b.add_body(xform=[50.0, 0.0, 5.0, *I4])
    b.add_shape_sphere(other, radius=0.25)
    sim = _Sim(b.finalize(device=_device(backend)), backend, iterations=2)
    sim.step(1.0 / 60.0)
    assert np.all(sim.contacts()[2] == 0.0)

    # inside the gap but not touching: a contact exists, its force is zero
    b = nt.ModelBuilder(gravity=0.0)
    b.default_shape_cfg.gap = 1.0
    b.add_ground_plane()
    body = b.add_body(xform=[0.0, 0.0, 0.25 + 0.5, *I4])
    b.add_shape_sphere(body, radius=0.25)
    b.request_contact_attributes("force")
    sim = _Sim(b.finalize(device=_device(backend)), backend, iterations=2)
    sim.step(1.0 / 60.0)
    s0, _, force = sim.contacts()
    assert len(s0) > 0
    np.testing.assert_allclose(force, 0.0, atol=1e-6)


@pytest.mark.gpu
def test_update_contacts_misuse_raises():
    b = nt.ModelBuilder()
    b.add_ground_plane()
    body = b.add_body(xform=[0.0, 0.0, 0.25, *I4])
    b.add_shape_sphere(body, radius=0.25)
    model = b.finalize(device="cuda:0")
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s0, s1 = model.state(), model.state()
    pipe.collide(s0, contacts)
    solver.step(s0, s1, None, contacts, 1.0 / 60.0)
    assert contacts.force is None
    with pytest.raises(ValueError):
        solver.update_contacts(contacts)
    model.request_contact_attributes("force")
    contacts = pipe.contacts()
    assert contacts.force is not None
    with pytest.raises(ValueError):  # no step has filled the impulses yet
        nt.solvers.SolverXPBD(model, iterations=2).update_contacts(contacts)


def _touching_cubes():
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_touching_cubes_vectors as mk

    return mk, np.load(os.path.join(here, "golden", "touching_cubes_reference_vectors.npz"))


def _assert_touching_cubes(ref, count, arrays):
    n = int(ref["count"][0])
    assert count == n
    assert np.array_equal(arrays["shape0"][:n], ref["shape0"]) and np.array_equal(arrays["shape1"][:n], ref["shape1"])
    for k in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
        assert np.max(np.abs(arrays[k][:n] - ref[k])) <= 2e-6, k
    i = [k for k in range(n) if (ref["shape0"][k], ref["shape1"][k]) == (4, 5)]
    assert len(i) == 1 and abs(float(ref["point0"][i[0]][0])) > 0.6  # the degenerate contact: 0.65 m from the centre of a 0.5 m cube


def test_touching_cubes_degenerate_contact_is_reference_behaviour(oracle_lib):
    """tests/golden/make_touching_cubes_vectors.py: the reference's own collision kernels, executed, on a state of this file's scene with
    the bottom cubes exactly face to face -- one bogus contact 0.65 m from the cube centres.  The checker reproduces it."""
    from oracle_bridge import Oracle

    mk, ref = _touching_cubes()
    model = mk.scene()
    o = Oracle(model)
    oc = o.contacts()
    pairs, _, _ = o.collide(ref["body_q"], oc)
    assert np.array_equal(np.asarray(pairs, np.int32), ref["pairs"])
    _assert_touching_cubes(ref, int(oc.count[0]), {k: getattr(oc, k) for k in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")})


@pytest.mark.gpu
def test_touching_cubes_degenerate_contact_on_device():
    """... and so does CollisionPipeline.collide on the device, from the same state."""
    import torch

    mk, ref = _touching_cubes()
    model = mk.scene(device="cuda:0")
    pipe = nt.CollisionPipeline(model)
    ct = pipe.contacts()
    s = model.state()
    s.body_q = torch.from_numpy(ref["body_q"])
    pipe.collide(s, ct)
    n = int(ct.rigid_contact_count.cpu().numpy()[0])
    _assert_touching_cubes(ref, n, {k: getattr(ct, "rigid_contact_" + k).cpu().numpy() for k in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")})
