"""Reduced vs unreduced hydroelastic contacts through CollisionPipeline.collide, the reference's own acceptance tests
(newton/tests/test_hydroelastic.py:904-1035 helpers, :1412-1516): a hydroelastic sphere pressed into a hydroelastic cube;
the net contact force of the reduced rows (aggregate stiffness, matched normals, optionally anchors) equals the force of all
marching-cubes faces within 1 %, and with moment matching the friction moment about the centre of pressure within 40 %."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(pen, device="cuda:0"):
    import newton_amd as nt

    cube_half, radius = 0.1, 0.1
    b = nt.ModelBuilder()
    cfg = b.default_shape_cfg
    cfg.gap = 0.01
    cfg.configure_sdf(max_resolution=32, is_hydroelastic=True, kh=1.0e9)
    cfg.sdf_narrow_band_range = (-0.01, 0.01)
    cube = b.add_body(xform=[0.0, 0.0, cube_half, 0, 0, 0, 1])
    b.add_shape_box(cube, hx=cube_half, hy=cube_half, hz=cube_half)
    rest_z = 2 * cube_half + radius
    ball = b.add_body(xform=[0.0, 0.0, rest_z - pen, 0, 0, 0, 1])
    b.add_shape_sphere(ball, radius=radius)
    return b.finalize(device=device)


def _forces(contacts, model, state):
    """_extract_contact_forces (test_hydroelastic.py:904-948)."""
    n = int(contacts.rigid_contact_count.item())
    if n == 0 or contacts.rigid_contact_stiffness is None:
        e3 = np.empty((0, 3))
        return e3, e3, e3, np.empty(0), np.empty(0)
    g = lambda a: a.cpu().numpy()[:n]  # noqa: E731
    normals, p0, p1 = g(contacts.rigid_contact_normal), g(contacts.rigid_contact_point0), g(contacts.rigid_contact_point1)
    stiffness, shape0, shape1 = g(contacts.rigid_contact_stiffness), g(contacts.rigid_contact_shape0), g(contacts.rigid_contact_shape1)
    shape_body = np.asarray(model.shape_body)
    body_q = state.body_q.cpu().numpy()
    b0, b1 = shape_body[shape0], shape_body[shape1]
    p0w = p0 + np.where((b0 != -1)[:, None], body_q[np.maximum(b0, 0), :3], 0.0)
    p1w = p1 + np.where((b1 != -1)[:, None], body_q[np.maximum(b1, 0), :3], 0.0)
    depth = np.einsum("ij,ij->i", p0w - p1w, -normals)
    mask = (stiffness > 0) & (depth < 0)
    friction = g(contacts.rigid_contact_friction)[mask]
    return p0w[mask], p1w[mask], normals[mask], stiffness[mask] * (-depth[mask]), np.where(friction > 0.0, friction, 1.0)


def _net_force(contacts, model, state):
    _, _, normals, f, _ = _forces(contacts, model, state)
    return np.sum(f[:, None] * (-normals), axis=0) if len(f) else np.zeros(3)


def _collide(model, cfg):
    import newton_amd as nt

    pipe = nt.CollisionPipeline(model, broad_phase="sap", sdf_hydroelastic_config=cfg, sdf_contacts_per_shape=4000,
                                sdf_hydro_faces_per_shape=4000)
    contacts = pipe.contacts()
    state = model.state()
    pipe.collide(state, contacts)
    assert not pipe._sdf_leg.overflow(contacts._flat)["overflow"]
    return contacts, state


@pytest.mark.parametrize("anchor_contact", [False, True])
def test_reduced_vs_unreduced_contact_forces(anchor_contact):
    import newton_amd as nt

    H = nt.geometry.HydroelasticSDF
    for pen in (0.0, 1e-3, 1e-2):
        model = _scene(pen)
        c_red, s_red = _collide(model, H.Config(reduce_contacts=True, anchor_contact=anchor_contact))
        c_unr, s_unr = _collide(model, H.Config(reduce_contacts=False))
        f_red, f_unr = _net_force(c_red, model, s_red), _net_force(c_unr, model, s_unr)
        if pen == 0.0:
            assert np.linalg.norm(f_red) < 1e-3 and np.linalg.norm(f_unr) < 1e-3
            continue
        assert int(c_red.rigid_contact_count.item()) < int(c_unr.rigid_contact_count.item())
        assert abs(f_unr[2]) > 1.0  # (its sign is the pair's a / b order: the finer SDF is shape B, sdf_hydroelastic.py:1362-1366)
        assert abs(f_red[2] - f_unr[2]) / abs(f_unr[2]) < 0.01, (pen, f_red, f_unr)
        for axis in (0, 1):
            assert abs(f_red[axis] - f_unr[axis]) / abs(f_unr[2]) < 0.01, (pen, axis, f_red, f_unr)


def test_reduced_vs_unreduced_contact_moments():
    import newton_amd as nt

    H = nt.geometry.HydroelasticSDF

    def moment(contacts, model, state, anchor):
        p0w, p1w, normals, f, friction = _forces(contacts, model, state)
        if len(f) == 0:
            return 0.0
        lever = np.linalg.norm(np.cross((p0w + p1w) / 2.0 - anchor, -normals), axis=1)
        return float((friction * f * lever).sum())

    for pen in (1e-3, 1e-2):
        model = _scene(pen)
        c_red, s_red = _collide(model, H.Config(reduce_contacts=True, anchor_contact=True, moment_matching=True))
        c_unr, s_unr = _collide(model, H.Config(reduce_contacts=False))
        p0w, p1w, _, f, _ = _forces(c_unr, model, s_unr)
        anchor = (f[:, None] * (p0w + p1w) / 2.0).sum(axis=0) / f.sum()
        m_red, m_unr = moment(c_red, model, s_red, anchor), moment(c_unr, model, s_unr, anchor)
        assert m_unr >= 0.0
        if m_unr > 1e-6:
            assert abs(m_red - m_unr) / m_unr < 0.4, (pen, m_red, m_unr)
