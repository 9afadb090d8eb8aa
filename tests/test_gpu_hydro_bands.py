"""Margin / gap bands of the hydroelastic contacts through CollisionPipeline.collide, the reference's tests
(newton/tests/test_hydroelastic.py:1038-1080 scene + distances, :1165-1272): two hydroelastic boxes with asymmetric margins
(0.05 / 0.07) and gaps (0.03 / 0.05); depending on the real surface separation the contacts are penetrating (margin-inflated
surfaces overlap), speculative (within the gap, distance >= 0, positive stiffness) or absent; with zero gaps the speculative band
does not exist."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _boxes(z_b, gaps=(0.03, 0.05), device="cuda:0"):
    import newton_amd as nt

    b = nt.ModelBuilder()

    def cfg(margin, gap, kh):
        c = b.default_shape_cfg.copy()
        c.margin, c.gap = margin, gap
        c.configure_sdf(max_resolution=64, is_hydroelastic=True, kh=kh)
        c.sdf_narrow_band_range = (-0.25, 0.25)
        return c

    body_a = b.add_body(xform=[0.0, 0.0, 0.0, 0, 0, 0, 1])
    body_b = b.add_body(xform=[0.0, 0.0, z_b, 0, 0, 0, 1])
    b.add_shape_box(body_a, hx=0.5, hy=0.5, hz=0.5, cfg=cfg(0.05, gaps[0], 1.0e8))
    b.add_shape_box(body_b, hx=0.5, hy=0.5, hz=0.5, cfg=cfg(0.07, gaps[1], 2.0e8))
    return b.finalize(device=device)


def _collide(model, reduce_contacts):
    import newton_amd as nt

    config = nt.geometry.HydroelasticSDF.Config(reduce_contacts=reduce_contacts, pre_prune_contacts=reduce_contacts)
    pipe = nt.CollisionPipeline(model, broad_phase="explicit", sdf_hydroelastic_config=config, sdf_contacts_per_shape=20000,
                                sdf_hydro_faces_per_shape=20000)
    contacts, state = pipe.contacts(), model.state()
    pipe.collide(state, contacts)
    assert not pipe._sdf_leg.overflow(contacts._flat)["overflow"]
    return contacts, state


def _distances(contacts, model, state):
    n = int(contacts.rigid_contact_count.item())
    if n == 0:
        return np.empty(0)
    g = lambda a: a.cpu().numpy()[:n]  # noqa: E731
    shape_body, body_q = np.asarray(model.shape_body), state.body_q.cpu().numpy()
    b0, b1 = shape_body[g(contacts.rigid_contact_shape0)], shape_body[g(contacts.rigid_contact_shape1)]
    off0 = np.where((b0 != -1)[:, None], body_q[np.maximum(b0, 0), :3], 0.0)
    off1 = np.where((b1 != -1)[:, None], body_q[np.maximum(b1, 0), :3], 0.0)
    return np.einsum("ij,ij->i", g(contacts.rigid_contact_point1) + off1 - g(contacts.rigid_contact_point0) - off0,
                     g(contacts.rigid_contact_normal))


@pytest.mark.parametrize("reduce_contacts", [False, True])
def test_hydroelastic_margin_gap_bands(reduce_contacts):
    gap_sum = 0.08
    for separation, expected, expect_contacts in ((0.08, -0.04, True), (0.12, 0.0, True), (0.16, 0.04, True), (0.24, 0.12, False)):
        model = _boxes(1.0 + separation)
        tol = 2.0 * float(max(np.max(t.voxel_size) for t in model._texture_sdf_data))
        contacts, state = _collide(model, reduce_contacts)
        d = _distances(contacts, model, state)
        if not expect_contacts:
            assert len(d) == 0
            continue
        assert len(d) > 0, separation
        assert abs(d.min() - expected) <= tol, (separation, d.min(), d.max())
        assert np.all(d <= gap_sum + tol)
        assert np.all(contacts.rigid_contact_stiffness.cpu().numpy()[: len(d)] > 0.0)
        if expected >= 0.0:
            assert np.all(d >= -tol)
            if expected > tol:
                assert np.all(d >= 0.0)  # speculative contacts must not move into the penetrating margin region


@pytest.mark.parametrize("reduce_contacts", [False, True])
def test_hydroelastic_zero_gap_omits_speculative_contacts(reduce_contacts):
    # real surfaces 0.16 m apart, margin-inflated surfaces 0.04 m apart: a non-zero gap would give speculative contacts
    contacts, _ = _collide(_boxes(1.16, gaps=(0.0, 0.0)), reduce_contacts)
    assert int(contacts.rigid_contact_count.item()) == 0
    model = _boxes(1.08, gaps=(0.0, 0.0))  # zero gap must not disable penetrating contacts
    contacts, state = _collide(model, reduce_contacts)
    d = _distances(contacts, model, state)
    assert len(d) > 0 and np.all(d < 0.0)
