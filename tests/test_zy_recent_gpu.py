"""Device checks for the features added after the last on-hardware run of the suite (they are covered on the CPU by the emulated
kernels, tests/test_emu_parity.py and tests/emu/run_gpu_tests_emulated.py); kept in one late-sorting file so that the order of
the on-device run is: long-validated tests first, newest last."""
import numpy as np
import pytest

from pair_scenes import RECENT_CONVEX_CASES
from test_gpu_parity_xpbd import _compare_contacts, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(RECENT_CONVEX_CASES))
def test_recent_convex_pair_contacts(name):
    """Cylinder / cone rolling on a hull face: the axial post-processing treats convex meshes as discrete shapes
    (collide.py contact reduction for curved-vs-polyhedral pairs)."""
    import newton_amd as nt
    from oracle_bridge import Oracle
    from pair_scenes import CONVEX_CASES, pair_model

    model = pair_model(CONVEX_CASES[name], device="cuda:0")
    o = Oracle(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    pipe.collide(model.state(), contacts)
    oc = o.contacts()
    pairs, _, _ = o.collide(model.body_q, oc)
    _compare_contacts(model, contacts, oc, pairs)


def test_deterministic_mode_orders_contacts_by_sort_key():
    """CollisionPipeline(deterministic=True): the flat contact arrays are sorted by the reference's contact key
    (shape0, shape1, sub-contact index: contact_data.py:60-90); same contacts as the default append order."""
    from scenes import mixed_primitive_scene

    nt, model, o = _setup(mixed_primitive_scene, 6)
    s0 = model.state()
    plain, det = nt.CollisionPipeline(model), nt.CollisionPipeline(model, deterministic=True)
    c0, c1 = plain.contacts(), det.contacts()
    plain.collide(s0, c0)
    det.collide(s0, c1)
    n = int(c0.rigid_contact_count.cpu().numpy()[0])
    assert n > 20 and int(c1.rigid_contact_count.cpu().numpy()[0]) == n
    a0, a1 = c0.rigid_contact_shape0.cpu().numpy()[:n], c0.rigid_contact_shape1.cpu().numpy()[:n]
    order = np.lexsort((np.arange(n), a1, a0))  # stable: sub-contact order inside a pair is the export order
    assert not np.array_equal(order, np.arange(n))  # the append order (analytic first, then convex) is a different order
    for name in ("shape0", "shape1", "point0", "point1", "normal", "margin0"):
        want = getattr(c0, "rigid_contact_" + name).cpu().numpy()[:n][order]
        assert np.array_equal(getattr(c1, "rigid_contact_" + name).cpu().numpy()[:n], want), name
    key = c1.rigid_contact_shape0.cpu().numpy()[:n].astype(np.int64) * (1 << 20) + c1.rigid_contact_shape1.cpu().numpy()[:n]
    assert np.all(np.diff(key) >= 0)


def test_eval_fk_selection_on_device():
    """Same selections through the device path: full FK into a scratch State (one eval_fk_kernel launch), then the selected
    body rows are copied."""
    import newton_amd as nt

    env = nt.ModelBuilder()
    for k in range(2):
        link = env.add_link(is_kinematic=(k == 1), mass=1.0)
        env.add_shape_box(link, hx=0.1, hy=0.1, hz=0.1)
        j = env.add_joint_revolute(-1, link, axis=(0, 0, 1), parent_xform=[float(k), 0.0, 0.0, 0.0, 0.0, 0.0, 1.0],
                                   child_xform=[-0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
        env.add_articulation([j])
    scene = nt.ModelBuilder()
    scene.replicate(env, 5)
    model = scene.finalize(device="cuda:0")
    rng = np.random.default_rng(0)
    q = rng.uniform(-1.0, 1.0, size=10).astype(np.float32)
    qd = rng.normal(size=10).astype(np.float32)
    full_q, full_qd = nt.articulation.eval_fk_numpy(model, q, qd)
    base = model.state()
    base_q = base.body_q.cpu().numpy().copy()
    mask = rng.random(10) < 0.5
    mask[0], mask[1] = True, False
    for kw, sel in ((dict(mask=mask), mask), (dict(indices=np.flatnonzero(mask)), mask),
                    (dict(body_flag_filter=nt.BodyFlags.KINEMATIC), np.arange(10) % 2 == 1)):
        s = model.state()
        nt.eval_fk(model, q, qd, s, **kw)
        got_q, got_qd = s.body_q.cpu().numpy(), s.body_qd.cpu().numpy()
        assert np.allclose(got_q[sel], full_q[sel], atol=1e-6) and np.allclose(got_qd[sel], full_qd[sel], atol=1e-5)
        assert np.array_equal(got_q[~sel], base_q[~sel])
