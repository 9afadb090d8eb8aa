"""The reference-held known answers of the convex narrow phase (tests/convex_known_answers.py), asserted on the HIP path
directly: pair scene -> nt_collide on the device -> exported contacts decoded to world space.  Independent of the oracle."""
import numpy as np
import pytest

import convex_known_answers as KA
from pair_scenes import CONVEX_CASES, decode_world, pair_model

pytestmark = pytest.mark.gpu


def _cs(name):
    import newton_amd as nt

    model = pair_model(CONVEX_CASES[name], device="cuda:0")
    pipe = nt.CollisionPipeline(model)
    ct = pipe.contacts()
    pipe.collide(model.state(), ct)
    n = int(ct.rigid_contact_count.cpu().numpy()[0])
    g = lambda k: getattr(ct, "rigid_contact_" + k).cpu().numpy()[:n]  # noqa: E731
    return decode_world(model, model.body_q, g("shape0"), g("shape1"), g("point0"), g("point1"), g("normal"), g("margin0"),
                        g("margin1"))


@pytest.mark.parametrize("check", KA.CHECKS, ids=lambda f: f.__name__)
def test_convex_known_answers_on_device(check):
    check(_cs)


@pytest.mark.parametrize("name,expected", KA.PENETRATION_CASES)
def test_box_box_penetration_accuracy_on_device(name, expected):
    KA.check_box_box_penetration_accuracy(_cs, name, expected)


def test_gjk_and_mpr_probe_answers_on_device():
    """test_gjk.py:160-235 / test_mpr.py:175-197 through the full pipeline: separated spheres and boxes report their true
    distance inside the contact gap, and the centred-tie box-box configuration keeps unit normals and exact depth."""
    import newton_amd as nt
    from pair_scenes import box, sphere

    def run(geoms):
        model = pair_model(geoms, device="cuda:0")
        pipe = nt.CollisionPipeline(model)
        ct = pipe.contacts()
        pipe.collide(model.state(), ct)
        n = int(ct.rigid_contact_count.cpu().numpy()[0])
        g = lambda k: getattr(ct, "rigid_contact_" + k).cpu().numpy()[:n]  # noqa: E731
        return decode_world(model, model.body_q, g("shape0"), g("shape1"), g("point0"), g("point1"), g("normal"), g("margin0"),
                            g("margin1"))

    cs = run([box(1.0, [-2, 0, 0], gap=1.5), box(1.0, [2.5, 0, 0], gap=1.5)])  # GJK: boxes 2.5 apart (test_gjk.py:203-211)
    assert len(cs) >= 1 and abs(min(d for _, _, d in cs) - 2.5) < 1e-5 and np.allclose(cs[0][1], [1, 0, 0], atol=1e-5)
    for k in (0.5, 2.0):  # MPR centred-tie boundary (test_mpr.py:175-197)
        ang = np.float32(1.0e-6 * k)
        q = [float(np.sin(0.5 * ang)), 0.0, 0.0, float(np.cos(0.5 * ang))]
        cs = run([box(0.5, [0, 0, 0]), box(0.5, [0, 0.999, 0], q)])
        expected = 0.5 + 0.5 * (np.cos(ang) + np.sin(ang)) - 0.999
        assert len(cs) >= 1
        for _, n, _ in cs:
            assert abs(np.linalg.norm(n) - 1.0) < 1e-6 and np.allclose(n, [0, 1, 0], atol=1e-5)
        assert abs(-min(d for _, _, d in cs) - expected) < 2e-5
