"""Known answers of the convex path (MPR -> GJK -> manifold), restated as numbers from the reference's own tests
(newton/tests/test_narrow_phase.py:707-800,1677-1960,2739-3150) and from plain geometry.  Each check takes `contacts(name)`
-> list of (center, normal, penetration) for the pair scene `name` of pair_scenes.CONVEX_CASES, so the SAME assertions pin the
CPU oracle (tests/test_oracle_convex.py) and the HIP kernels on the device (tests/test_gpu_convex_known_answers.py) --
the device path's parity does not rest on the oracle sharing its source."""
import numpy as np


def _sd_box(p, center, half):
    q = np.abs(np.asarray(p) - center) - half
    return np.linalg.norm(np.maximum(q, 0.0)) + min(q.max(), 0.0)


def check_box_box_face(contacts):
    cs = contacts("box_box_face")
    assert len(cs) == 4
    for c, n, d in cs:
        assert abs(np.linalg.norm(n) - 1.0) < 1e-5 and n[0] > 0.9
        assert abs(d + 0.2) < 1e-4
        assert abs(_sd_box(c - n * d / 2, [0, 0, 0], 1.0)) < 5e-5 and abs(_sd_box(c + n * d / 2, [1.8, 0, 0], 1.0)) < 5e-5


def check_box_box_edge(contacts):
    cs = contacts("box_box_edge")
    assert len(cs) > 0
    assert abs(np.linalg.norm(cs[0][1]) - 1.0) < 1e-5
    assert abs(min(d for _, _, d in cs) - (1.2 - 0.5 - np.sqrt(0.5))) < 1e-4


PENETRATION_CASES = [("box_box_overlap_0p01", -0.01), ("box_box_touching", 0.0),
                                           ("box_box_small_thickness", -(0.01 + 5e-5)), ("box_box_large_thickness", -0.02)]


def check_box_box_penetration_accuracy(contacts, name, expected):
    """test_narrow_phase.py:2739-2807,2869-2951: deepest penetration within 5e-5 for each `enlarge` branch."""
    cs = contacts(name)
    assert len(cs) > 0
    assert abs(min(d for _, _, d in cs) - expected) < 5e-5


def check_box_box_contact_point_on_surface(contacts):
    cs = contacts("box_box_overlap_0p05")
    ok = 0
    for c, n, d in cs:
        if d >= 0:
            continue
        assert abs(_sd_box(c - n * d / 2, [0, 0, 0], 0.5)) < 5e-5
        assert abs(_sd_box(c + n * d / 2, [0, 0, 0.95], 0.5)) < 5e-5
        ok += 1
    assert ok > 0


def check_ellipsoid_family(contacts):
    cs = contacts("ell_ell_separated")
    assert len(cs) == 0 or cs[0][2] > 0.0
    cs = contacts("ell_ell_penetrating")
    assert len(cs) == 1 and cs[0][2] < 0 and abs(np.linalg.norm(cs[0][1]) - 1) < 1e-5 and cs[0][1][0] > 0
    assert abs(cs[0][2] + 0.2) < 1e-3
    # type sorting puts the sphere first (SPHERE < ELLIPSOID): normal points sphere -> ellipsoid = -x
    cs = contacts("ell_sphere")
    assert len(cs) == 1 and cs[0][2] < 0 and cs[0][1][0] < -0.9 and abs(cs[0][2] + 0.1) < 1e-3
    cs = contacts("ell_box")
    assert len(cs) == 1 and cs[0][1][0] > 0.9 and abs(cs[0][2] + 0.1) < 1e-3
    cs = contacts("ell_capsule")
    assert len(cs) == 1 and abs(np.linalg.norm(cs[0][1]) - 1) < 1e-5
    cs = contacts("ell_ell_spherelike")
    assert len(cs) == 1 and abs(cs[0][2] + 0.2) < 1e-3 and cs[0][1][0] > 0.99


def check_axial_shapes(contacts):
    """Cylinder / cone / capsule manifolds: counts and depths from plain geometry."""
    cs = contacts("capsule_box")       # capsule lying on the box top: 2 end contacts, depth 0.05
    assert len(cs) >= 2
    assert abs(min(d for _, _, d in cs) + 0.05) < 2e-4
    cs = contacts("cylinder_box_flat")  # cap face on box top: depth 0.02
    assert len(cs) >= 3
    assert all(abs(d + 0.02) < 2e-4 for _, _, d in cs)
    cs = contacts("cylinder_box_rolling")  # cylinder on its side: line contact, depth 0.01
    assert len(cs) >= 2
    assert abs(min(d for _, _, d in cs) + 0.01) < 2e-4
    for c, n, d in cs:  # rolling projection keeps contacts in the plane through the axis
        assert abs(c[0]) < 1e-4
    cs = contacts("cone_box")  # base on the box top, depth 0.01
    assert len(cs) >= 3 and all(abs(d + 0.01) < 2e-4 for _, _, d in cs)
    cs = contacts("sphere_cone")
    assert len(cs) == 1
    cs = contacts("cylinder_cylinder")
    assert len(cs) >= 3 and all(abs(d + 0.02) < 2e-4 for _, _, d in cs)


CHECKS = [check_box_box_face, check_box_box_edge, check_box_box_contact_point_on_surface, check_ellipsoid_family,
          check_axial_shapes]
