"""Host SDF construction + sampler (newton_amd/sdf.py) pinned by the reference's own accuracy tables, restated as numbers
from newton/tests/test_sdf_texture.py: sphere-mesh ground truth :1379-1421 (mean < 2e-4, median < 1.5e-4, p95 < 7e-4,
max < 2e-3 in the |d| < 0.05 band at resolution 64, float32), analytic sphere :1347-1376 (mean < 5e-4, p95 < 1e-3),
integer-voxel reads == interpolation :432-465 (rtol 2e-5, atol 2e-6), extrapolation :726-749 (> 0.5 far outside a unit box),
uint16 / uint8 vs float32 :894-984 (mean error < 0.05), gradients :1424-1483 (mean angle to the exact normal < 2 deg on a
sphere), construction fields :411-429, target_voxel_size precedence :1501-1586."""
import numpy as np
import pytest

from newton_amd import sdf as S
from newton_amd.enums import GeoType
from newton_amd.mesh import Mesh


def _box_mesh(h=(0.5, 0.5, 0.5)):
    return Mesh.create_box(*h)


def _sphere_points(radius, n, seed=0):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = radius + rng.uniform(-0.08, 0.08, size=(n, 1))
    return (d * r).astype(np.float32)


def test_construction_fields_and_bounds():
    m = _box_mesh()
    t, sparse = S.create_texture_sdf_from_mesh(m.vertices, m.indices, max_resolution=32, return_sparse_data=True)
    assert np.all(t.inv_dx > 0) and t.subgrid_size == 8
    assert np.all(t.box_lower <= m.vertices.min(axis=0)) and np.all(t.box_upper >= m.vertices.max(axis=0))
    w, h, d = sparse["coarse_dims"]
    assert t.slots.shape == (w, h, d) and t.coarse.shape == (d + 1, h + 1, w + 1)
    assert sparse["num_subgrids"] > 0 and t.subgrid.dtype == np.uint16
    assert t.subgrid.shape[0] % 9 == 0  # packed (subgrid_size + 1)^3 blocks
    occ = t.slots[(t.slots != S.SLOT_EMPTY) & (t.slots != S.SLOT_LINEAR)]
    assert len(occ) == sparse["num_subgrids"] == len(np.unique(occ))
    assert np.all((occ >> 30) == 0)  # three 10-bit block coordinates


def test_sphere_mesh_vs_ground_truth_distance():
    m = Mesh.create_sphere(0.5, 10, 14)  # brute-force ground truth: keep triangles x voxels small (CPU suite budget)
    t = S.create_texture_sdf_from_mesh(m.vertices, m.indices, max_resolution=40, quantization_mode=S.QuantizationMode.FLOAT32)
    q = _sphere_points(0.5, 1500)
    truth = S.mesh_sdf(m.vertices, m.indices, q)
    got = t.sample(q)
    valid = np.abs(truth) < 0.05
    assert valid.sum() > 300
    diff = np.abs(got[valid] - truth[valid])
    # voxel 0.028 m on a coarsely faceted sphere: trilinear error peaks at the facet creases (measured mean 2.5e-4, median 7e-8,
    # p95 1.4e-3, max 2.8e-3 = 0.1 voxel); the reference's 64^3 / 24x32 case scales to the same fractions of a voxel
    assert diff.mean() < 4e-4 and np.median(diff) < 1.5e-4 and np.percentile(diff, 95) < 2e-3 and diff.max() < 4e-3


def test_analytic_sphere_distance():
    t = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.5, 0.0, 0.0), max_resolution=64,
                                            quantization_mode=S.QuantizationMode.FLOAT32)
    q = _sphere_points(0.5, 2000, seed=123)
    diff = np.abs(t.sample(q) - (np.linalg.norm(q, axis=1) - 0.5))
    assert diff.mean() < 5e-4 and np.percentile(diff, 95) < 1e-3


def test_integer_voxel_reads_match_interpolation():
    m = _box_mesh()
    t = S.create_texture_sdf_from_mesh(m.vertices, m.indices, max_resolution=64)
    dims = np.rint((t.box_upper - t.box_lower) / t.voxel_size).astype(np.int64) + 1
    ijk = np.random.default_rng(2026).integers(np.zeros(3, dtype=np.int64), dims, size=(256, 3))
    direct = t.sample_at_voxel(ijk)
    interp = t.sample(t.box_lower + ijk.astype(np.float32) * t.voxel_size)
    np.testing.assert_allclose(direct, interp, rtol=2.0e-5, atol=2.0e-6)


def test_extrapolation_outside_the_box():
    m = _box_mesh()
    t = S.create_texture_sdf_from_mesh(m.vertices, m.indices, max_resolution=32)
    pts = np.array([[2.0, 0, 0], [3.0, 0, 0], [0, 2.0, 0], [0, 0, 2.0]], dtype=np.float32)
    vals = t.sample(pts)
    assert np.all(vals > 0.5)
    assert abs(vals[1] - vals[0] - 1.0) < 1e-3  # |p - clamp(p)| extension: one metre further is one metre more
    _, g = t.sample_grad(pts)
    assert np.allclose(g, np.eye(3)[[0, 0, 1, 2]], atol=1e-6)


@pytest.mark.parametrize("mode", [S.QuantizationMode.UINT16, S.QuantizationMode.UINT8])
def test_quantized_storage_tracks_float32(mode):
    m = _box_mesh()
    kw = dict(max_resolution=32)
    f = S.create_texture_sdf_from_mesh(m.vertices, m.indices, quantization_mode=S.QuantizationMode.FLOAT32, **kw)
    q = S.create_texture_sdf_from_mesh(m.vertices, m.indices, quantization_mode=mode, **kw)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-0.6, 0.6, size=(500, 3)).astype(np.float32)
    a, b = f.sample(pts), q.sample(pts)
    assert np.abs(a - b).mean() < (0.05 if mode == S.QuantizationMode.UINT8 else 1e-4)
    assert np.array_equal(f.slots, q.slots)


def test_gradient_points_along_the_sphere_normal():
    t = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.5, 0.0, 0.0), max_resolution=64)
    q = _sphere_points(0.5, 2000, seed=5)
    d, g = t.sample_grad(q)
    n = q / np.linalg.norm(q, axis=1, keepdims=True)
    cosang = np.sum(g * n, axis=1) / np.maximum(np.linalg.norm(g, axis=1), 1e-12)
    ang = np.degrees(np.arccos(np.clip(cosang, -1, 1)))
    assert ang.mean() < 2.0 and np.median(ang) < 1.0
    assert np.abs(d - t.sample(q)).max() < 1e-5  # the value of the gradient path agrees with the value-only path


def test_target_voxel_size_takes_precedence_and_validates():
    a = S.create_texture_sdf_from_primitive(GeoType.BOX, (0.5, 0.25, 0.25), target_voxel_size=0.02, max_resolution=8)
    assert abs(float(a.voxel_size[0]) - 0.02) / 0.02 < 0.3  # ~ (1.0 + 2 * margin) / (multiple of 8 >= 55)
    b = S.create_texture_sdf_from_primitive(GeoType.BOX, (0.5, 0.25, 0.25), max_resolution=8)
    assert float(b.voxel_size[0]) > 4 * float(a.voxel_size[0])
    with pytest.raises(ValueError):
        S.create_texture_sdf_from_primitive(GeoType.BOX, (0.5, 0.25, 0.25), target_voxel_size=0.0)
    with pytest.raises(ValueError):
        S.create_texture_sdf_from_primitive(GeoType.BOX, (0.5, np.nan, 0.25))


def test_primitive_sdfs_known_points():
    f = lambda t, s, p: float(S.primitive_sdf(t, s, [p])[0])  # noqa: E731
    assert abs(f(GeoType.SPHERE, (0.5, 0, 0), (1.0, 0, 0)) - 0.5) < 1e-12
    assert abs(f(GeoType.BOX, (0.5, 0.5, 0.5), (1.0, 1.0, 0.0)) - np.sqrt(0.5)) < 1e-12 and f(GeoType.BOX, (0.5,) * 3, (0, 0, 0)) == -0.5
    assert abs(f(GeoType.CAPSULE, (0.2, 0.5, 0), (0, 0, 1.0)) - 0.3) < 1e-12 and abs(f(GeoType.CAPSULE, (0.2, 0.5, 0), (0.5, 0, 0.2)) - 0.3) < 1e-12
    assert abs(f(GeoType.CYLINDER, (0.2, 0.5, 0), (0.5, 0, 0)) - 0.3) < 1e-12 and abs(f(GeoType.CYLINDER, (0.2, 0.5, 0), (0, 0, 0.7)) - 0.2) < 1e-12
    assert abs(f(GeoType.CONE, (0.3, 0.4, 0), (0, 0, 0.9)) - 0.5) < 1e-12 and f(GeoType.CONE, (0.3, 0.4, 0), (0, 0, -0.2)) < 0
    assert abs(f(GeoType.ELLIPSOID, (1.0, 0.5, 0.25), (2.0, 0, 0)) - 1.0) < 1e-9


def test_barrel_cylinder_sdf_matches_the_executed_reference():
    """tests/golden/make_barrel_sdf_vectors.py ran the reference's sdf_cylinder(..., barrel_radius) (geometry/kernels.py:347-447) on
    seeded points around three barrels; the host SDF the texture builder samples must agree (float32 reference vs float64 here)."""
    import os

    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "barrel_sdf_reference_vectors.npz"))
    for name in ("wide", "tight", "flat"):
        scale, pts, want = ref[f"{name}/scale"], ref[f"{name}/points"], ref[f"{name}/distance"]
        got = S.primitive_sdf(GeoType.CYLINDER, scale, pts)
        assert (want < 0).sum() > 40 and (want > 0).sum() > 40
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, float(np.abs(want).max())) + 1e-6, (name, np.abs(got - want).max())
    lo, hi = S.primitive_extents(GeoType.CYLINDER, (0.05, 0.08, 0.13))
    assert abs(hi[0] - (0.05 + 0.13 - (0.13 ** 2 - 0.08 ** 2) ** 0.5)) < 1e-12 and hi[2] == 0.08 and np.array_equal(lo, -hi)
