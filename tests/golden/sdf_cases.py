"""Cases shared by make_sdf_reference_vectors.py (which EXECUTES the reference on them) and tests/test_sdf_reference_vectors.py
(which holds the checker and the kernels against the record): texture SDFs built by newton_amd.sdf (host construction, pinned
by the reference's accuracy tables) in every storage mode, query points inside / outside the box, edges for the Brent search and
shape-pair scenes for the mesh-vs-SDF kernel."""
import numpy as np

from newton_amd import sdf as S
from newton_amd.enums import GeoType
from newton_amd.mesh import Mesh, mesh_edge_tables


def hull_mesh(seed=0, n=14, radius=0.3):
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(n, 3))
    pts *= radius / np.linalg.norm(pts, axis=1).max()
    return Mesh.convex_hull_of(pts)


def sdfs():
    """name -> TextureSDF"""
    out = {}
    for name, mode in (("box_u16", S.QuantizationMode.UINT16), ("box_u8", S.QuantizationMode.UINT8), ("box_f32", S.QuantizationMode.FLOAT32)):
        out[name] = S.create_texture_sdf_from_primitive(GeoType.BOX, (0.5, 0.4, 0.3), max_resolution=32, quantization_mode=mode)
    m = hull_mesh()
    out["hull_u16"] = S.create_texture_sdf_from_mesh(m.vertices.astype(np.float64), m.indices.reshape(-1, 3), margin=0.05,
                                                     narrow_band_range=(-0.1, 0.1), max_resolution=24)
    out["sphere_u16_baked"] = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.5, 0.5, 0.5), max_resolution=32, scale_baked=True)
    return out


def query_points(t, n=160, seed=1):
    """Points inside the SDF box, near the surface band, on the box faces and outside (extrapolation)."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(t.box_lower, np.float64), np.asarray(t.box_upper, np.float64)
    inside = rng.uniform(lo, hi, size=(n // 2, 3))
    outside = rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), size=(n // 2 - 6, 3))
    corners = np.array([lo, hi, [lo[0], hi[1], lo[2]], 0.5 * (lo + hi), [hi[0], 0.5 * (lo[1] + hi[1]), lo[2]], lo + 1e-7])
    return np.concatenate([inside, outside, corners]).astype(np.float32)


def voxels(t, n=60, seed=2):
    rng = np.random.default_rng(seed)
    cx, cy, cz = t.slots.shape
    ss = int(t.subgrid_size)
    return np.stack([rng.integers(0, cx * ss + 1, n), rng.integers(0, cy * ss + 1, n), rng.integers(0, cz * ss + 1, n)], axis=1).astype(np.int32)


def edges(t, n=80, seed=3):
    """Random edges around the SDF box: (v0 [n,3], v1 [n,3], precision target [n]); a few degenerate / very short ones."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(t.box_lower, np.float64), np.asarray(t.box_upper, np.float64)
    c = rng.uniform(lo, hi, size=(n, 3))
    d = rng.normal(size=(n, 3)) * rng.uniform(0.01, 0.5, size=(n, 1)) * (hi - lo).max()
    v0, v1 = c - 0.5 * d, c + 0.5 * d
    v1[0] = v0[0]                       # zero-length edge
    v1[1] = v0[1] + 1e-5                # shorter than any precision target
    prec = rng.choice([1e-4, float(t.voxel_radius), 0.02], size=n)
    return v0.astype(np.float32), v1.astype(np.float32), prec.astype(np.float32)


def pair_scenes():
    """name -> dict(pairs, X, data (scale, margin), gap, sdf_index, sdfs, er, ec, eh) on Newton's flat arrays."""
    out = {}
    box = Mesh.create_box(0.5, 0.5, 0.5)
    ec_b, eh_b = mesh_edge_tables(box.vertices, box.indices.reshape(-1, 3))
    sdf_b = S.create_texture_sdf_from_primitive(GeoType.BOX, (0.5, 0.5, 0.5), max_resolution=32)
    for name, dz, margin, gap in (("cube_on_cube", 0.98, 0.0, 0.02), ("cube_on_cube_margin", 1.0, 0.01, 0.03), ("cube_apart", 1.2, 0.0, 0.02)):
        yaw = 0.3
        q = [0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)]
        out[name] = dict(pairs=np.array([[0, 1]], np.int32), X=np.array([[0, 0, 0, 0, 0, 0, 1], [0.05, -0.03, dz, *q]], np.float32),
                         data=np.array([[1, 1, 1, margin]] * 2, np.float32), gap=np.array([gap, gap], np.float32),
                         sdf_index=np.array([0, 0], np.int32), sdfs=[sdf_b], er=np.array([[0, len(ec_b)]] * 2, np.int32), ec=ec_b, eh=eh_b)
    # two hulls + a scaled box (unbaked SDF of the unit cube used at scale (0.6, 0.4, 0.2): the anisotropic-scale path)
    h0, h1 = hull_mesh(0), hull_mesh(1, 18, 0.25)
    tabs = [mesh_edge_tables(m.vertices, m.indices.reshape(-1, 3)) for m in (h0, h1)]
    sc = (1.2, 0.8, 0.4)
    ec_s, eh_s = mesh_edge_tables(box.vertices, box.indices.reshape(-1, 3), scale=sc)
    sd = [S.create_texture_sdf_from_mesh(m.vertices.astype(np.float64), m.indices.reshape(-1, 3), margin=0.05, narrow_band_range=(-0.1, 0.1),
                                         max_resolution=24) for m in (h0, h1)] + [sdf_b]
    n0, n1 = len(tabs[0][0]), len(tabs[1][0])
    rng = np.random.default_rng(5)
    q = rng.normal(size=(3, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = np.array([[0, 0, 0.08, *q[0]], [0.3, 0.1, 0.1, *q[1]], [0.0, 0.0, -0.18, 0, 0, 0, 1]], np.float32)
    out["hulls_on_scaled_box"] = dict(pairs=np.array([[0, 1], [0, 2], [1, 2]], np.int32), X=X,
                                      data=np.array([[1, 1, 1, 0.002], [1, 1, 1, 0.002], [*sc, 0.0]], np.float32),
                                      gap=np.array([0.02, 0.02, 0.01], np.float32), sdf_index=np.array([0, 1, 2], np.int32), sdfs=sd,
                                      er=np.array([[0, n0], [n0, n1], [n0 + n1, len(ec_s)]], np.int32),
                                      ec=np.concatenate([tabs[0][0], tabs[1][0], ec_s]), eh=np.concatenate([tabs[0][1], tabs[1][1], eh_s]))
    return out
