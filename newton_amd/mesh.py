"""Triangle mesh asset (host only): the subset of newton.Mesh (newton/_src/geometry/types.py) that convex-hull collision
shapes need -- vertices, triangles, solid mass properties at unit density, an optional hull reduction."""
from __future__ import annotations

import numpy as np


def solid_mesh_mass_properties(vertices, indices):
    """(volume, com[3], inertia about the COM [3,3]) of the closed triangle mesh at unit density: signed tetrahedra against
    the origin, like compute_solid_mesh_inertia / compute_inertia_mesh (newton/_src/geometry/inertia.py:307-470,473-600)."""
    v = np.asarray(vertices, dtype=np.float64)
    tri = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
    a, b, c = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    vol6 = np.einsum("ij,ij->i", a, np.cross(b, c))
    V = vol6.sum() / 6.0
    F = ((a + b + c) / 4.0 * (vol6 / 6.0)[:, None]).sum(axis=0)  # first moment
    # second moment  S = integral x x^T dV  over each tetrahedron (0, a, b, c)
    S = np.zeros((3, 3))
    for p, q in ((a, a), (b, b), (c, c)):
        S += np.einsum("i,ij,ik->jk", vol6 / 60.0, p, q)
    for p, q in ((a, b), (a, c), (b, c)):
        S += np.einsum("i,ij,ik->jk", vol6 / 120.0, p, q) + np.einsum("i,ij,ik->jk", vol6 / 120.0, q, p)
    if V < 0.0:  # inward winding
        V, F, S = -V, -F, -S
    com = F / V if V > 0.0 else F
    I_origin = np.trace(S) * np.eye(3) - S
    I_com = I_origin - V * (np.dot(com, com) * np.eye(3) - np.outer(com, com))
    return float(V), com, I_com


class Mesh:
    """newton.Mesh(vertices, indices, compute_inertia=True, is_solid=True)."""

    def __init__(self, vertices, indices, compute_inertia: bool = True, is_solid: bool = True, maxhullvert: int = 64):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        self.is_solid = is_solid
        self.maxhullvert = maxhullvert
        self.has_inertia = False
        self.sdf = None        # TextureSDF attached by build_sdf()
        self.sdf_scale = None  # the scale baked into it (None: unbaked)
        self.mass, self.com, self.inertia = 1.0, np.zeros(3), np.eye(3)
        if compute_inertia:
            if not is_solid:
                raise NotImplementedError("hollow meshes are not supported")
            self.mass, self.com, self.inertia = solid_mesh_mass_properties(self.vertices, self.indices)
            self.has_inertia = True

    def build_sdf(self, *, device=None, narrow_band_range=None, target_voxel_size=None, max_resolution=None, margin=None,
                  shape_margin: float = 0.0, scale=None, texture_format: str = "uint16", **unsupported):
        """Build and attach the sparse texture SDF of this mesh (Mesh.build_sdf, geometry/types.py:820-1010): what
        ModelBuilder.finalize() registers in Model._texture_sdf_data for every shape that uses the mesh, which routes its pairs
        with other SDF shapes through the mesh-SDF narrow phase.  Host construction (newton_amd.sdf); `scale` bakes a shape
        scale into the grid.  The reference's edge simplification options and on-disk cache are not offered."""
        from . import sdf as S  # noqa: PLC0415

        unsupported = {k: v for k, v in unsupported.items() if k not in ("sign_method", "paired_samples") or v not in ("auto", True)}
        if unsupported or shape_margin != 0.0:
            raise NotImplementedError(f"Mesh.build_sdf options not supported: {sorted(unsupported) or ['shape_margin']}")
        fmt = {"float32": S.QuantizationMode.FLOAT32, "uint16": S.QuantizationMode.UINT16, "uint8": S.QuantizationMode.UINT8}
        if texture_format not in fmt:
            raise ValueError(f"Unknown texture_format {texture_format!r}. Expected one of {list(fmt)}.")
        if max_resolution is not None and max_resolution % 8 != 0:
            raise ValueError(f"max_resolution must be divisible by 8 (got {max_resolution}).")
        if max_resolution is None and target_voxel_size is None:
            max_resolution = 64
        sc = (1.0, 1.0, 1.0) if scale is None else tuple(float(x) for x in scale)
        v = np.asarray(self.vertices, dtype=np.float64) * np.asarray(sc, dtype=np.float64)
        self.sdf = S.create_texture_sdf_from_mesh(v, np.asarray(self.indices).reshape(-1, 3), margin=0.05 if margin is None else margin,
                                                  narrow_band_range=narrow_band_range or (-0.1, 0.1), max_resolution=max_resolution,
                                                  target_voxel_size=target_voxel_size, quantization_mode=fmt[texture_format],
                                                  scale_baked=scale is not None)
        self.sdf_scale = sc if scale is not None else None
        self.sdf._construction_padding = 0.05 if margin is None else margin
        return self.sdf

    @staticmethod
    def create_box(hx: float, hy: float | None = None, hz: float | None = None, *, duplicate_vertices: bool = False,
                   compute_normals: bool = False, compute_uvs: bool = False, compute_inertia: bool = True,
                   reference_layout: bool = False) -> Mesh:
        """newton.Mesh.create_box signature (geometry/types.py:560-620).  duplicate_vertices=True builds the reference's per-face
        vertex table (utils/mesh.py:2044-2065: faces -x, +x, -y, +y, -z, +z, four corners each, cycling (-,-), (-,+), (+,+), (+,-)
        over the face's two other axes) -- the vertex ORDER matters once the mesh collides vertex by vertex (add_shape_mesh: the
        vertex index is the contact fingerprint); pinned by tests/golden/mesh_box_tables.json.  The default keeps this package's
        8-corner hull (convex hulls de-duplicate anyway; normals / uvs are rendering data: accepted and ignored);
        reference_layout=True with duplicate_vertices=False gives the reference's shared-vertex table instead (utils/mesh.py:2066-2100:
        the z- ring counter-clockwise from (-, -), then the z+ ring; same pin) -- same solid, the reference's contact fingerprints."""
        hy = hx if hy is None else hy
        hz = hx if hz is None else hz
        if reference_layout and not duplicate_vertices:
            ring = ((-1, -1), (1, -1), (1, 1), (-1, 1))
            verts = [(sx * hx, sy * hy, sz * hz) for sz in (-1, 1) for sx, sy in ring]
            quads = ((4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (0, 4, 7, 3), (1, 2, 6, 5))  # z+, y-, y+, x-, x+
            tris = [(0, 2, 1), (0, 3, 2)] + [t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))]  # (z- first)
            return Mesh(np.asarray(verts, dtype=np.float32), np.asarray(tris, dtype=np.int32), compute_inertia=compute_inertia)
        if duplicate_vertices:
            h = (float(hx), float(hy), float(hz))
            cyc = ((-1, -1), (-1, 1), (1, 1), (1, -1))
            verts, tris = [], []
            for axis in range(3):
                u, v = [k for k in range(3) if k != axis]
                for sign in (-1, 1):
                    base = len(verts)
                    for cu, cv in cyc:
                        p = [0.0, 0.0, 0.0]
                        p[axis], p[u], p[v] = sign * h[axis], cu * h[u], cv * h[v]
                        verts.append(p)
                    forward = (sign < 0) != (axis == 1)  # outward winding: x- / y+ / z- list the cycle as it is
                    tris += [(base, base + 1, base + 2), (base, base + 2, base + 3)] if forward else \
                        [(base, base + 2, base + 1), (base, base + 3, base + 2)]
            return Mesh(np.asarray(verts, dtype=np.float32), np.asarray(tris, dtype=np.int32), compute_inertia=compute_inertia)
        s = np.array([hx, hy, hz], dtype=np.float64)
        v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64) * s
        return Mesh.convex_hull_of(v)

    @staticmethod
    def create_sphere(radius: float = 1.0, num_latitudes: int = 32, num_longitudes: int = 32, compute_inertia: bool = True,
                      reference_layout: bool = False) -> Mesh:
        """UV sphere, outward winding.  Default: poles + latitude rings about +z, every vertex once (2 + (lat - 1) * lon).
        reference_layout=True: the reference's own table (utils/mesh.py:1026-1075 behind newton.Mesh.create_sphere) -- the
        (lat + 1) x (lon + 1) grid about +y with the seam column and the poles repeated, two triangles per grid cell (degenerate at
        the poles) -- vertex for vertex, so that a mesh sphere colliding vertex by vertex carries the reference's contact
        fingerprints (pinned by tests/golden/mesh_box_tables.json, recorded by executing the reference function)."""
        if reference_layout:
            verts, tris = [], []
            for i in range(num_latitudes + 1):
                th = i * np.pi / num_latitudes
                for j in range(num_longitudes + 1):
                    ph = j * 2 * np.pi / num_longitudes
                    verts.append((np.cos(ph) * np.sin(th) * radius, np.cos(th) * radius, np.sin(ph) * np.sin(th) * radius))
            for i in range(num_latitudes):
                for j in range(num_longitudes):
                    a = i * (num_longitudes + 1) + j
                    b = a + num_longitudes + 1
                    tris += [(a, a + 1, b), (b, a + 1, b + 1)]
            return Mesh(np.asarray(verts, dtype=np.float32), np.asarray(tris, dtype=np.int32), compute_inertia=compute_inertia)
        verts = [(0.0, 0.0, radius)]
        for i in range(1, num_latitudes):
            th = np.pi * i / num_latitudes
            for j in range(num_longitudes):
                ph = 2.0 * np.pi * j / num_longitudes
                verts.append((radius * np.sin(th) * np.cos(ph), radius * np.sin(th) * np.sin(ph), radius * np.cos(th)))
        verts.append((0.0, 0.0, -radius))
        south = len(verts) - 1

        def ring(i, j):
            return 1 + (i - 1) * num_longitudes + (j % num_longitudes)

        tris = []
        for j in range(num_longitudes):
            tris.append((0, ring(1, j), ring(1, j + 1)))
            tris.append((south, ring(num_latitudes - 1, j + 1), ring(num_latitudes - 1, j)))
        for i in range(1, num_latitudes - 1):
            for j in range(num_longitudes):
                a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
                tris.append((a, c, d))
                tris.append((a, d, b))
        return Mesh(np.asarray(verts), np.asarray(tris, dtype=np.int32), compute_inertia=compute_inertia)

    @staticmethod
    def convex_hull_of(points) -> Mesh:
        """Mesh of the convex hull of a point cloud, outward winding (scipy.spatial.ConvexHull)."""
        from scipy.spatial import ConvexHull  # noqa: PLC0415

        pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
        hull = ConvexHull(pts)
        used = np.unique(hull.simplices)
        remap = -np.ones(len(pts), dtype=np.int64)
        remap[used] = np.arange(len(used))
        verts = pts[used]
        center = verts.mean(axis=0)
        tris = []
        for simplex, eq in zip(hull.simplices, hull.equations):
            i, j, k = remap[simplex]
            n = np.cross(verts[j] - verts[i], verts[k] - verts[i])
            if np.dot(n, eq[:3]) < 0.0:  # make the winding agree with the outward facet normal
                j, k = k, j
            tris.append((i, j, k))
        del center
        return Mesh(verts, np.asarray(tris, dtype=np.int32))


def deduplicate_vertices(mesh: Mesh) -> np.ndarray:
    """Collision vertex set: each exact position once, first-occurrence order
    (_deduplicate_convex_collision_mesh, newton/_src/sim/builder.py:104-137)."""
    v = mesh.vertices
    _, first = np.unique(v, axis=0, return_index=True)
    return v[np.sort(first)]


def canonical_vertex_ids(vertices) -> np.ndarray:
    """Per-vertex ids that fold geometrically coincident vertices (1e-7 m buckets) together
    (Mesh._canonical_vertex_ids, newton/_src/geometry/types.py:1172-1180)."""
    q = np.ascontiguousarray(np.round(np.asarray(vertices, dtype=np.float32) * 1e7).astype(np.int64))
    void = q.view(np.dtype((np.void, q.dtype.itemsize * q.shape[1])))
    _, canonical = np.unique(void, return_inverse=True)
    return canonical.ravel()


def mesh_edges(vertices, indices) -> np.ndarray:
    """Unique edges [N,2] (original vertex indices, first occurrence order) with geometric de-duplication
    (Mesh.edges, newton/_src/geometry/types.py:1183-1212)."""
    tris = np.asarray(indices, dtype=np.int32).reshape(-1, 3)
    if tris.size == 0:
        return np.empty((0, 2), dtype=np.int32)
    c = canonical_vertex_ids(vertices)[tris]
    n = len(tris)
    canon = np.empty((n * 3, 2), dtype=np.int64)
    orig = np.empty((n * 3, 2), dtype=np.int32)
    for k, (a, b) in enumerate(((0, 1), (1, 2), (0, 2))):
        canon[k::3, 0], canon[k::3, 1] = np.minimum(c[:, a], c[:, b]), np.maximum(c[:, a], c[:, b])
        orig[k::3, 0], orig[k::3, 1] = tris[:, a], tris[:, b]
    canon = np.ascontiguousarray(canon)
    _, first = np.unique(canon.view(np.dtype((np.void, canon.dtype.itemsize * 2))), return_index=True)
    first.sort()
    return orig[first]


def mesh_edge_tables(vertices, indices, scale=(1.0, 1.0, 1.0)):
    """(edge_centers [N,4], edge_halves [N,4]) of one shape: scaled local centre + radius, scaled half vector + corner
    ownership code 4 + owns_v0 + 2 * owns_v1 (the first edge that touches a canonical vertex owns it), as the reference's
    finalize() packs them (newton/_src/sim/builder.py:12088-12116)."""
    edges = mesh_edges(vertices, indices)
    v = np.asarray(vertices, dtype=np.float32) * np.asarray(scale, dtype=np.float32)
    if len(edges) == 0:
        return np.zeros((0, 4), dtype=np.float32), np.zeros((0, 4), dtype=np.float32)
    v0, v1 = v[edges[:, 0]], v[edges[:, 1]]
    halves = np.ascontiguousarray((v1 - v0) * 0.5, dtype=np.float32)
    centers = np.ascontiguousarray((v0 + v1) * 0.5, dtype=np.float32)
    radii = np.linalg.norm(halves, axis=1, keepdims=True)
    canon = canonical_vertex_ids(vertices)[edges].reshape(-1)
    ends = np.arange(2 * len(edges), dtype=np.int64)
    first = np.full(int(canon.max()) + 1, 2 * len(edges), dtype=np.int64)
    np.minimum.at(first, canon, ends)
    owns = (first[canon] == ends).reshape(-1, 2)
    code = (4.0 + owns[:, 0].astype(np.float32) + 2.0 * owns[:, 1]).reshape(-1, 1)
    return (np.ascontiguousarray(np.concatenate([centers, radii], axis=1), dtype=np.float32),
            np.ascontiguousarray(np.concatenate([halves, code], axis=1), dtype=np.float32))


TRIANGLE_BLOCK = 64  # NT_MESH_TRIANGLE_BLOCK (include/newton_hip_mesh.h)


def triangle_block_bounds(vertices, indices, block: int = TRIANGLE_BLOCK) -> np.ndarray:
    """Bounds of every `block` consecutive triangles of a mesh: [ceil(T / block)][6] float32 (lower xyz, upper xyz) of the UNSCALED
    vertices -- min / max only, so the values are exact.  The triangle leg (csrc/nt_mesh_triangle.hip) skips the blocks that miss a
    pair's query box before it tests triangles: the host-built stand-in for the upper levels of the reference's mesh BVH
    (collision_core.py:1144-1180 queries wp.Mesh's), over the mesh's own index order."""
    v = np.asarray(vertices, dtype=np.float32).reshape(-1, 3)
    tri = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
    n = -(-len(tri) // block)
    out = np.zeros((n, 6), dtype=np.float32)
    for b in range(n):
        p = v[tri[b * block:(b + 1) * block].reshape(-1)]
        out[b, :3], out[b, 3:] = p.min(axis=0), p.max(axis=0)
    return out


class Heightfield:
    """newton.Heightfield (geometry/types.py:2240-2340): a 2-D elevation grid for terrain.  The data is normalised to [0, 1];
    min_z / max_z give the world-space range (derived from the data when both are omitted); the grid spans [-hx, hx] x [-hy, hy],
    rows along Y, columns along X.  Always static.  Collides with convex shapes cell by cell (two TRIANGLE_PRISM triangles per cell,
    utils/heightfield.py:280-462) through the triangle leg of CollisionPipeline."""

    def __init__(self, data, nrow: int, ncol: int, hx: float = 1.0, hy: float = 1.0, min_z=None, max_z=None):
        if nrow < 2 or ncol < 2:
            raise ValueError(f"Heightfield requires nrow >= 2 and ncol >= 2, got nrow={nrow}, ncol={ncol}")
        if (min_z is None) != (max_z is None):
            raise ValueError("min_z and max_z must both be provided or both omitted")
        raw = np.array(data, dtype=np.float32).reshape(nrow, ncol)
        d_min, d_max = float(raw.min()), float(raw.max())
        self._data = (raw - d_min) / (d_max - d_min) if d_max > d_min else np.zeros_like(raw)
        self.nrow, self.ncol, self.hx, self.hy = int(nrow), int(ncol), float(hx), float(hy)
        self.min_z = d_min if min_z is None else float(min_z)
        self.max_z = d_max if max_z is None else float(max_z)
        self.is_solid, self.has_inertia, self.mass = True, False, 0.0

    @property
    def data(self):
        """The normalised [0, 1] elevation data, [nrow][ncol]."""
        return self._data

    @property
    def vertices(self):
        """The eight corners of the field's bounding box: what the tiles need of a heightfield (a pre-computed local AABB)."""
        return np.array([(sx * self.hx, sy * self.hy, z) for sx in (-1.0, 1.0) for sy in (-1.0, 1.0) for z in (self.min_z, self.max_z)],
                        dtype=np.float32)
