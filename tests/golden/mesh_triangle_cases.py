"""Case table shared by tests/golden/make_mesh_triangle_reference_vectors.py (which runs the REFERENCE's mesh-vs-convex leg: the
triangle midphase, GJK / MPR + manifold per triangle, the global contact reducer) and the tests (which hold
oracle/oracle_mesh_triangle.py and the HIP kernel nt_mesh_triangle_pairs against the record).

A case is a small scene in Newton's flat layout: shape types, world transforms, shape_data (scale xyz, margin), gaps, the local
AABBs and voxel resolutions of the shapes, vertex / index arrays of the triangle meshes, and the (mesh, convex) pairs the reference
routes to `shape_pairs_mesh` (newton/_src/geometry/narrow_phase.py:633-638) in the type-sorted order it stores them."""
import numpy as np

from mesh_plane_cases import _quat
from reduce_cases import voxel_resolution

CASES = ["box_on_grid", "sphere_on_terrain", "capsule_margins_scaled", "cylinder_and_cone", "mirrored_mesh", "ellipsoid_in_bowl",
         "separated", "hulls_on_terrain"]
# heightfield scenes (GeoType.HFIELD = 2 as shape a of every pair; TRIANGLE_PRISM cells, utils/heightfield.py:280-462)
HF_CASES = ["hf_box_sphere", "hf_capsule_cylinder_tilted", "hf_hull_cone_border"]
HFIELD = 2
# GeoType values (newton/_src/geometry/types.py)
SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX, MESH, CONE, CONVEX_MESH = 3, 4, 5, 6, 7, 8, 9, 10


def grid_mesh(nx, ny, sx, sy, height=None):
    """(nx x ny cells) height-field style triangle mesh over [-sx, sx] x [-sy, sy], two triangles per cell, normals up (+z)."""
    xs, ys = np.linspace(-sx, sx, nx + 1), np.linspace(-sy, sy, ny + 1)
    pts = np.array([(x, y, 0.0 if height is None else height(x, y)) for y in ys for x in xs], np.float32)
    idx = []
    for j in range(ny):
        for i in range(nx):
            a, b, c, d = j * (nx + 1) + i, j * (nx + 1) + i + 1, (j + 1) * (nx + 1) + i, (j + 1) * (nx + 1) + i + 1
            idx += [(a, b, d), (a, d, c)]
    return pts, np.array(idx, np.int32)


def closed_box_mesh(h):
    """12 triangles of a box with half extents h, outward winding."""
    hx, hy, hz = h
    p = np.array([(-hx, -hy, -hz), (hx, -hy, -hz), (hx, hy, -hz), (-hx, hy, -hz), (-hx, -hy, hz), (hx, -hy, hz), (hx, hy, hz),
                  (-hx, hy, hz)], np.float32)
    t = np.array([(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (2, 3, 7), (2, 7, 6), (1, 2, 6), (1, 6, 5),
                  (3, 0, 4), (3, 4, 7)], np.int32)
    return p, t


def primitive_local_aabb(t, scale):
    """builder.py:11601-11652: the local AABB Model.shape_collision_aabb_* holds for a primitive (the heightfield midphase reads it)."""
    sx, sy, sz = (float(x) for x in scale)
    if t == SPHERE:
        e = (sx, sx, sx)
    elif t in (BOX, ELLIPSOID):
        e = (sx, sy, sz)
    elif t == CAPSULE:
        e = (sx, sx, sy + sx)
    elif t in (CYLINDER, CONE):
        e = (sx, sx, sy)
    else:
        return None
    return -np.array(e, np.float32), np.array(e, np.float32)


def _tables(shapes):
    """shapes: list of dict(type, points / tris or None, xform[7], scale[3], margin, gap) -> the flat arrays."""
    S = len(shapes)
    out = dict(shape_type=np.array([s["type"] for s in shapes], np.int32),
               shape_transform=np.stack([np.asarray(s["xform"], np.float32) for s in shapes]),
               shape_data=np.array([[*s["scale"], s["margin"]] for s in shapes], np.float32),
               shape_gap=np.array([s["gap"] for s in shapes], np.float32),
               aabb_lo=np.zeros((S, 3), np.float32), aabb_hi=np.zeros((S, 3), np.float32), res=np.ones((S, 3), np.int32),
               vertex_start=np.zeros(S, np.int32), vertex_count=np.zeros(S, np.int32),
               tri_start=np.zeros(S, np.int32), tri_count=np.zeros(S, np.int32),
               hull_start=np.zeros(S, np.int32), hull_count=np.zeros(S, np.int32))
    out["hf_index"] = np.full(S, -1, np.int32)
    hf_table, hf_elev = [], []
    verts, tris, hulls = [], [], []
    nv = nt = nh = 0
    for k, s in enumerate(shapes):
        pa = primitive_local_aabb(s["type"], s["scale"])
        if pa is not None:
            out["aabb_lo"][k], out["aabb_hi"][k] = pa
            out["res"][k] = voxel_resolution(*pa)
        if s.get("hfield") is not None:  # Heightfield source: normalized data [nrow][ncol], hx, hy, min_z, max_z (geometry/types.py:2286-2340)
            data, hx, hy, zlo, zhi = s["hfield"]
            data = np.asarray(data, np.float32)
            out["hf_index"][k] = len(hf_table)
            hf_table.append((sum(len(e) for e in hf_elev), data.shape[0], data.shape[1], hx, hy, zlo, zhi))
            hf_elev.append(data.reshape(-1))
            sc = np.asarray(s["scale"], np.float64)  # builder.py:11653-11660
            lo = np.array([-abs(hx * sc[0]), -abs(hy * sc[1]), min(zlo * sc[2], zhi * sc[2])], np.float32)
            hi = np.array([abs(hx * sc[0]), abs(hy * sc[1]), max(zlo * sc[2], zhi * sc[2])], np.float32)
            out["aabb_lo"][k], out["aabb_hi"][k] = lo, hi
            out["res"][k] = voxel_resolution(lo, hi)
        if s.get("hull") is not None:  # CONVEX_MESH partner: its vertex table (wp.Mesh.points of the hull), scaled local AABB
            h = np.asarray(s["hull"], np.float32)
            sc = np.asarray(s["scale"], np.float32)
            out["aabb_lo"][k], out["aabb_hi"][k] = (h * sc).min(axis=0), (h * sc).max(axis=0)
            out["hull_start"][k], out["hull_count"][k] = nh, len(h)
            hulls.append(h)
            nh += len(h)
        if s.get("points") is None:
            continue
        p, t = np.asarray(s["points"], np.float32), np.asarray(s["tris"], np.int32)
        sc = np.asarray(s["scale"], np.float32)
        lo, hi = (p * sc).min(axis=0), (p * sc).max(axis=0)  # scaled local AABB (builder.py: mesh shapes)
        out["aabb_lo"][k], out["aabb_hi"][k] = lo, hi
        out["res"][k] = voxel_resolution(lo, hi)
        out["vertex_start"][k], out["vertex_count"][k] = nv, len(p)
        out["tri_start"][k], out["tri_count"][k] = nt, len(t)
        verts.append(p)
        tris.append(t)
        nv += len(p)
        nt += len(t)
    out["vertices"] = np.concatenate(verts).astype(np.float32) if verts else np.zeros((0, 3), np.float32)
    out["indices"] = np.concatenate(tris).astype(np.int32) if tris else np.zeros((0, 3), np.int32)  # mesh-local vertex ids
    out["hull_points"] = np.concatenate(hulls).astype(np.float32) if hulls else np.zeros((0, 3), np.float32)
    out["hf_table"] = np.array(hf_table, np.float32).reshape(-1, 7)  # data offset, nrow, ncol, hx, hy, min_z, max_z
    out["hf_elev"] = np.concatenate(hf_elev).astype(np.float32) if hf_elev else np.zeros(0, np.float32)
    return out


def scene(name):
    """-> dict of flat arrays + `pairs` [(shape a, shape b)] type-sorted like the reference's routing stores them (MESH = 8: the
    mesh is shape b except against a CONE = 9); the midphase emits (mesh, convex, triangle) whatever the stored order."""
    ident = [0, 0, 0, 1]

    def mesh(points, tris, xform, scale=(1, 1, 1), margin=0.0, gap=0.002):
        return dict(type=MESH, points=points, tris=tris, xform=xform, scale=list(scale), margin=margin, gap=gap)

    def prim(t, xform, scale, margin=0.0, gap=0.002, hull=None):
        return dict(type=t, xform=xform, scale=list(scale), margin=margin, gap=gap, hull=hull)

    def hfield(nrow, ncol, hx, hy, height, xform, margin=0.0, gap=0.002):
        xs, ys = np.linspace(-hx, hx, ncol), np.linspace(-hy, hy, nrow)
        raw = np.array([[height(x, y) for x in xs] for y in ys], np.float32)
        lo, hi = float(raw.min()), float(raw.max())
        return dict(type=HFIELD, xform=xform, scale=[1.0, 1.0, 1.0], margin=margin, gap=gap,
                    hfield=(((raw - lo) / (hi - lo)).astype(np.float32), hx, hy, lo, hi))

    def bump(x, y):
        return 0.03 * np.sin(6 * x) * np.cos(5 * y)

    if name == "box_on_grid":  # a box resting 1 mm inside a flat 8 x 8 grid, slightly rotated: face manifolds on many triangles
        p, t = grid_mesh(8, 8, 0.4, 0.4)
        shapes = [mesh(p, t, [0, 0, 0, *ident]),
                  prim(BOX, [0.03, -0.02, 0.099, *_quat((0.2, 0.1, 1), 0.4)], (0.12, 0.1, 0.1))]
        pairs = [(1, 0)]
    elif name == "sphere_on_terrain":  # bumpy terrain, tilted mesh frame; a sphere touching a slope
        p, t = grid_mesh(10, 10, 0.5, 0.5, height=lambda x, y: 0.05 * np.sin(7 * x) * np.cos(5 * y))
        qm = _quat((1, 0.5, 0), 0.2)
        shapes = [prim(SPHERE, [0.05, 0.03, 0.115, *ident], (0.1, 0.1, 0.1), gap=0.004),
                  mesh(p, t, [0, 0, 0, *qm], gap=0.003)]
        pairs = [(0, 1)]
    elif name == "capsule_margins_scaled":  # non-uniform mesh scale, shape margins on both sides, a lying capsule
        p, t = grid_mesh(6, 6, 0.5, 0.5, height=lambda x, y: 0.02 * x * y)
        shapes = [mesh(p, t, [0.1, 0, 0, *_quat((0, 0, 1), 0.3)], scale=(0.8, 1.3, 1.5), margin=0.002, gap=0.003),
                  prim(CAPSULE, [0.05, 0.1, 0.052, *_quat((0, 1, 0.1), 1.5)], (0.05, 0.15, 0.0), margin=0.001, gap=0.002)]
        pairs = [(1, 0)]
    elif name == "cylinder_and_cone":  # two pairs on one mesh: a standing cylinder and a tilted cone
        p, t = grid_mesh(8, 8, 0.5, 0.5)
        shapes = [mesh(p, t, [0, 0, 0, *ident]),
                  prim(CYLINDER, [-0.2, -0.1, 0.0995, *_quat((1, 0, 0), 0.01)], (0.08, 0.1, 0.0)),
                  prim(CONE, [0.2, 0.15, 0.098, *_quat((0, 1, 0), 0.05)], (0.1, 0.1, 0.0), gap=0.004)]
        pairs = [(1, 0), (0, 2)]
    elif name == "mirrored_mesh":  # negative scale component (mirror parity swaps the winding) + a closed mesh under a box
        p, t = closed_box_mesh((0.2, 0.15, 0.05))
        # mirrored in x: the reference swaps idx1 / idx2 when building the triangle, NOT in the midphase's front-face test
        shapes = [mesh(p, t, [0, 0, 0, *_quat((0, 0, 1), 0.2)], scale=(-1.0, 1.0, 1.0)),
                  prim(BOX, [0.05, 0.02, 0.0995, *_quat((0, 0, 1), -0.3)], (0.05, 0.06, 0.05))]
        pairs = [(1, 0)]
    elif name == "ellipsoid_in_bowl":  # concave bowl: contacts on a ring of triangles with very different normals
        p, t = grid_mesh(10, 10, 0.3, 0.3, height=lambda x, y: 1.2 * (x * x + y * y))
        shapes = [mesh(p, t, [0, 0, 0, *ident], gap=0.004),
                  prim(ELLIPSOID, [0.0, 0.0, 0.049, *_quat((1, 0, 0), 0.1)], (0.16, 0.12, 0.05), gap=0.004)]
        pairs = [(1, 0)]
    elif name == "hf_box_sphere":  # a box with a face manifold over several cells and a sphere in a dimple
        shapes = [hfield(13, 17, 0.8, 0.6, bump, [0, 0, 0, *ident]),
                  prim(BOX, [0.21, -0.13, bump(0.21, -0.13) + 0.049, *_quat((0.1, 0.2, 1), 0.6)], (0.09, 0.07, 0.05)),
                  prim(SPHERE, [-0.3, 0.2, bump(-0.3, 0.2) + 0.0595, *ident], (0.06, 0.06, 0.06), gap=0.004)]
        pairs = [(0, 1), (0, 2)]
    elif name == "hf_capsule_cylinder_tilted":  # tilted, offset heightfield frame; shape margins; lying capsule, standing cylinder
        qh = _quat((1, 0.4, 0.1), 0.25)
        shapes = [prim(CAPSULE, [0.12, 0.05, 0.068, *_quat((0, 1, 0.05), 1.45)], (0.04, 0.1, 0.0), margin=0.001, gap=0.003),
                  hfield(11, 11, 0.5, 0.5, lambda x, y: 0.02 * x * y + 0.01 * np.cos(9 * x), [0.05, -0.02, 0.01, *qh], margin=0.002, gap=0.003),
                  prim(CYLINDER, [-0.15, -0.1, 0.083, *_quat((1, 0.4, 0.1), 0.27)], (0.05, 0.06, 0.0))]
        pairs = [(1, 0), (1, 2)]
    elif name == "hf_hull_cone_border":  # a hull hanging over the border of the field (the cell range is clamped) and a tilted cone
        wedge = np.array([(-0.1, -0.06, 0.0), (0.1, -0.06, 0.0), (0.1, 0.06, 0.0), (-0.1, 0.06, 0.0), (-0.1, -0.06, 0.07), (-0.1, 0.06, 0.07)],
                         np.float32)
        shapes = [hfield(9, 9, 0.4, 0.4, lambda x, y: 0.04 * np.sin(5 * x + 1.0) * np.sin(4 * y), [0, 0, 0, *ident], gap=0.003),
                  prim(CONVEX_MESH, [0.36, 0.3, 0.04 * np.sin(5 * 0.36 + 1.0) * np.sin(1.2) + 0.003, *_quat((0, 0, 1), 0.4)], (1.0, 1.0, 1.0), hull=wedge),
                  prim(CONE, [-0.1, 0.1, 0.04 * np.sin(0.5) * np.sin(0.4) + 0.0985, *_quat((0, 1, 0), 0.06)], (0.08, 0.1, 0.0), gap=0.004)]
        pairs = [(0, 1), (0, 2)]
    elif name == "hulls_on_terrain":  # two convex hulls (a 20-vertex polytope, non-uniformly scaled; a wedge) on a bumpy terrain:
        # face, edge and vertex contacts; the hull's Minkowski seed is the centre of its scaled bounds, not its origin
        rng = np.random.default_rng(77)
        d = rng.normal(size=(20, 3))
        poly = (d / np.linalg.norm(d, axis=1, keepdims=True) * 0.08 + np.array([0.01, -0.005, 0.02])).astype(np.float32)
        wedge = np.array([(-0.1, -0.06, 0.0), (0.1, -0.06, 0.0), (0.1, 0.06, 0.0), (-0.1, 0.06, 0.0), (-0.1, -0.06, 0.07), (-0.1, 0.06, 0.07)],
                         np.float32)
        p, t = grid_mesh(12, 12, 0.5, 0.5, height=lambda x, y: 0.015 * np.sin(9 * x) * np.cos(7 * y))
        shapes = [mesh(p, t, [0, 0, 0, *_quat((0, 1, 0), 0.05)], gap=0.003),
                  prim(CONVEX_MESH, [-0.2, 0.1, 0.062, *_quat((1, 0.3, 0.2), 0.5)], (1.0, 0.8, 1.2), hull=poly, gap=0.003),
                  prim(CONVEX_MESH, [0.2, -0.1, 0.004, *_quat((0, 0, 1), 0.7)], (1.0, 1.0, 1.0), hull=wedge, margin=0.001, gap=0.002)]
        pairs = [(0, 1), (0, 2)]
    else:  # a V-shaped valley of large triangles: their bounds overlap the box's query AABB, the surfaces stay 5 cm away -- every
        # triangle pair buffers ONE contact beyond margin + gap (the writer's gap test drops them after the reduction)
        p, t = grid_mesh(2, 2, 0.4, 0.4, height=lambda x, y: 0.3 * abs(x))
        shapes = [mesh(p, t, [0, 0, 0, *ident]), prim(BOX, [0, 0, 0.12, *ident], (0.05, 0.05, 0.05))]
        pairs = [(1, 0)]
    out = _tables(shapes)
    out["pairs"] = np.array(pairs, np.int32).reshape(-1, 2)
    return out
