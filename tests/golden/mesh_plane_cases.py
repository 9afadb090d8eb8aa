"""Case table shared by tests/golden/make_mesh_plane_reference_vectors.py (which runs the REFERENCE's mesh-plane narrow phase and
its global contact reducer) and the tests (which hold oracle/oracle_mesh_plane.py and the HIP kernel against the record).

A case is a small scene in Newton's flat layout: shape transforms (world), shape_data (scale xyz, margin), gaps, local AABBs and
voxel resolutions of the shapes, the vertex arrays of the meshes, and the (mesh, plane) pairs the reference routes to its mesh-plane
kernel (newton/_src/geometry/narrow_phase.py:618-631)."""
import numpy as np

from reduce_cases import voxel_resolution

CASES = ["cube_flat", "sphere_on_tilted_plane", "two_meshes_margins", "roundoff_twins", "separated"]


def _quat(axis, angle):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    return np.concatenate([a * np.sin(0.5 * angle), [np.cos(0.5 * angle)]]).astype(np.float32)


def grid_box(h, n):
    """Vertices of a box with half extents h, every face an (n+1) x (n+1) grid (shared edges / corners listed once)."""
    h = np.asarray(h, np.float64)
    t = np.linspace(-1.0, 1.0, n + 1)
    pts = set()
    for ax in range(3):
        o = [k for k in range(3) if k != ax]
        for s in (-1.0, 1.0):
            for a in t:
                for b in t:
                    p = [0.0, 0.0, 0.0]
                    p[ax], p[o[0]], p[o[1]] = s, a, b
                    pts.add(tuple(np.round(np.asarray(p) * h, 9)))
    return np.array(sorted(pts), np.float32)


def uv_sphere(r, rings, segs):
    pts = [(0.0, 0.0, r), (0.0, 0.0, -r)]
    for i in range(1, rings):
        th = np.pi * i / rings
        for j in range(segs):
            ph = 2.0 * np.pi * j / segs
            pts.append((r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), r * np.cos(th)))
    return np.array(pts, np.float32)


def _shape_tables(shapes):
    """shapes: list of dict(points or None, xform[7], scale[3], margin, gap) -> the flat arrays."""
    S = len(shapes)
    out = dict(shape_transform=np.stack([np.asarray(s["xform"], np.float32) for s in shapes]),
               shape_data=np.array([[*s["scale"], s["margin"]] for s in shapes], np.float32),
               shape_gap=np.array([s["gap"] for s in shapes], np.float32),
               aabb_lo=np.zeros((S, 3), np.float32), aabb_hi=np.zeros((S, 3), np.float32), res=np.ones((S, 3), np.int32),
               vertex_start=np.zeros(S, np.int32), vertex_count=np.zeros(S, np.int32))
    verts = []
    n = 0
    for k, s in enumerate(shapes):
        if s["points"] is None:
            continue
        p = np.asarray(s["points"], np.float32)
        sc = np.asarray(s["scale"], np.float32)
        lo, hi = (p * sc).min(axis=0), (p * sc).max(axis=0)  # scaled local AABB (builder.py: mesh shapes)
        out["aabb_lo"][k], out["aabb_hi"][k] = lo, hi
        out["res"][k] = voxel_resolution(lo, hi)
        out["vertex_start"][k], out["vertex_count"][k] = n, len(p)
        verts.append(p)
        n += len(p)
    out["vertices"] = np.concatenate(verts).astype(np.float32) if verts else np.zeros((0, 3), np.float32)
    return out


def scene(name):
    """-> dict of flat arrays + `pairs` [(mesh shape, plane shape)] (ascending pair order, the order the pipeline keeps)."""
    rng = np.random.default_rng(500 + CASES.index(name))
    plane0 = dict(points=None, xform=[0, 0, 0, 0, 0, 0, 1], scale=[0, 0, 0], margin=0.0, gap=0.002)
    if name == "cube_flat":  # a 9 x 9 grid per face, resting 1 mm inside the ground, a hair of tilt: one normal bin, many candidates
        q = _quat((1, 0.3, 0), 0.004)
        shapes = [plane0, dict(points=grid_box((0.1, 0.1, 0.1), 8), xform=[0.3, -0.2, 0.099, *q], scale=[1, 1, 1], margin=0.0, gap=0.002)]
        pairs = [(1, 0)]
    elif name == "sphere_on_tilted_plane":  # tilted, offset plane; non-uniform mesh scale; only the cap is inside the gap
        qp = _quat((0.2, 1, 0.1), 0.35)
        plane = dict(points=None, xform=[0.1, 0.2, -0.05, *qp], scale=[0, 0, 0], margin=0.0, gap=0.004)
        n = np.array([0, 0, 1.0])
        # plane normal in world = rotate z by qp
        x, y, z, w = qp.astype(np.float64)
        nw = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])
        c = np.array([0.1, 0.2, -0.05]) + nw * 0.118 + np.array([0.03, -0.02, 0.0]) - nw * np.dot(nw, [0.03, -0.02, 0.0])
        shapes = [plane, dict(points=uv_sphere(0.1, 24, 32), xform=[*c, *_quat((0, 0, 1), 0.7)], scale=[1.0, 0.8, 1.2], margin=0.0,
                              gap=0.006)]
        pairs = [(1, 0)]
    elif name == "two_meshes_margins":  # two meshes on one plane, shape margins; the plane has the LARGER shape id for one pair
        q1, q2 = _quat((0, 1, 0), 0.3), _quat((1, 1, 0), -0.2)
        shapes = [dict(points=grid_box((0.08, 0.12, 0.05), 5), xform=[0.0, 0.0, 0.0475, *q1], scale=[1, 1, 1], margin=0.002, gap=0.003),
                  dict(points=None, xform=[0, 0, 0, 0, 0, 0, 1], scale=[0, 0, 0], margin=0.001, gap=0.001),
                  dict(points=uv_sphere(0.07, 10, 12), xform=[0.5, 0.1, 0.069, *q2], scale=[1, 1, 1], margin=0.0, gap=0.004)]
        pairs = [(0, 1), (2, 1)]
    elif name == "roundoff_twins":  # duplicated vertices (mesh seams) + vertices a few ulps apart: the export drops the twins
        base = grid_box((0.1, 0.1, 0.02), 3)
        dup = base[rng.integers(0, len(base), size=12)]
        near = np.nextafter(base[rng.integers(0, len(base), size=8)], np.float32(1.0))
        shapes = [plane0, dict(points=np.concatenate([base, dup, near]), xform=[0, 0, 0.0195, 0, 0, 0, 1], scale=[1, 1, 1], margin=0.0,
                               gap=0.002)]
        pairs = [(1, 0)]
    else:  # no vertex within margin + gap of the plane
        shapes = [plane0, dict(points=grid_box((0.1, 0.1, 0.1), 2), xform=[0, 0, 0.2, 0, 0, 0, 1], scale=[1, 1, 1], margin=0.0, gap=0.002)]
        pairs = [(1, 0)]
    out = _shape_tables(shapes)
    out["pairs"] = np.array(pairs, np.int32).reshape(-1, 2)
    return out
