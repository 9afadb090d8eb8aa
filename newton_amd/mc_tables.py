"""Marching-cubes case tables for the hydroelastic iso-pressure surface (newton/_src/geometry/sdf_mc.py:47-106).

The reference takes them from ``wp.MarchingCubes`` (CASE_TO_TRI_RANGE, TRI_LOCAL_INDICES) -- part of the un-vendored warp-lang
runtime, so the literal tables are not in /root/reference.  They are generated here instead, from the published construction:
corner numbering of ``_mc_corner_offset`` (sdf_hydroelastic.py:206-213; corners 0-3 the z = 0 ring (0,0) (1,0) (1,1) (0,1), 4-7
the z = 1 ring), the 12 edges of ``edge_to_verts`` (sdf_mc.py:60-75), and per case: the crossed edges of every cube face are
joined pairwise (a face with four crossings cuts off each inside corner separately), the segments chain into closed loops,
each loop becomes a triangle fan.  Triangles are wound so that their normal points towards the INSIDE corners (bit set <=>
value < 0): in the hydroelastic kernel the value is p_other - p_self, so the normal runs from the "other" shape (A) to the
shape whose grid is traversed (B) -- the a -> b direction ContactData needs (sdf_hydroelastic.py:1803-1928).
Any valid triangulation describes the same surface; the per-voxel triangle split may differ from Warp's table (the patch area,
force and normal sums do not), which is stated in DESIGN.md as part of "parity unpinned at bit level".
"""
from __future__ import annotations

import numpy as np

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=np.int64)
EDGE_TO_VERTS = np.array([(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)],
                         dtype=np.int64)
# the six faces as corner cycles
FACES = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
MAX_MC_FACES_PER_VOXEL = 5


def _edge_id(a, b):
    for k, (u, v) in enumerate(EDGE_TO_VERTS):
        if (u == a and v == b) or (u == b and v == a):
            return k
    raise KeyError((a, b))


def _case_triangles(case: int):
    inside = [(case >> i) & 1 == 1 for i in range(8)]
    nbr = {}  # crossed edge -> the (up to two) crossed edges it is joined with

    def link(e0, e1):
        nbr.setdefault(e0, []).append(e1)
        nbr.setdefault(e1, []).append(e0)

    for cyc in FACES:
        crossed = []  # (edge id, index of the corner the edge starts at in the cycle)
        for k in range(4):
            a, b = cyc[k], cyc[(k + 1) % 4]
            if inside[a] != inside[b]:
                crossed.append((_edge_id(a, b), k))
        if len(crossed) == 2:
            link(crossed[0][0], crossed[1][0])
        elif len(crossed) == 4:
            # two diagonal inside corners: cut each one off on its own (join the two edges that meet at an inside corner)
            for k in range(4):
                if inside[cyc[k]]:
                    link(_edge_id(cyc[k - 1], cyc[k]), _edge_id(cyc[k], cyc[(k + 1) % 4]))
    tris, seen = [], set()
    for start in sorted(nbr):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [e for e in nbr[cur] if e != prev]
            if len(nbr[cur]) == 2 and nbr[cur][0] == nbr[cur][1]:  # (cannot happen on a cube, kept as a guard)
                nxt = [nbr[cur][0]]
            step = None
            for e in nxt:
                if e not in seen:
                    step = e
                    break
            if step is None:
                break
            loop.append(step)
            seen.add(step)
            prev, cur = cur, step
        if len(loop) < 3:
            continue
        # orientation: normal towards the inside corners
        mid = lambda e: 0.5 * (CORNERS[EDGE_TO_VERTS[e][0]] + CORNERS[EDGE_TO_VERTS[e][1]])  # noqa: E731
        pts = np.array([mid(e) for e in loop], dtype=np.float64)
        centroid = pts.mean(axis=0)
        normal = np.zeros(3)
        for k in range(1, len(loop) - 1):
            normal += np.cross(pts[k] - pts[0], pts[k + 1] - pts[0])
        # the inside corners touched by the loop's edges
        touched = {v for e in loop for v in EDGE_TO_VERTS[e] if inside[v]}
        toward = np.mean([CORNERS[v] for v in touched], axis=0) - centroid
        if np.dot(normal, toward) < 0.0:
            loop = loop[::-1]
        for k in range(1, len(loop) - 1):
            tris.append((loop[0], loop[k], loop[k + 1]))
    return tris


def build_tables():
    """(tri_range [257] int32, flat_edge_verts [n][2] uint8): triangle vertex k of case c lies on the cube edge joining corners
    flat_edge_verts[tri_range[c] + k] (three consecutive entries per triangle), like get_mc_tables (sdf_mc.py:47-106)."""
    ranges, flat = [0], []
    for case in range(256):
        tris = _case_triangles(case)
        assert len(tris) <= MAX_MC_FACES_PER_VOXEL, (case, len(tris))
        for tri in tris:
            for e in tri:
                flat.append(tuple(int(x) for x in EDGE_TO_VERTS[e]))
        ranges.append(len(flat))
    return np.asarray(ranges, dtype=np.int32), np.asarray(flat, dtype=np.uint8).reshape(-1, 2)


_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = build_tables()
    return _TABLES
