"""Case table shared by tests/golden/make_reduce_reference_vectors.py (which runs the REFERENCE's global contact reducer) and the
tests (which hold oracle/oracle_reduce.py and the HIP reduction against the record).  A case is an UNREDUCED contact list the
way the mesh-SDF kernel hands it to `export_and_reduce_contact_centered_two_spatial_depths`
(newton/_src/geometry/sdf_contact.py:1905-1947): per contact the shape pair, world point, normal a -> b, distance, fingerprint
((edge << 2) | (mode << 1)), midpoint-centred point, inner / outer spatial depth, the point in the edge shape's frame with that
shape's local AABB and voxel resolution."""
import numpy as np

CASES = ["patch", "two_pairs_many_normals", "duplicates_and_ties", "outer_only", "single"]


def voxel_resolution(lo, hi, budget=100):
    """builder.py:11544-11570 (compute_voxel_resolution_from_aabb), float64 like the builder."""
    size = np.maximum(np.asarray(hi, np.float64) - np.asarray(lo, np.float64), 1e-6)
    v = max((size[0] * size[1] * size[2] / budget) ** (1.0 / 3.0), 1e-6)
    nx, ny, nz = (max(1, round(size[k] / v)) for k in range(3))
    while nx * ny * nz > budget:
        if nx >= ny and nx >= nz and nx > 1:
            nx -= 1
        elif ny >= nz and ny > 1:
            ny -= 1
        elif nz > 1:
            nz -= 1
        else:
            break
    return nx, ny, nz


def _unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def _pair_block(rng, pair, n, centres, spread, normal_dirs, normal_noise, depth_lo, depth_hi, inner, outer, edge0=0):
    """n contacts of one shape pair: half from mode 0 (edges of shape a), half from mode 1."""
    mid = rng.uniform(-0.2, 0.2, size=3)
    aabbs = [(np.array([-0.3, -0.2, -0.25], np.float32), np.array([0.3, 0.25, 0.2], np.float32)),
             (np.array([-0.15, -0.4, -0.1], np.float32), np.array([0.2, 0.35, 0.3], np.float32))]
    rows = []
    for k in range(n):
        mode = k & 1
        c = centres[rng.integers(len(centres))]
        pos = (mid + c + rng.normal(0.0, spread, size=3)).astype(np.float32)
        nd = normal_dirs[rng.integers(len(normal_dirs))]
        nrm = _unit(np.asarray(nd) + rng.normal(0.0, normal_noise, size=3))
        depth = np.float32(rng.uniform(depth_lo, depth_hi))
        lo, hi = aabbs[mode]
        local = rng.uniform(lo - 0.02, hi + 0.02).astype(np.float32)  # some points outside the box: the voxel index clamps
        rows.append(dict(pair=pair, pos=pos, normal=nrm, depth=depth, fp=((edge0 + k) << 2) | (mode << 1),
                         centered=(pos - mid.astype(np.float32)).astype(np.float32), inner=np.float32(inner),
                         outer=np.float32(outer), local=local, aabb_lo=lo, aabb_hi=hi, res=voxel_resolution(lo, hi)))
    return rows


def contacts(name):
    rng = np.random.default_rng(100 + CASES.index(name))
    z = [(0.0, 0.0, 0.0)]
    if name == "patch":  # a flat contact patch: one normal bin, the spatial extremes matter
        rows = _pair_block(rng, (3, 7), 400, z, 0.05, [(0.05, 0.1, 1.0)], 0.02, -0.004, 0.012, 0.002, 0.01)
    elif name == "two_pairs_many_normals":  # gear-like: normals everywhere, several clusters, two shape pairs
        dirs = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0.2, 0), (0.3, -1, 0.1), (0, 0.1, -1), (1, 1, 1), (-1, 1, -1)]
        cs = [(0.1, 0, 0), (-0.1, 0.05, 0), (0, 0, 0.12), (0, -0.1, -0.05)]
        rows = _pair_block(rng, (2, 5), 500, cs, 0.03, dirs, 0.25, -0.01, 0.03, 0.003, 0.02)
        rows += _pair_block(rng, (5, 9), 300, cs, 0.02, dirs, 0.4, -0.002, 0.01, 0.001, 0.008)
    elif name == "duplicates_and_ties":
        rows = _pair_block(rng, (1, 2), 120, z, 0.04, [(0, 0, 1), (0, 1, 0.2)], 0.05, -0.003, 0.006, 0.002, 0.005)
        base = len(rows)
        for k in range(24):  # roundoff twins: the same geometry under another fingerprint (shared mesh corner seen from two edges)
            src = dict(rows[rng.integers(base)])
            src["fp"] = ((base + k) << 2) | (src["fp"] & 2)
            if k % 3 == 1:  # a few ulps apart
                src["pos"] = np.nextafter(src["pos"], np.float32(1.0)).astype(np.float32)
            if k % 3 == 2:  # same truncated score, not equivalent: the fingerprint decides
                src["pos"] = (src["pos"] + np.float32(3e-6)).astype(np.float32)
                src["centered"] = (src["centered"] + np.float32(3e-6)).astype(np.float32)
            rows.append(src)
    elif name == "outer_only":  # nothing inside the inner depth: only the directional slots fill, no depth / voxel slots
        rows = _pair_block(rng, (4, 6), 150, z, 0.05, [(0, 0, 1), (1, 0, 0)], 0.1, 0.004, 0.02, 0.002, 0.015)
    else:
        rows = _pair_block(rng, (0, 1), 1, z, 0.0, [(0, 0, 1)], 0.0, -0.001, -0.001, 0.002, 0.01)
    order = rng.permutation(len(rows))  # arrival order of the threads
    return [rows[i] for i in order]


def pack(rows):
    """-> dict of arrays (the layout the checker and the device entry take)."""
    f = lambda k: np.stack([np.asarray(r[k], np.float32) for r in rows])  # noqa: E731
    return dict(pair=np.array([r["pair"] for r in rows], np.int32), pos=f("pos"), normal=f("normal"), depth=f("depth"),
                fp=np.array([r["fp"] for r in rows], np.int32), centered=f("centered"), inner=f("inner"), outer=f("outer"),
                local=f("local"), aabb_lo=f("aabb_lo"), aabb_hi=f("aabb_hi"), res=np.array([r["res"] for r in rows], np.int32))
