"""TEST INFRASTRUCTURE ONLY -- float32 restatement of the reference's hydroelastic contact generation for SDF pairs
(newton/_src/geometry/sdf_hydroelastic.py), unreduced path (reduce_contacts=False: generate -> decode):
  get_effective_stiffness :216-222, linear_pressure :237-248, classify_hydroelastic_contact :140-145
  mc_iterate_voxel_vertices :1716-1798, mc_calc_face_texture :282-362, get_triangle_fraction sdf_mc.py:112-162
  generate_contacts_kernel :1982-2140 (pre_prune off), decode_contacts_kernel :1823-1928, pair normalisation :1329-1376
The reference finds the voxels that carry iso-surface faces with a block broad phase + octree refinement (:1026-1170); this
restatement visits every voxel of the finer SDF's grid inside the other SDF's box, which yields the same voxel set as long as
that search has no false negatives (its purpose).  Marching-cubes case tables: newton_amd/mc_tables.py (Warp's own table is not
in /root/reference; triangulations of a case may differ, the surface does not).  The octahedral normal encoding of the contact
buffer (a storage format) is skipped.  The reduce_contacts=True path (per-bin aggregates, local-first pruning,
HydroelasticContactReduction.reduce / export of contact_reduction_hydroelastic.py) is restated in reduce_pair_faces below.
PINNED: tests/golden/hydro_reference_vectors.npz holds the outputs of the reference's own kernels executed on the stand-in
(make_hydro_reference_vectors.py); hydro_pipeline reproduces them bit for bit, reduced and unreduced
(tests/test_hydro_reference_vectors.py).  Only tests/ may import this."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_sdf import OracleSDF, _q_rot, _x_inv, _x_mul, _x_point  # noqa: E402

f32 = np.float32
MC_EDGE_VAL_DIFF_EPS = f32(1.0e-10)
MC_DEGENERATE_N_SQ_EPS = f32(1.0e-20)
EPS_SMALL = f32(1e-20)
MAX_MC_FACES_PER_VOXEL = 5
CORNER = [((i & 3) ^ ((i & 3) >> 1)) & 1 for i in range(8)], [(i >> 1) & 1 for i in range(8)], [(i >> 2) & 1 for i in range(8)]


def effective_stiffness(ka, kb):
    d = f32(ka + kb)
    return f32(0.0) if d <= 0.0 else f32(f32(ka * kb) / d)


def classify(pair_separation, gap_sum):
    if pair_separation < 0.0:
        return -1
    return int(pair_separation > gap_sum)


def triangle_fraction(d, num_inside):
    if num_inside == 3:
        return f32(1.0)
    if num_inside == 0:
        return f32(0.0)
    d0, d1, d2 = f32(d[0]), f32(d[1]), f32(d[2])
    if num_inside == 1:
        if d[1] < 0.0:
            d0, d1, d2 = f32(d[1]), f32(d[2]), f32(d[0])
        elif d[2] < 0.0:
            d0, d1, d2 = f32(d[2]), f32(d[0]), f32(d[1])
    else:
        if d[1] >= 0.0:
            d0, d1, d2 = f32(d[1]), f32(d[2]), f32(d[0])
        elif d[2] >= 0.0:
            d0, d1, d2 = f32(d[2]), f32(d[0]), f32(d[1])
    denom = f32(f32(d0 - d1) * f32(d0 - d2))
    if abs(denom) < f32(1e-8):
        return f32(0.0) if num_inside == 1 else f32(1.0)
    fr = min(max(f32(f32(d0 * d0) / denom), f32(0.0)), f32(1.0))
    return f32(f32(1.0) - fr) if num_inside == 2 else fr


def hydro_collide(pairs, shape_transform, shape_data, shape_gap, shape_kh, sdfs, tables, margin_contact_area=1.0e-2,
                  edge_clamp_min=0.02):
    """-> list of (pair_idx, fingerprint, shape_a, shape_b, centre_world[3], normal_world[3], depth, stiffness, area, pressure)
    in (pair, voxel, face) order; fingerprint = voxel_linear * 5 + face, voxel_linear = (z * ny + y) * nx + x over B's fine cells."""
    tri_range, flat = tables
    X = np.asarray(shape_transform, dtype=f32)
    D = np.asarray(shape_data, dtype=f32)
    out = []
    cmin, cmax = f32(edge_clamp_min), f32(1.0 - edge_clamp_min)
    for pair_idx, (sa, sb) in enumerate(np.asarray(pairs).reshape(-1, 2)):
        ta, tb = sdfs[sa], sdfs[sb]
        if ta is None or tb is None:
            continue
        if tb.voxel_radius > ta.voxel_radius:  # keep the finer SDF as shape B
            sa, sb, ta, tb = sb, sa, tb, ta
        oa, ob = OracleSDF(ta), OracleSDF(tb)
        gap_sum = f32(f32(shape_gap[sa]) + f32(shape_gap[sb]))
        margin_a, margin_b = D[sa, 3], D[sb, 3]
        kh_a, kh_b = f32(shape_kh[sa]), f32(shape_kh[sb])
        X_b, X_a_inv = X[sb], _x_inv(X[sa])
        X_b2a = _x_mul(X_a_inv, X_b)
        vs = tb.voxel_size.astype(f32)
        nx, ny, nz = (int(c) * tb.subgrid_size for c in tb.slots.shape)
        # candidate voxel range: A's SDF box (its 8 corners, widened by the gap) seen from B's grid
        X_a2b = _x_inv(X_b2a)
        cs = np.array([[(ta.box_lower if (k >> a) & 1 == 0 else ta.box_upper)[a] for a in range(3)] for k in range(8)], dtype=f32)
        cb = np.array([_x_point(X_a2b, c) for c in cs])
        lo = np.floor((cb.min(axis=0) - gap_sum - ob.lo) / vs).astype(np.int64) - 1
        hi = np.ceil((cb.max(axis=0) + gap_sum - ob.lo) / vs).astype(np.int64) + 1
        lo, hi = np.maximum(lo, 0), np.minimum(hi, [nx, ny, nz])
        step = [_q_rot(X_b2a[3:], np.array([vs[0], 0, 0], dtype=f32)), _q_rot(X_b2a[3:], np.array([0, vs[1], 0], dtype=f32)),
                _q_rot(X_b2a[3:], np.array([0, 0, vs[2]], dtype=f32))]
        for z in range(lo[2], hi[2]):
            for y in range(lo[1], hi[1]):
                for x in range(lo[0], hi[0]):
                    base_b = (ob.lo + np.array([x, y, z], dtype=f32) * vs).astype(f32)
                    base_a = _x_point(X_b2a, base_b)
                    cube, any_gap, valid = 0, False, True
                    cv, cs_self, cs_other = np.zeros(8, f32), np.zeros(8, f32), np.zeros(8, f32)
                    for i in range(8):
                        ox, oy, oz = CORNER[0][i], CORNER[1][i], CORNER[2][i]
                        pa = (base_a + f32(ox) * step[0] + f32(oy) * step[1] + f32(oz) * step[2]).astype(f32)
                        v_self = tb.sample_at_voxel([[x + ox, y + oy, z + oz]])[0]
                        v_other = oa.sample(pa)
                        if np.isnan(v_self) or np.isnan(v_other):
                            valid = False
                            break
                        es, eo = f32(v_self - margin_b), f32(v_other - margin_a)
                        vd = f32(f32(-kh_a * eo) - f32(-kh_b * es))
                        cv[i], cs_self[i], cs_other[i] = vd, es, eo
                        if vd < 0.0:
                            cube |= 1 << i
                        if f32(es + eo) <= gap_sum:
                            any_gap = True
                    if not valid or not any_gap:
                        continue
                    t0, t1 = int(tri_range[cube]), int(tri_range[cube + 1])
                    for fi in range((t1 - t0) // 3):
                        verts, v_sdf, v_sep, n_in = np.zeros((3, 3), f32), np.zeros(3, f32), np.zeros(3, f32), 0
                        for vi in range(3):
                            a, b = int(flat[t0 + 3 * fi + vi][0]), int(flat[t0 + 3 * fi + vi][1])
                            vd = f32(cv[b] - cv[a])
                            t = f32(0.5) if abs(vd) < MC_EDGE_VAL_DIFF_EPS else min(max(f32(f32(f32(0.0) - cv[a]) / vd), cmin), cmax)
                            p0 = np.array([CORNER[0][a], CORNER[1][a], CORNER[2][a]], dtype=f32)
                            p1 = np.array([CORNER[0][b], CORNER[1][b], CORNER[2][b]], dtype=f32)
                            vol = (p0 + t * (p1 - p0) + np.array([x, y, z], dtype=f32)).astype(f32)
                            verts[vi] = ob.lo + vol * vs
                            s_self = f32(cs_self[a] + f32(t * f32(cs_self[b] - cs_self[a])))
                            s_other = f32(cs_other[a] + f32(t * f32(cs_other[b] - cs_other[a])))
                            v_sdf[vi], v_sep[vi] = s_self, f32(s_self + s_other)
                            if v_sep[vi] < 0.0:
                                n_in += 1
                        n = np.cross(verts[1] - verts[0], verts[2] - verts[0]).astype(f32)
                        n_sq = f32(np.dot(n, n))
                        if n_sq < MC_DEGENERATE_N_SQ_EPS:
                            garea, normal = f32(0.0), np.array([0, 0, 1], dtype=f32)
                        else:
                            inv = f32(1.0) / np.sqrt(n_sq)
                            normal, garea = (n * inv).astype(f32), f32(f32(n_sq * inv) * f32(0.5))
                        center = ((verts[0] + verts[1] + verts[2]) / f32(3.0)).astype(f32)
                        adj = f32(f32(f32(v_sdf[0] + v_sdf[1]) + v_sdf[2]) / f32(3.0))
                        sep = f32(f32(f32(v_sep[0] + v_sep[1]) + v_sep[2]) / f32(3.0))
                        farea = f32(garea * triangle_fraction(v_sep, n_in))
                        if garea <= 0.0 or classify(sep, gap_sum) > 0:
                            continue
                        pressure = max(f32(-kh_b * adj), f32(0.0)) if sep < 0.0 else f32(0.0)
                        area = farea if sep < 0.0 else garea
                        if sep < 0.0:
                            stiff = f32(f32(area * pressure) / max(f32(-sep), EPS_SMALL))
                        else:
                            stiff = f32(f32(margin_contact_area) * effective_stiffness(kh_a, kh_b))
                        out.append((pair_idx, ((z * ny + y) * nx + x) * MAX_MC_FACES_PER_VOXEL + fi, int(sa), int(sb),
                                    _x_point(X_b, center), _q_rot(X_b[3:], normal), sep, stiff, area, pressure))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# The staged pipeline of HydroelasticSDF.launch (reduce_contacts=False), pair by pair, in the reference's ordered-scatter order.
# PINNED by tests/golden/hydro_reference_vectors.npz (tests/golden/make_hydro_reference_vectors.py executes the reference's
# broadphase_collision_pairs_count, count_iso_voxels_block (4 levels), scatter_iso_subblock, generate_contacts_kernel and
# decode_contacts_kernel on the stand-in of tests/golden/refshim; the marching-cubes case tables are newton_amd/mc_tables.py on
# both sides, Warp's own wp.MarchingCubes tables are not part of /root/reference).
# ---------------------------------------------------------------------------------------------------------------------------------
def _dot3(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross3(a, b):
    return np.array([f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
                     f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))], dtype=f32)


def sat_box_intersection(Ta, ea, Tb, eb):
    """collision_core.py:1281-1374."""
    ex = [np.array(v, f32) for v in ((1, 0, 0), (0, 1, 0), (0, 0, 1))]
    axa, axb = [_q_rot(Ta[3:], e) for e in ex], [_q_rot(Tb[3:], e) for e in ex]

    def project(T, ax, e, n):
        c = _dot3(T[:3], n)
        ext = f32(0.0)
        for k in range(3):
            ext = f32(ext + f32(f32(e[k]) * abs(_dot3(ax[k], n))))
        return f32(c - ext), f32(c + ext)

    def separated(axis):
        ln = np.sqrt(_dot3(axis, axis))
        if ln < f32(1e-8):
            return False
        n = (axis / ln).astype(f32)
        la, ha = project(Ta, axa, ea, n)
        lb, hb = project(Tb, axb, eb, n)
        return bool(ha < lb or hb < la)

    for a in axa:
        if separated(a):
            return False
    for b in axb:
        if separated(b):
            return False
    for a in axa:
        for b in axb:
            if separated(_cross3(a, b)):
                return False
    return True


def encode_oct(n):
    """contact_reduction_global.py:631-658."""
    l1 = f32(f32(abs(n[0]) + abs(n[1])) + abs(n[2]))
    if l1 < f32(1.0e-20):
        return f32(0.0), f32(0.0)
    inv = f32(f32(1.0) / l1)
    ox, oy, oz = f32(n[0] * inv), f32(n[1] * inv), f32(n[2] * inv)
    if oz < 0.0:
        sx, sy = (f32(-1.0) if ox < 0.0 else f32(1.0)), (f32(-1.0) if oy < 0.0 else f32(1.0))
        ox, oy = f32(f32(f32(1.0) - abs(oy)) * sx), f32(f32(f32(1.0) - abs(ox)) * sy)
    return ox, oy


def decode_oct(ex, ey):
    """contact_reduction_global.py:661-683."""
    nz = f32(f32(f32(1.0) - abs(ex)) - abs(ey))
    nx, ny = f32(ex), f32(ey)
    if nz < 0.0:
        sx, sy = (f32(-1.0) if nx < 0.0 else f32(1.0)), (f32(-1.0) if ny < 0.0 else f32(1.0))
        nx, ny = f32(f32(f32(1.0) - abs(ny)) * sx), f32(f32(f32(1.0) - abs(nx)) * sy)
    v = np.array([nx, ny, nz], f32)
    ln = np.sqrt(_dot3(v, v))
    return (v / ln).astype(f32) if ln > 0.0 else v


def _node_survives(oa, ob, ta, tb, X_b2a, x, y, z, size, margin_a, margin_b, kh_a, kh_b, gap_sum):
    """count_iso_voxels_block's test of one cube of `size` voxels of B (sdf_hydroelastic.py:1499-1560)."""
    r = f32(f32(size) * f32(tb.voxel_radius))
    h = f32(f32(0.5) * f32(size))
    centre = np.array([f32(f32(x) + h), f32(f32(y) + h), f32(f32(z) + h)], f32)
    local_b = (ob.lo + centre * tb.voxel_size.astype(f32)).astype(f32)
    point_a = _x_point(X_b2a, local_b)
    vb = tb.sample_at_voxel([[x + size // 2, y + size // 2, z + size // 2]])[0] if size % 2 == 0 else ob.sample(local_b)
    va = oa.sample(point_a)
    if np.isnan(vb) or np.isnan(va):
        return False
    eva, evb = f32(va - margin_a), f32(vb - margin_b)
    if f32(eva + evb) > f32(f32(f32(2.0) * r) + gap_sum):
        return False
    pa_lo, pa_hi = f32(-kh_a * f32(eva + r)), f32(-kh_a * f32(eva - r))
    pb_lo, pb_hi = f32(-kh_b * f32(evb + r)), f32(-kh_b * f32(evb - r))
    return not (pa_hi < pb_lo or pb_hi < pa_lo)


def hydro_pipeline(pairs, shape_transform, shape_data, shape_gap, shape_kh, sdfs, tables, margin_contact_area=1.0e-2,
                   edge_clamp_min=0.02, reduce=None):
    """-> (rows, voxels): rows = list of (pair_idx, fingerprint, shape_a, shape_b, centre_world[3], normal_world[3], separation,
    stiffness) in (pair, voxel traversal, face) order with fingerprint = pair-local voxel rank * 5 + face; voxels = per pair the
    surviving (x, y, z) list in traversal order.  `sdfs[s]`: TextureSDF of shape s.
    reduce = dict(aabb_lo [S,3], aabb_hi [S,3], res [S,3], pre_prune=True, normal_matching=True): the reduce_contacts=True path
    (reduce_pair_faces): rows then are (pair_idx, fingerprint, shape_a, shape_b, centre, normal, separation, stiffness,
    friction scale) in export order."""
    X = np.asarray(shape_transform, dtype=f32)
    D = np.asarray(shape_data, dtype=f32)
    rows, vox_all = [], []
    for pair_idx, (sa, sb) in enumerate(np.asarray(pairs).reshape(-1, 2)):
        sa, sb = int(sa), int(sb)
        ta, tb = sdfs[sa], sdfs[sb]
        vox = []
        vox_all.append(vox)
        if ta is None or tb is None:
            continue
        lo_a, hi_a, lo_b, hi_b = (np.asarray(v, f32) for v in (ta.box_lower, ta.box_upper, tb.box_lower, tb.box_upper))
        ident = np.array([0, 0, 0, 1], f32)
        Ca = _x_mul(X[sa], np.concatenate([(f32(0.5) * (lo_a + hi_a)).astype(f32), ident]))
        Cb = _x_mul(X[sb], np.concatenate([(f32(0.5) * (lo_b + hi_b)).astype(f32), ident]))
        collide = sat_box_intersection(Ca, (f32(0.5) * (hi_a - lo_a)).astype(f32), Cb, (f32(0.5) * (hi_b - lo_b)).astype(f32))
        if tb.voxel_radius > ta.voxel_radius:  # keep the finer SDF as shape B
            sa, sb, ta, tb = sb, sa, tb, ta
        if not collide:
            continue
        oa, ob = OracleSDF(ta), OracleSDF(tb)
        gap_sum = f32(f32(shape_gap[sa]) + f32(shape_gap[sb]))
        margin_a, margin_b = D[sa, 3], D[sb, 3]
        kh_a, kh_b = f32(shape_kh[sa]), f32(shape_kh[sb])
        X_b = X[sb]
        X_b2a = _x_mul(_x_inv(X[sa]), X_b)
        nbx, nby, nbz = (int(c) for c in tb.slots.shape)
        sgs = int(tb.subgrid_size)
        args = (oa, ob, ta, tb, X_b2a)
        tail = (margin_a, margin_b, kh_a, kh_b, gap_sum)
        code = lambda c: (c & 1, (c >> 1) & 1, (c >> 2) & 1)  # noqa: E731
        for bz in range(nbz):
            for by in range(nby):
                for bx in range(nbx):
                    x0, y0, z0 = bx * sgs, by * sgs, bz * sgs
                    if not _node_survives(*args, x0, y0, z0, sgs, *tail):
                        continue
                    for c4 in range(8):
                        a = code(c4)
                        p4 = (x0 + 4 * a[0], y0 + 4 * a[1], z0 + 4 * a[2])
                        if not _node_survives(*args, *p4, 4, *tail):
                            continue
                        for c2 in range(8):
                            b = code(c2)
                            p2 = (p4[0] + 2 * b[0], p4[1] + 2 * b[1], p4[2] + 2 * b[2])
                            if not _node_survives(*args, *p2, 2, *tail):
                                continue
                            for c1 in range(8):
                                c = code(c1)
                                p1 = (p2[0] + c[0], p2[1] + c[1], p2[2] + c[2])
                                if _node_survives(*args, *p1, 1, *tail):
                                    vox.append(p1)
        if reduce is not None:
            faces = []
            for rank, (x, y, z) in enumerate(vox):
                for fi, center, normal, sep, _stiff, farea, garea, pressure in _voxel_faces(
                        oa, ob, ta, tb, X_b2a, x, y, z, tables, *tail, margin_contact_area, edge_clamp_min, full=True):
                    faces.append(dict(voxel=rank, fp=rank * MAX_MC_FACES_PER_VOXEL + fi, center=center, normal=normal, sep=sep,
                                      farea=farea, garea=garea, pressure=pressure))
            for fp, pw, nw, depth, stiff, fscale in reduce_pair_faces(faces, X_b, effective_stiffness(kh_a, kh_b), margin_contact_area,
                                                                    reduce["aabb_lo"][sb], reduce["aabb_hi"][sb], reduce["res"][sb],
                                                                    reduce.get("pre_prune", True), reduce.get("normal_matching", True),
                                                                    reduce.get("anchor_contact", False), reduce.get("moment_matching", False)):
                rows.append((pair_idx, fp, sa, sb, pw, nw, depth, stiff, fscale))
            continue
        for rank, (x, y, z) in enumerate(vox):
            for fi, center, normal, sep, stiff in _voxel_faces(oa, ob, ta, tb, X_b2a, x, y, z, tables, *tail, margin_contact_area,
                                                              edge_clamp_min):
                e = encode_oct(normal)
                rows.append((pair_idx, rank * MAX_MC_FACES_PER_VOXEL + fi, sa, sb, _x_point(X_b, center), _q_rot(X_b[3:], decode_oct(*e)),
                             sep, stiff))
    return rows, vox_all


# ---- reduce_contacts=True: aggregates per normal bin in the generate kernel (sdf_hydroelastic.py:2131-2154), local-first pruning
# (:2156-2312), HydroelasticContactReduction.reduce / export (contact_reduction_hydroelastic.py:596-755, 756-850, 983-1460) for ONE
# shape pair, the non-deterministic variant (winners ranked by score | contact id, float sums) executed in thread order: contact ids
# follow the face order, hashtable entries their first use, sums the contact / entry order.  Default options only: no anchor
# contacts, no moment matching (friction scale 1).
PRE_PRUNE_MAX_PENETRATING = 2
BETA_THRESHOLD = f32(0.0001)
SPECULATIVE_BIN_OFFSET = 128
EPS_LARGE = f32(1e-8)
MAXVAL = f32(1.0e10)


def _vadd(a, b):
    return np.array([f32(a[k] + b[k]) for k in range(3)], f32)


def _vscale(a, s):
    return np.array([f32(a[k] * s) for k in range(3)], f32)


def _vlen(a):
    return np.sqrt(_dot3(a, a), dtype=f32)


def _value_fast(score, cid):  # _make_contact_value_fast: float_flip(score) << 32 | contact id
    from oracle_reduce import float_flip  # noqa: PLC0415

    return (float_flip(f32(score)) << 32) | cid


def _normal_matching_rotation(nsum, agg, agg_mag):
    """_compute_normal_matching_rotation (:261-292) -> quaternion (x, y, z, w)."""
    q = np.array([0, 0, 0, 1], f32)
    sel_mag = _vlen(nsum)
    if sel_mag > EPS_LARGE and agg_mag > EPS_SMALL:
        sel = np.array([f32(nsum[k] / sel_mag) for k in range(3)], f32)
        ad = np.array([f32(agg[k] / agg_mag) for k in range(3)], f32)
        cr = _cross3(sel, ad)
        cr_mag = _vlen(cr)
        d = _dot3(sel, ad)
        axis, angle = None, None
        if cr_mag > EPS_LARGE:
            axis = np.array([f32(cr[k] / cr_mag) for k in range(3)], f32)
            angle = np.arccos(min(max(d, f32(-1.0)), f32(1.0)), dtype=f32)
        elif d < 0.0:
            perp = np.array([1, 0, 0], f32)
            if abs(_dot3(sel, perp)) > f32(0.9):
                perp = np.array([0, 1, 0], f32)
            c2 = _cross3(sel, perp)
            l2 = _vlen(c2)
            axis = np.array([f32(c2[k] / l2) for k in range(3)], f32) if l2 > 0 else np.zeros(3, f32)
            angle = f32(3.14159265359)
        if axis is not None:  # wp.quat_from_axis_angle
            half = f32(angle * f32(0.5))
            w, sn = np.cos(half, dtype=f32), np.sin(half, dtype=f32)
            q = np.array([f32(axis[0] * sn), f32(axis[1] * sn), f32(axis[2] * sn), w], f32)
    return q


def _normalize(v):
    ln = _vlen(v)
    return np.array([f32(v[k] / ln) for k in range(3)], f32) if ln > 0 else np.zeros(3, f32)


MIN_FRICTION_SCALE = f32(1e-2)


def reduce_pair_faces(faces, X_b, k_eff, margin_contact_area, aabb_lo, aabb_hi, res, pre_prune=True, normal_matching=True,
                      anchor_contact=False, moment_matching=False):
    """-> rows [(fingerprint, world point, world normal, separation, stiffness, friction scale)] in export order (entries in
    first-use order, winners in slot order, the entry's anchor contact last: fingerprint 0x400000 | bin)."""
    anchor_contact = anchor_contact or moment_matching
    from oracle_reduce import FACE_FRAMES, NUM_NORMAL_BINS, NUM_SPATIAL_DIRECTIONS, SPATIAL_DIRS, VALUES_PER_KEY, get_slot, voxel_index  # noqa: PLC0415
    from oracle_reduce import decode_oct as dec2  # noqa: PLC0415
    from oracle_reduce import encode_oct as enc2  # noqa: PLC0415

    entries = {}  # bin id -> dict(slots, agg...)   (insertion order = hashtable active-slot order)

    def entry(bin_id):
        if bin_id not in entries:
            entries[bin_id] = dict(bin=bin_id, slots=[0] * VALUES_PER_KEY, agg_force=np.zeros(3, f32), wps=np.zeros(3, f32), ws=f32(0.0),
                                   adv=np.zeros(3, f32), total_depth=f32(0.0), total_normal=np.zeros(3, f32), m_unr=f32(0.0),
                                   s1=f32(0.0), s2=f32(0.0))
        return entries[bin_id]

    # ---- generate: aggregates over ALL penetrating faces, buffer = all faces or the voxel-local selection
    buf = []  # contacts: dict(center, oct, sep, area, pressure, fp)
    by_voxel = {}
    for f in faces:
        by_voxel.setdefault(f["voxel"], []).append(f)
    for vox in sorted(by_voxel):
        pen = [None, None]
        nonpen, nonpen_depth = None, MAXVAL
        for f in by_voxel[vox]:
            if f["sep"] < 0.0:
                e = entry(get_slot(f["normal"]))
                fw = f32(f["farea"] * f["pressure"])
                e["agg_force"] = _vadd(e["agg_force"], _vscale(f["normal"], fw))
                e["wps"] = _vadd(e["wps"], _vscale(f["center"], fw))
                e["ws"] = f32(e["ws"] + fw)
                e["adv"] = _vadd(e["adv"], _vscale(f["normal"], f32(f["farea"] * f32(-f["sep"]))))
            if not pre_prune:
                buf.append(dict(center=f["center"], oct=enc2(f["normal"]), sep=f["sep"], area=f["farea"] if f["sep"] < 0.0 else f["garea"],
                                pressure=f["pressure"], fp=f["fp"]))
                continue
            if f["sep"] < 0.0:
                score = f32(f["farea"] * f["pressure"])
                c = dict(center=f["center"], oct=enc2(f["normal"]), sep=f["sep"], area=f["farea"], pressure=f["pressure"], fp=f["fp"], score=score)
                if pen[0] is None or score > pen[0]["score"]:
                    pen[1], pen[0] = pen[0], c
                elif pen[1] is None or score > pen[1]["score"]:
                    pen[1] = c
            elif f["sep"] < nonpen_depth:
                nonpen_depth = f["sep"]
                nonpen = dict(center=f["center"], oct=enc2(f["normal"]), sep=f["sep"], area=f["garea"], pressure=f32(0.0), fp=f["fp"])
        if pre_prune:
            buf.extend(c for c in (pen[0], pen[1], nonpen) if c is not None)
    # ---- reduce: register every buffered contact (contact id = position + 1)
    lo, hi = np.asarray(aabb_lo, f32), np.asarray(aabb_hi, f32)
    aabb_size = _vlen(np.array([f32(hi[k] - lo[k]) for k in range(3)], f32))
    nbin_of = {}
    for i, c in enumerate(buf):
        cid = i + 1
        n = dec2(c["oct"])
        depth = c["sep"]
        vox = min(max(voxel_index(c["center"], lo, hi, res), 0), 99)
        if depth >= 0.0:
            e = entry(SPECULATIVE_BIN_OFFSET + vox // VALUES_PER_KEY)
            e["slots"][vox % VALUES_PER_KEY] = max(e["slots"][vox % VALUES_PER_KEY], _value_fast(f32(-depth), cid))
            continue
        b = get_slot(n)
        e = entry(b)
        nbin_of[cid] = b
        if depth < f32(BETA_THRESHOLD * aabb_size):
            with np.errstate(all="ignore"):
                anchor = np.array([f32(e["wps"][k] / e["ws"]) for k in range(3)], f32)
            rel = np.array([f32(c["center"][k] - anchor[k]) for k in range(3)], f32)
            u, v = FACE_FRAMES[b]
            p2 = (_dot3(rel, u), _dot3(rel, v))
            pen_w = max(f32(-depth), f32(0.0))
            for d in range(NUM_SPATIAL_DIRECTIONS):
                score = f32(f32(f32(p2[0] * SPATIAL_DIRS[d][0]) + f32(p2[1] * SPATIAL_DIRS[d][1])) * pen_w)
                e["slots"][d] = max(e["slots"][d], _value_fast(score, cid))
        e["slots"][NUM_SPATIAL_DIRECTIONS] = max(e["slots"][NUM_SPATIAL_DIRECTIONS], _value_fast(f32(-depth), cid))
        if moment_matching and e["ws"] > EPS_SMALL:  # unreduced friction moment about the centre of pressure (:717-727)
            anchor = np.array([f32(e["wps"][k] / e["ws"]) for k in range(3)], f32)
            lever = _vlen(_cross3(np.array([f32(c["center"][k] - anchor[k]) for k in range(3)], f32), n))
            e["m_unr"] = f32(e["m_unr"] + f32(f32(c["area"] * c["pressure"]) * lever))
        ev = entry(NUM_NORMAL_BINS + vox // VALUES_PER_KEY)
        ev["slots"][vox % VALUES_PER_KEY] = max(ev["slots"][vox % VALUES_PER_KEY], _value_fast(f32(-depth), cid))

    def winners(e):
        out = []
        for v in e["slots"]:
            if v and (v & 0xFFFFFFFF) not in out:
                out.append(v & 0xFFFFFFFF)
        return out

    # ---- accumulate_reduced_depth_kernel
    for e in entries.values():
        for cid in winners(e):
            c = buf[cid - 1]
            if c["sep"] < 0.0:
                nb = e["bin"] if e["bin"] < NUM_NORMAL_BINS else nbin_of.get(cid, -1)
                if nb >= 0:
                    pen = f32(-c["sep"])
                    t = entries[nb]
                    t["total_depth"] = f32(t["total_depth"] + pen)
                    t["total_normal"] = _vadd(t["total_normal"], _vscale(dec2(c["oct"]), pen))
    def reliable_of(t):
        mag = _vlen(t["agg_force"])
        return mag, bool(_vlen(t["adv"]) > EPS_LARGE and mag > EPS_SMALL)

    # ---- accumulate_moments_kernel (:852-980): reduced friction moments S1 = sum pen * lever, S2 = sum pen * lever^2
    if moment_matching:
        for e in entries.values():
            for cid in winners(e):
                c = buf[cid - 1]
                if not c["sep"] < 0.0:
                    continue
                nb = e["bin"] if e["bin"] < NUM_NORMAL_BINS else nbin_of.get(cid, -1)
                if nb < 0:
                    continue
                t = entries[nb]
                if not t["ws"] > EPS_SMALL:
                    continue
                anchor = np.array([f32(t["wps"][k] / t["ws"]) for k in range(3)], f32)
                n = dec2(c["oct"])
                if normal_matching:
                    mag, rel = reliable_of(t)
                    if rel:
                        n = _normalize(_q_rot(_normal_matching_rotation(t["total_normal"], t["agg_force"], mag), n))
                lever = _vlen(_cross3(np.array([f32(c["center"][k] - anchor[k]) for k in range(3)], f32), n))
                pen = f32(-c["sep"])
                t["s1"] = f32(t["s1"] + f32(pen * lever))
                t["s2"] = f32(t["s2"] + f32(f32(pen * lever) * lever))
    # ---- export
    rows = []
    mca_k = f32(f32(margin_contact_area) * k_eff)
    for e in entries.values():
        ids = winners(e)
        if not ids:
            continue
        agg_mag, reliable = reliable_of(e)
        max_pen = f32(0.0)
        for cid in ids:
            if buf[cid - 1]["sep"] < 0.0:
                max_pen = max(max_pen, f32(-buf[cid - 1]["sep"]))
        add_anchor, anchor_pos = 0, np.zeros(3, f32)
        if anchor_contact and reliable and max_pen > 0.0 and e["ws"] > EPS_SMALL:
            anchor_pos = np.array([f32(e["wps"][k] / e["ws"]) for k in range(3)], f32)
            add_anchor = 1
        anchor_depth = max_pen
        rot = np.array([0, 0, 0, 1], f32)
        if normal_matching and reliable:
            rot = _normal_matching_rotation(e["total_normal"], e["agg_force"], agg_mag)
        if normal_matching:
            eff = _vlen(e["total_normal"])
            if eff < EPS_LARGE:
                eff = e["total_depth"]
        else:
            eff = e["total_depth"]
        tdwa = f32(eff + f32(f32(add_anchor) * anchor_depth))
        shared = f32(agg_mag / tdwa) if (agg_mag > EPS_SMALL and tdwa > 0.0) else f32(0.0)
        alpha, l_avg, uniform_fs, anchor_fs = f32(0.0), f32(0.0), f32(1.0), f32(1.0)
        if moment_matching:
            m_unr, m_red, m_red2 = e["m_unr"], e["s1"], e["s2"]
            s0 = f32(e["total_depth"] + f32(f32(add_anchor) * anchor_depth))
            if m_unr > EPS_SMALL and s0 > EPS_SMALL and m_red > EPS_SMALL and agg_mag > EPS_SMALL:
                m_target = f32(f32(m_unr * tdwa) / agg_mag)
                if m_target < m_red:
                    uniform_fs = f32(m_target / m_red)
                else:
                    l_avg = f32(m_red / s0)
                    variance = f32(f32(m_red2 * s0) - f32(m_red * m_red))
                    if variance > EPS_SMALL:
                        alpha = min(max(f32(f32(f32(m_target - m_red) * m_red) / variance), f32(0.0)), f32(1.0))
            if add_anchor == 1 and anchor_depth > 0.0:
                anchor_fs = max(MIN_FRICTION_SCALE,
                                f32(f32(f32(1.0) + f32(f32(e["total_depth"] / anchor_depth) * f32(f32(1.0) - uniform_fs))) - alpha))
        for cid in ids:
            c = buf[cid - 1]
            depth, n = c["sep"], dec2(c["oct"])
            final = n
            fscale = f32(1.0)
            if reliable:
                if normal_matching and depth < 0.0:
                    final = _normalize(_q_rot(rot, n))
                stiff = shared
                if shared == 0.0:
                    stiff = f32(f32(c["area"] * c["pressure"]) / max(f32(-depth), EPS_SMALL)) if depth < 0.0 else mca_k
                if moment_matching and depth < 0.0:
                    if l_avg > EPS_SMALL:
                        lever = _vlen(_cross3(np.array([f32(c["center"][k] - anchor_pos[k]) for k in range(3)], f32), final))
                        fscale = max(MIN_FRICTION_SCALE, f32(f32(1.0) + f32(f32(alpha * f32(lever - l_avg)) / l_avg)))
                    else:
                        fscale = uniform_fs
            else:
                nb = nbin_of.get(cid, -1)
                if nb >= 0 and depth < 0.0:
                    t = entries[nb]
                    t_mag, t_rel = reliable_of(t)
                    if normal_matching and t_rel:
                        final = _normalize(_q_rot(_normal_matching_rotation(t["total_normal"], t["agg_force"], t_mag), n))
                    if normal_matching:
                        t_eff0 = _vlen(t["total_normal"])
                        if t_eff0 < EPS_LARGE:
                            t_eff0 = t["total_depth"]
                    else:
                        t_eff0 = t["total_depth"]
                    t_eff, t_anchor_depth = t_eff0, f32(0.0)
                    if anchor_contact and t_rel:
                        v = t["slots"][NUM_SPATIAL_DIRECTIONS]
                        if v:
                            md = buf[(v & 0xFFFFFFFF) - 1]["sep"]
                            if md < 0.0:
                                t_anchor_depth = f32(-md)
                        if t["ws"] > EPS_SMALL and t_anchor_depth > 0.0:
                            t_eff = f32(t_eff0 + t_anchor_depth)
                    if t_mag > EPS_SMALL and t_eff > 0.0:
                        stiff = f32(t_mag / t_eff)
                    else:
                        stiff = f32(f32(c["area"] * c["pressure"]) / max(f32(-depth), EPS_SMALL))
                    if moment_matching:
                        v_unr, v_s1, v_s2 = t["m_unr"], t["s1"], t["s2"]
                        v_s0 = f32(t["total_depth"] + t_anchor_depth)
                        if v_unr > EPS_SMALL and v_s0 > EPS_SMALL and v_s1 > EPS_SMALL and t_mag > EPS_SMALL:
                            v_target = f32(f32(v_unr * t_eff) / t_mag)
                            if v_target < v_s1:
                                fscale = f32(v_target / v_s1)
                            else:
                                v_lavg = f32(v_s1 / v_s0)
                                v_var = f32(f32(v_s2 * v_s0) - f32(v_s1 * v_s1))
                                v_alpha = f32(0.0)
                                if v_var > EPS_SMALL:
                                    v_alpha = min(max(f32(f32(f32(v_target - v_s1) * v_s1) / v_var), f32(0.0)), f32(1.0))
                                v_anchor = np.zeros(3, f32)
                                if t["ws"] > EPS_SMALL:
                                    v_anchor = np.array([f32(t["wps"][k] / t["ws"]) for k in range(3)], f32)
                                v_lever = _vlen(_cross3(np.array([f32(c["center"][k] - v_anchor[k]) for k in range(3)], f32), final))
                                if v_lavg > EPS_SMALL:
                                    fscale = max(MIN_FRICTION_SCALE, f32(f32(1.0) + f32(f32(v_alpha * f32(v_lever - v_lavg)) / v_lavg)))
                elif depth < 0.0:
                    stiff = f32(f32(c["area"] * c["pressure"]) / max(f32(-depth), EPS_SMALL))
                else:
                    stiff = mca_k
            if depth >= 0.0:
                stiff, fscale = mca_k, f32(1.0)
            rows.append((c["fp"], _x_point(X_b, c["center"]), _q_rot(X_b[3:], final), depth, stiff, fscale))
        if add_anchor == 1:
            rows.append((0x400000 | e["bin"], _x_point(X_b, anchor_pos), _q_rot(X_b[3:], _normalize(e["agg_force"])), f32(-anchor_depth),
                         shared, anchor_fs))
    return rows


def _voxel_faces(oa, ob, ta, tb, X_b2a, x, y, z, tables, margin_a, margin_b, kh_a, kh_b, gap_sum, margin_contact_area, edge_clamp_min,
                 full=False):
    """mc_iterate_voxel_vertices + mc_calc_face_texture + the face filters / stiffness of generate + decode for one voxel of B:
    yields (face index, centre in B's frame, normal in B's frame, separation, stiffness)."""
    tri_range, flat = tables
    cmin, cmax = f32(edge_clamp_min), f32(1.0 - edge_clamp_min)
    vs = tb.voxel_size.astype(f32)
    step = [_q_rot(X_b2a[3:], np.array([vs[0], 0, 0], dtype=f32)), _q_rot(X_b2a[3:], np.array([0, vs[1], 0], dtype=f32)),
            _q_rot(X_b2a[3:], np.array([0, 0, vs[2]], dtype=f32))]
    base_b = (ob.lo + np.array([x, y, z], dtype=f32) * vs).astype(f32)
    base_a = _x_point(X_b2a, base_b)
    cube, any_gap = 0, False
    cv, cs_self, cs_other = np.zeros(8, f32), np.zeros(8, f32), np.zeros(8, f32)
    for i in range(8):
        ox, oy, oz = CORNER[0][i], CORNER[1][i], CORNER[2][i]
        pa = (((base_a + f32(ox) * step[0]).astype(f32) + f32(oy) * step[1]).astype(f32) + f32(oz) * step[2]).astype(f32)
        v_self = tb.sample_at_voxel([[x + ox, y + oy, z + oz]])[0]
        v_other = oa.sample(pa)
        if np.isnan(v_self) or np.isnan(v_other):
            return
        es, eo = f32(v_self - margin_b), f32(v_other - margin_a)
        vd = f32(f32(-kh_a * eo) - f32(-kh_b * es))
        cv[i], cs_self[i], cs_other[i] = vd, es, eo
        if vd < 0.0:
            cube |= 1 << i
        if f32(es + eo) <= gap_sum:
            any_gap = True
    if not any_gap:
        return
    t0, t1 = int(tri_range[cube]), int(tri_range[cube + 1])
    for fi in range((t1 - t0) // 3):
        verts, v_sdf, v_sep, n_in = np.zeros((3, 3), f32), np.zeros(3, f32), np.zeros(3, f32), 0
        for vi in range(3):
            a, b = int(flat[t0 + 3 * fi + vi][0]), int(flat[t0 + 3 * fi + vi][1])
            vd = f32(cv[b] - cv[a])
            t = f32(0.5) if abs(vd) < MC_EDGE_VAL_DIFF_EPS else min(max(f32(f32(f32(0.0) - cv[a]) / vd), cmin), cmax)
            p0 = np.array([CORNER[0][a], CORNER[1][a], CORNER[2][a]], dtype=f32)
            p1 = np.array([CORNER[0][b], CORNER[1][b], CORNER[2][b]], dtype=f32)
            vol = ((p0 + (t * (p1 - p0)).astype(f32)).astype(f32) + np.array([x, y, z], dtype=f32)).astype(f32)
            verts[vi] = (ob.lo + (vol * vs).astype(f32)).astype(f32)
            s_self = f32(cs_self[a] + f32(t * f32(cs_self[b] - cs_self[a])))
            s_other = f32(cs_other[a] + f32(t * f32(cs_other[b] - cs_other[a])))
            v_sdf[vi], v_sep[vi] = s_self, f32(s_self + s_other)
            if v_sep[vi] < 0.0:
                n_in += 1
        n = _cross3((verts[1] - verts[0]).astype(f32), (verts[2] - verts[0]).astype(f32))
        n_sq = _dot3(n, n)
        if n_sq < MC_DEGENERATE_N_SQ_EPS:
            garea, normal = f32(0.0), np.array([0, 0, 1], dtype=f32)
        else:
            inv = f32(f32(1.0) / np.sqrt(n_sq))
            normal, garea = (n * inv).astype(f32), f32(f32(n_sq * inv) * f32(0.5))
        center = (((verts[0] + verts[1]).astype(f32) + verts[2]).astype(f32) / f32(3.0)).astype(f32)
        adj = f32(f32(f32(v_sdf[0] + v_sdf[1]) + v_sdf[2]) / f32(3.0))
        sep = f32(f32(f32(v_sep[0] + v_sep[1]) + v_sep[2]) / f32(3.0))
        farea = f32(garea * triangle_fraction(v_sep, n_in))
        if garea <= 0.0 or classify(sep, gap_sum) > 0:
            continue
        pressure = max(f32(-kh_b * adj), f32(0.0)) if sep < 0.0 else f32(0.0)
        area = farea if sep < 0.0 else garea
        if sep < 0.0:
            stiff = f32(f32(area * pressure) / max(f32(-sep), EPS_SMALL))
        else:
            stiff = f32(f32(margin_contact_area) * effective_stiffness(kh_a, kh_b))
        if full:
            yield fi, center, normal, sep, stiff, farea, garea, pressure
        else:
            yield fi, center, normal, sep, stiff
