"""TEST INFRASTRUCTURE ONLY -- float32 restatement of the reference's hydroelastic contact generation for SDF pairs
(newton/_src/geometry/sdf_hydroelastic.py), unreduced path (reduce_contacts=False: generate -> decode):
  get_effective_stiffness :216-222, linear_pressure :237-248, classify_hydroelastic_contact :140-145
  mc_iterate_voxel_vertices :1716-1798, mc_calc_face_texture :282-362, get_triangle_fraction sdf_mc.py:112-162
  generate_contacts_kernel :1982-2140 (pre_prune off), decode_contacts_kernel :1823-1928, pair normalisation :1329-1376
The reference finds the voxels that carry iso-surface faces with a block broad phase + octree refinement (:1026-1170); this
restatement visits every voxel of the finer SDF's grid inside the other SDF's box, which yields the same voxel set as long as
that search has no false negatives (its purpose).  Marching-cubes case tables: newton_amd/mc_tables.py (Warp's own table is not
in /root/reference; triangulations of a case may differ, the surface does not).  The octahedral normal encoding of the contact
buffer (a storage format) is skipped.  Only tests/ may import this."""
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
