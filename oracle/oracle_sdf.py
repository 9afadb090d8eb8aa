"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy / plain Python loops, small cases) of the reference's texture-SDF sampler
and mesh-vs-SDF narrow phase.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows, function by function (paths under /root/reference/newton/_src/geometry):
  sample / sample_clamped   sdf_texture.py:786-828 (_locate_cell_coords) + :1008-1126 (_texture_sample_sdf_variant, software trilinear)
  sample_hw(_clamped)       sdf_texture.py:1415-1538 (one filtered fetch at a fractional texture coordinate: what the mesh-SDF narrow
                            phase samples with; on Warp's CPU device a float trilinear blend)
  sample_grad_fd            sdf_texture.py:1619-1697 (_texture_sample_sdf_grad_hw_impl_variant: six hw fetches, centred differences)
  edge_search               sdf_contact.py:704-938 (do_edge_sdf_collision, texture-only variant: golden pair + 3 Brent steps)
  mesh_sdf_collide          sdf_contact.py:1098-1515 (mesh_sdf_collision_kernel with reduce_contacts=False), helpers :80-135,154-182
All arithmetic in numpy.float32, one operation per statement, in the reference's order.  PINNED by tests/golden/sdf_reference_vectors.npz
(tests/golden/make_sdf_reference_vectors.py EXECUTES the reference's samplers, do_edge_sdf_collision and mesh_sdf_collision_kernel on
the stand-in of tests/golden/refshim): tests/test_sdf_reference_vectors.py holds every function below against that record bit for
bit.  What stays restated is Warp's native texture fetch and vector builtins (not part of /root/reference).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
SLOT_LINEAR = np.uint32(0xFFFFFFFE)


def _clampf(v, lo, hi):
    return min(max(f32(v), f32(lo)), f32(hi))


class OracleSDF:
    """Scalar (one point at a time) sampler over the arrays of a newton_amd.sdf.TextureSDF."""

    def __init__(self, t):
        self.t = t
        self.cx, self.cy, self.cz = t.slots.shape
        self.scale = {np.dtype(np.float32): f32(1.0), np.dtype(np.uint16): f32(1.0) / f32(65535.0),
                      np.dtype(np.uint8): f32(1.0) / f32(255.0)}[t.subgrid.dtype]
        self.lo, self.hi, self.inv_dx = t.box_lower.astype(f32), t.box_upper.astype(f32), t.inv_dx.astype(f32)
        self.vmin, self.vrange = f32(t.min_value), f32(t.value_range)
        self.ss = int(t.subgrid_size)

    def clamp(self, p):
        return np.array([_clampf(p[0], self.lo[0], self.hi[0]), _clampf(p[1], self.lo[1], self.hi[1]),
                         _clampf(p[2], self.lo[2], self.hi[2])], dtype=f32)

    def sample_clamped(self, clamped, diff_mag):
        t = self.t
        f = (clamped - self.lo) * self.inv_dx
        ssf = f32(self.ss)
        fv = [f32(self.cx) * ssf, f32(self.cy) * ssf, f32(self.cz) * ssf]
        fc = [_clampf(f[k], 0.0, fv[k]) for k in range(3)]
        i = [min(max(int(np.floor(fc[k])), 0), int(fv[k]) - 1) for k in range(3)]
        tt = [f32(fc[k] - f32(i[k])) for k in range(3)]
        f2c = f32(1.0) / ssf
        b = [min(max(int(f32(i[k]) * f2c), 0), (self.cx, self.cy, self.cz)[k] - 1) for k in range(3)]
        slot = t.slots[b[0], b[1], b[2]]
        if slot >= SLOT_LINEAR:
            tt = [f32(f32(f32(i[k]) + tt[k]) * f2c - f32(b[k])) for k in range(3)]
            g = t.coarse
            v = [g[b[2] + dz, b[1] + dy, b[0] + dx] for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)]
            scale = False
        else:
            s = int(slot)
            spd = self.ss + 1
            o = [(s & 0x3FF) * spd + (i[0] - b[0] * self.ss), ((s >> 10) & 0x3FF) * spd + (i[1] - b[1] * self.ss),
                 ((s >> 20) & 0x3FF) * spd + (i[2] - b[2] * self.ss)]
            v = [f32(t.subgrid[o[2] + dz, o[1] + dy, o[0] + dx]) * self.scale for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)]
            scale = True
        v000, v100, v010, v110, v001, v101, v011, v111 = [f32(x) for x in v]
        tx, ty, tz = tt
        c00 = f32(v000 + f32(f32(v100 - v000) * tx))
        c10 = f32(v010 + f32(f32(v110 - v010) * tx))
        c01 = f32(v001 + f32(f32(v101 - v001) * tx))
        c11 = f32(v011 + f32(f32(v111 - v011) * tx))
        c0 = f32(c00 + f32(f32(c10 - c00) * ty))
        c1 = f32(c01 + f32(f32(c11 - c01) * ty))
        val = f32(c0 + f32(f32(c1 - c0) * tz))
        if scale:
            val = f32(f32(val * self.vrange) + self.vmin)
        return f32(val + f32(diff_mag))

    def sample(self, p):
        p = np.asarray(p, dtype=f32)
        c = self.clamp(p)
        d = p - c
        dsq = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
        return self.sample_clamped(c, np.sqrt(dsq))

    # ---- the "hardware" samplers the mesh-SDF narrow phase uses (sdf_texture.py:1267-1557): ONE filtered fetch at a fractional
    # texture coordinate.  On Warp's CPU device the fetch is a float32 trilinear blend of the eight texels around
    # (u - 0.5, v - 0.5, w - 0.5) -- restated in tests/golden/refshim/warp.texture_sample; the coordinate round trip
    # (block origin + 0.5 + t, then - 0.5, floor, fraction) is part of the result: it costs the low bits of t.
    def _fetch(self, tex, norm, dims, u):
        c = []
        for k in range(3):
            x = f32(f32(u[k]) - f32(0.5))
            i0 = int(np.floor(x))
            c.append((i0, f32(x - f32(i0))))
        (x0, tx), (y0, ty), (z0, tz) = c

        def t(dx, dy, dz):
            x, y, z = (min(max(v, 0), n - 1) for v, n in zip((x0 + dx, y0 + dy, z0 + dz), dims))
            return f32(f32(tex[z, y, x]) * norm) if norm is not None else f32(tex[z, y, x])

        c00 = f32(t(0, 0, 0) + f32(f32(t(1, 0, 0) - t(0, 0, 0)) * tx))
        c10 = f32(t(0, 1, 0) + f32(f32(t(1, 1, 0) - t(0, 1, 0)) * tx))
        c01 = f32(t(0, 0, 1) + f32(f32(t(1, 0, 1) - t(0, 0, 1)) * tx))
        c11 = f32(t(0, 1, 1) + f32(f32(t(1, 1, 1) - t(0, 1, 1)) * tx))
        c0 = f32(c00 + f32(f32(c10 - c00) * ty))
        c1 = f32(c01 + f32(f32(c11 - c01) * ty))
        return f32(c0 + f32(f32(c1 - c0) * tz))

    def sample_hw_clamped(self, clamped, diff_mag):
        """_texture_sample_sdf_hw_clamped_variant (:1415-1461)."""
        t = self.t
        f = (clamped - self.lo) * self.inv_dx
        ssf = f32(self.ss)
        fv = [f32(self.cx) * ssf, f32(self.cy) * ssf, f32(self.cz) * ssf]
        fc = [_clampf(f[k], 0.0, fv[k]) for k in range(3)]
        i = [min(max(int(np.floor(fc[k])), 0), int(fv[k]) - 1) for k in range(3)]
        tt = [f32(fc[k] - f32(i[k])) for k in range(3)]
        f2c = f32(1.0) / ssf
        b = [min(max(int(f32(i[k]) * f2c), 0), (self.cx, self.cy, self.cz)[k] - 1) for k in range(3)]
        slot = t.slots[b[0], b[1], b[2]]
        if slot >= SLOT_LINEAR:
            u = []
            for k in range(3):
                cf = f32(f32(f32(i[k]) + tt[k]) * f2c)
                cb = f32(b[k])
                u.append(f32(f32(cb + f32(cf - cb)) + f32(0.5)))
            g = t.coarse
            val = self._fetch(g, None, (g.shape[2], g.shape[1], g.shape[0]), u)
        else:
            s = int(slot)
            samples = f32(self.ss + 1)
            blk = [f32(s & 0x3FF), f32((s >> 10) & 0x3FF), f32((s >> 20) & 0x3FF)]
            u = []
            for k in range(3):
                l = f32(f32(i[k]) - f32(f32(b[k]) * ssf))
                o = f32(f32(f32(blk[k] * samples) + l) + f32(0.5))
                u.append(f32(o + tt[k]))
            sg = t.subgrid
            raw = self._fetch(sg, None if sg.dtype == np.float32 else self.scale, (sg.shape[2], sg.shape[1], sg.shape[0]), u)
            val = f32(f32(raw * self.vrange) + self.vmin)
        return f32(val + f32(diff_mag))

    def sample_hw(self, p):
        """texture_sample_sdf_hw (:1495-1538)."""
        p = np.asarray(p, dtype=f32)
        c = self.clamp(p)
        d = p - c
        mag = f32(0.0)
        if d[0] != 0.0 or d[1] != 0.0 or d[2] != 0.0:
            mag = np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])))
        return self.sample_hw_clamped(c, mag)

    def sample_grad_fd(self, p):
        p = np.asarray(p, dtype=f32)
        c = self.clamp(p)
        d = p - c
        if d[0] != 0.0 or d[1] != 0.0 or d[2] != 0.0:
            m = np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])))
            if m > 0.0:
                return (d / m).astype(f32)
        g = np.zeros(3, dtype=f32)
        for a in range(3):
            h = f32(0.5) / self.inv_dx[a]
            p0, p1 = p.copy(), p.copy()
            p0[a] = f32(p[a] + h)
            p1[a] = f32(p[a] - h)
            c0, c1 = _clampf(p0[a], self.lo[a], self.hi[a]), _clampf(p1[a], self.lo[a], self.hi[a])
            d0, d1 = f32(p0[a] - c0), f32(p1[a] - c1)
            q0, q1 = p0.copy(), p1.copy()
            q0[a], q1[a] = c0, c1
            v0 = self.sample_hw_clamped(q0, np.sqrt(f32(d0 * d0)) if f32(d0 * d0) != 0.0 else f32(0.0))
            v1 = self.sample_hw_clamped(q1, np.sqrt(f32(d1 * d1)) if f32(d1 * d1) != 0.0 else f32(0.0))
            g[a] = f32(f32(v0 - v1) * self.inv_dx[a])
        return g


def edge_search(sdf: OracleSDF, v0, v1, midpoint_sdf, precision_target):
    """do_edge_sdf_collision -> (distance, point, endpoint code 0 interior / 1 v0 / 2 v1)."""
    golden = f32(0.3819660112501051)
    v0, v1 = np.asarray(v0, dtype=f32), np.asarray(v1, dtype=f32)
    e = v1 - v0
    len_sq = f32(f32(f32(e[0] * e[0]) + f32(e[1] * e[1])) + f32(e[2] * e[2]))
    inv_len = f32(1.0e12)
    if len_sq > 0.0:
        inv_len = f32(1.0) / np.sqrt(len_sq)
    tol_floor = f32(f32(f32(0.5) * f32(precision_target)) * inv_len)
    at = lambda t: sdf.sample_hw(v0 + e * f32(t))  # noqa: E731
    a, b, x, w, v = f32(0.0), f32(1.0), f32(0.5), f32(0.5), f32(0.5)
    fx = f32(midpoint_sdf)
    fw = fv = fx
    d_step = e_step = f32(0.0)
    if tol_floor < 0.25:
        offset = f32(f32(0.5) * golden)
        left, right = f32(f32(0.5) - offset), f32(f32(0.5) + offset)
        f_left, f_right = at(left), at(right)
        if f_left < fx and f_left <= f_right:
            b, x, fx, w, fw, v, fv = f32(0.5), left, f_left, f32(0.5), f32(midpoint_sdf), right, f_right
        elif f_right < fx:
            a, x, fx, w, fw, v, fv = f32(0.5), right, f_right, f32(0.5), f32(midpoint_sdf), left, f_left
        else:
            a, b, w, fw, v, fv = left, right, left, f_left, right, f_right
    for _ in range(3):
        m = f32(f32(0.5) * f32(a + b))
        tol = max(f32(f32(f32(1.0e-2) * abs(x)) + f32(1.0e-8)), tol_floor)
        tol2 = f32(f32(2.0) * tol)
        if abs(f32(x - m)) <= f32(tol2 - f32(f32(0.5) * f32(b - a))):
            break
        parabolic, trial = False, f32(0.0)
        if abs(e_step) > tol:
            r = f32(f32(x - w) * f32(fx - fv))
            q = f32(f32(x - v) * f32(fx - fw))
            pnum = f32(f32(f32(x - v) * q) - f32(f32(x - w) * r))
            q = f32(f32(2.0) * f32(q - r))
            if q > 0.0:
                pnum = -pnum
            else:
                q = -q
            if abs(pnum) < f32(f32(0.5) * abs(f32(q * e_step))):
                trial = f32(pnum / q)
                u_trial = f32(x + trial)
                if f32(u_trial - a) >= tol2 and f32(b - u_trial) >= tol2:
                    parabolic = True
        if parabolic:
            e_step, d_step = d_step, trial
        else:
            e_step = f32(a - x) if x >= m else f32(b - x)
            d_step = f32(golden * e_step)
        if abs(d_step) >= tol:
            u = f32(x + d_step)
        else:
            u = f32(x + tol) if d_step > 0.0 else f32(x - tol)
        fu = at(u)
        if fu <= fx:
            if u < x:
                b = x
            else:
                a = x
            v, fv, w, fw, x, fx = w, fw, x, fx, u, fu
        else:
            if u < x:
                a = u
            else:
                b = u
            if fu <= fw or w == x:
                v, fv, w, fw = w, fw, u, fu
            elif fu <= fv or v == x or v == w:
                v, fv = u, fu
    best_endpoint, best_t, best_f = 0, x, fx
    if a == 0.0:
        fe = at(0.0)
        if fe < best_f:
            best_t, best_f, best_endpoint = f32(0.0), fe, 1
    if b == 1.0:
        fe = at(1.0)
        if fe < best_f:
            best_t, best_f, best_endpoint = f32(1.0), fe, 2
    return f32(best_f), (v0 + e * f32(best_t)).astype(f32), best_endpoint


def _dot3(a, b):  # wp.dot: left to right, one float32 rounding per operation (np.dot may reorder / fuse through BLAS)
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross3(a, b):
    return np.array([f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
                     f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))], dtype=f32)


def _q_rot(q, v):
    """wp.quat_rotate in float32 (oracle/wp_builtins.h order: v*(2w^2-1) + cross(qv, v)*w*2 + qv*dot(qv, v)*2)."""
    q, v = np.asarray(q, dtype=f32), np.asarray(v, dtype=f32)
    qv, w = q[:3], q[3]
    c, d = _cross3(qv, v), _dot3(qv, v)
    k = f32(f32(f32(f32(2.0) * w) * w) - f32(1.0))
    return np.array([f32(f32(f32(v[i] * k) + f32(f32(c[i] * w) * f32(2.0))) + f32(f32(qv[i] * d) * f32(2.0))) for i in range(3)],
                    dtype=f32)


def _q_mul(a, b):
    a, b = np.asarray(a, dtype=f32), np.asarray(b, dtype=f32)
    m = lambda x, y: f32(x * y)  # noqa: E731
    return np.array([f32(f32(f32(m(a[3], b[0]) + m(b[3], a[0])) + m(a[1], b[2])) - m(b[1], a[2])),
                     f32(f32(f32(m(a[3], b[1]) + m(b[3], a[1])) + m(a[2], b[0])) - m(b[2], a[0])),
                     f32(f32(f32(m(a[3], b[2]) + m(b[3], a[2])) + m(a[0], b[1])) - m(b[0], a[1])),
                     f32(f32(f32(m(a[3], b[3]) - m(a[0], b[0])) - m(a[1], b[1])) - m(a[2], b[2]))], dtype=f32)


def _x_mul(a, b):  # transform_multiply
    return np.concatenate([_q_rot(a[3:], b[:3]) + a[:3], _q_mul(a[3:], b[3:])]).astype(f32)


def _x_inv(t):
    qi = np.array([-t[3], -t[4], -t[5], t[6]], dtype=f32)
    return np.concatenate([-_q_rot(qi, t[:3]), qi]).astype(f32)


def _x_point(t, p):
    return (t[:3] + _q_rot(t[3:], p)).astype(f32)


def mesh_sdf_collide(pairs, shape_transform, shape_data, shape_gap, shape_sdf_index, sdfs, shape_edge_range, edge_centers, edge_halves):
    """mesh_sdf_collision_kernel, reduce_contacts=False -> list of (pair_idx, key, centre[3], normal[3], distance, margin0, margin1)
    in (pair, mode, edge) order."""
    out = []
    X = np.asarray(shape_transform, dtype=f32)
    D = np.asarray(shape_data, dtype=f32)
    for pair_idx, (s0, s1) in enumerate(np.asarray(pairs).reshape(-1, 2)):
        gap_sum = f32(f32(shape_gap[s0]) + f32(shape_gap[s1]))
        for mode in range(2):
            tri, sd = (s0, s1) if mode == 0 else (s1, s0)
            idx = int(shape_sdf_index[sd])
            e0, ne = int(shape_edge_range[tri][0]), int(shape_edge_range[tri][1])
            if idx < 0 or idx >= len(sdfs) or ne <= 0 or sdfs[idx] is None:
                continue
            t = sdfs[idx]
            o = OracleSDF(t)
            sdf_scale = np.ones(3, dtype=f32) if t.scale_baked else D[sd, :3].copy()
            X_tri, X_sdf = X[tri], X[sd]
            X_m2s = _x_mul(_x_inv(X_sdf), X_tri)
            tri_margin, sdf_margin = D[tri, 3], D[sd, 3]
            eps = f32(1.0e-10)
            g = np.array([s if abs(s) > eps else (eps if s >= 0.0 else -eps) for s in sdf_scale], dtype=f32)
            inv_scale = (f32(1.0) / g).astype(f32)
            min_scale = f32(np.min(np.abs(g)))
            radius_scale = f32(np.max(np.abs(inv_scale)))
            contact_threshold = f32(f32(gap_sum + tri_margin) + sdf_margin)
            thr_u = f32(contact_threshold / min_scale)
            inner = f32(tri_margin + sdf_margin)
            precision = min(f32(inner / min_scale), f32(t.voxel_radius))
            for e in range(ne):
                ec, eh = np.asarray(edge_centers[e0 + e], dtype=f32), np.asarray(edge_halves[e0 + e], dtype=f32)
                center = (_x_point(X_m2s, ec[:3]) * inv_scale).astype(f32)
                threshold = f32(f32(ec[3] * radius_scale) + thr_u)
                cl = np.minimum(np.maximum(center, o.lo), o.hi)
                dd = center - cl
                d2 = f32(f32(f32(dd[0] * dd[0]) + f32(dd[1] * dd[1])) + f32(dd[2] * dd[2]))
                if d2 > f32(threshold * threshold):
                    continue
                mid = o.sample_hw_clamped(cl, np.sqrt(d2) if d2 > 0.0 else f32(0.0))
                if not (mid <= threshold):
                    continue
                c_loc = _x_point(X_m2s, ec[:3])
                h_loc = _q_rot(X_m2s[3:], eh[:3])
                own = int(eh[3])
                v0, v1 = ((c_loc - h_loc) * inv_scale).astype(f32), ((c_loc + h_loc) * inv_scale).astype(f32)
                dist_u, p_u, endpoint = edge_search(o, v0, v1, mid, precision)
                dist_approx = f32(dist_u * min_scale)
                consistent = True
                if dist_approx < inner:
                    ic = ((v0 + v1) * f32(0.5)).astype(f32)
                    ir = f32(np.sqrt(_dot3(v1 - v0, v1 - v0)) * f32(0.5))
                    cr = f32(ir + f32(inner / min_scale))
                    icl = np.minimum(np.maximum(ic, o.lo), o.hi)
                    if _dot3(ic - icl, ic - icl) > f32(cr * cr):
                        consistent = False
                    else:
                        consistent = bool(mid <= cr)
                owns = endpoint == 0 or own == 0 or (own & endpoint) != 0
                if not (dist_approx < contact_threshold and consistent and owns):
                    continue
                dir_u = o.sample_grad_fd(p_u)
                dist = f32(dist_u * min_scale)
                direction = (dir_u * inv_scale).astype(f32)
                point = (p_u * sdf_scale).astype(f32)
                pw = _x_point(X_sdf, point)
                dw = _q_rot(X_sdf[3:], direction)
                dl2 = _dot3(dw, dw)
                if dl2 > 0.0:
                    dw = (dw * (f32(1.0) / np.sqrt(dl2))).astype(f32)
                else:
                    fb = pw - X_sdf[:3]
                    fl2 = _dot3(fb, fb)
                    dw = (fb * (f32(1.0) / np.sqrt(fl2))).astype(f32) if fl2 > 0.0 else np.array([0.0, 1.0, 0.0], dtype=f32)
                n = -dw if mode == 0 else dw
                out.append((pair_idx, (e << 2) | (mode << 1), pw, n.astype(f32), dist, f32(D[s0, 3]), f32(D[s1, 3])))
    return out
