"""TEST INFRASTRUCTURE ONLY (never imported by newton_amd/ or by bench.py's timed path).

float32 restatement of the two reference functions behind newton_amd/csrc/nt_flat_contacts.hip:
  * write_contact        newton/_src/sim/collide.py:203-254 (+ _write_contact_at_index :165-201), radius_eff = 0 (meshes)
  * eval_body_contact    newton/_src/solvers/semi_implicit/kernels_contact.py:381-556 (force_in_world_frame=False)
Builtin operation order as in oracle/wp_builtins.h (quat_rotate: v*(2w^2-1) + cross(qv, v)*w*2 + qv*dot(qv, v)*2).  Pinned by
tests/golden/flat_contact_reference_vectors.npz, recorded from the reference's own functions executed on the stand-in."""
import numpy as np

f32 = np.float32


def _v(*x):
    return np.array(x, np.float32)


def _dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross(a, b):
    return _v(f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0]))


def _rot(q, v, sign=f32(1.0)):
    qv, w = q[:3], q[3]
    k = f32(f32(f32(f32(2.0) * w) * w) - f32(1.0))
    c, d = _cross(qv, v), _dot(qv, v)
    out = np.zeros(3, np.float32)
    for i in range(3):
        t = f32(f32(c[i] * w) * f32(2.0))
        a = f32(f32(v[i] * k) + t) if sign > 0 else f32(f32(v[i] * k) - t)
        out[i] = f32(a + f32(f32(qv[i] * d) * f32(2.0)))
    return out


def _x_inv(t):  # transform_inverse: (-(q^-1 p), q^-1)
    qi = _v(-t[3], -t[4], -t[5], t[6])
    return np.concatenate([-_rot(qi, t[:3]), qi]).astype(np.float32)


def _x_point(t, p):  # p + rotate(q, x)
    return (t[:3] + _rot(t[3:], p)).astype(np.float32)


IDENT = _v(0, 0, 0, 0, 0, 0, 1)


def write_rows(rows, body_q, shape_body, shape_gap):
    """-> dict of flat contact arrays, one row per input row; rejected rows are (-1, -1) with zeros; `accepted` mask."""
    n = len(rows["key"])
    out = dict(shape0=np.full(n, -1, np.int32), shape1=np.full(n, -1, np.int32), accepted=np.zeros(n, bool),
               margin0=np.zeros(n, np.float32), margin1=np.zeros(n, np.float32))
    for k in ("point0", "point1", "offset0", "offset1", "normal"):
        out[k] = np.zeros((n, 3), np.float32)
    for i in range(n):
        sa, sb = int(rows["shape_a"][i]), int(rows["shape_b"][i])
        ma, mb, dist = f32(rows["margin_a"][i]), f32(rows["margin_b"][i]), f32(rows["distance"][i])
        # effective radii (compute_effective_radius): zero for SDF / mesh shapes; the triangle leg's partner can be a sphere / capsule
        ra = f32(rows["radius_a"][i]) if "radius_a" in rows else f32(0.0)
        rb = f32(rows["radius_b"][i]) if "radius_b" in rows else f32(0.0)
        total = f32(f32(f32(ra + rb) + ma) + mb)
        nr = rows["normal"][i].astype(np.float32)
        ln = np.sqrt(_dot(nr, nr), dtype=np.float32)
        nab = (nr / ln).astype(np.float32) if ln > 0 else np.zeros(3, np.float32)
        c = rows["center"][i].astype(np.float32)
        aw = (c - nab * f32(f32(f32(0.5) * dist) + ra)).astype(np.float32)
        bw = (c + nab * f32(f32(f32(0.5) * dist) + rb)).astype(np.float32)
        sep = f32(_dot((bw - aw).astype(np.float32), nab) - total)
        if sep > f32(f32(shape_gap[sa]) + f32(shape_gap[sb])):
            continue
        ba, bb = int(shape_body[sa]), int(shape_body[sb])
        Xa = IDENT if ba < 0 else _x_inv(body_q[ba].astype(np.float32))
        Xb = IDENT if bb < 0 else _x_inv(body_q[bb].astype(np.float32))
        m0, m1 = f32(ra + ma), f32(rb + mb)
        out["accepted"][i] = True
        out["shape0"][i], out["shape1"][i] = sa, sb
        out["point0"][i], out["point1"][i] = _x_point(Xa, aw), _x_point(Xb, bw)
        out["offset0"][i] = _rot(Xa[3:], (nab * m0).astype(np.float32))
        out["offset1"][i] = _rot(Xb[3:], (nab * f32(-m1)).astype(np.float32))
        out["normal"][i] = nab
        out["margin0"][i], out["margin1"][i] = m0, m1
    return out


def eval_body_contact(ct, body_q, body_qd, body_com, mat, shape_body, friction_smoothing, props=None, order=None):
    """-> body_f [B, 6]; contributions are summed in row order (`order` overrides it)."""
    B = len(body_q)
    body_f = np.zeros((B, 6), np.float32)
    rows = range(len(ct["shape0"])) if order is None else order
    for i in rows:
        sa, sb = int(ct["shape0"][i]), int(ct["shape1"][i])
        if sa == sb:
            continue
        ke = kd = kf = ka = mu = f32(0.0)
        nz, ba, bb = 0, -1, -1
        for s in (sa, sb):
            if s >= 0:
                nz += 1
                ke, kd, kf, ka, mu = (f32(ke + mat["ke"][s]), f32(kd + mat["kd"][s]), f32(kf + mat["kf"][s]),
                                      f32(ka + mat["ka"][s]), f32(mu + mat["mu"][s]))
        if sa >= 0:
            ba = int(shape_body[sa])
        if sb >= 0:
            bb = int(shape_body[sb])
        if nz > 0:
            ke, kd, kf, ka, mu = (f32(x / f32(nz)) for x in (ke, kd, kf, ka, mu))
        if props is not None:
            cke, ckd, cmu = f32(props["stiffness"][i]), f32(props["damping"][i]), f32(props["friction"][i])
            ke = cke if cke > 0 else ke
            kd = ckd if ckd > 0 else kd
            mu = f32(mu * cmu) if cmu > 0 else mu
        n = (-ct["normal"][i]).astype(np.float32)
        bx_a, bx_b = ct["point0"][i].astype(np.float32), ct["point1"][i].astype(np.float32)
        r_a = r_b = np.zeros(3, np.float32)
        if ba >= 0:
            X = body_q[ba].astype(np.float32)
            bx_a = (_x_point(X, bx_a) - n * f32(ct["margin0"][i])).astype(np.float32)
            r_a = (bx_a - _x_point(X, body_com[ba].astype(np.float32))).astype(np.float32)
        if bb >= 0:
            X = body_q[bb].astype(np.float32)
            bx_b = (_x_point(X, bx_b) + n * f32(ct["margin1"][i])).astype(np.float32)
            r_b = (bx_b - _x_point(X, body_com[bb].astype(np.float32))).astype(np.float32)
        d = _dot(n, (bx_a - bx_b).astype(np.float32))
        if d >= ka:
            continue
        bv_a = bv_b = np.zeros(3, np.float32)
        if ba >= 0:
            bv_a = (body_qd[ba, :3] + _cross(body_qd[ba, 3:].astype(np.float32), r_a)).astype(np.float32)
        if bb >= 0:
            bv_b = (body_qd[bb, :3] + _cross(body_qd[bb, 3:].astype(np.float32), r_b)).astype(np.float32)
        v = (bv_a - bv_b).astype(np.float32)
        vn = _dot(n, v)
        vt = (v - n * vn).astype(np.float32)
        fn = f32(d * ke)
        fd = f32(f32(min(vn, f32(0.0)) * kd) * (f32(1.0) if d < 0 else f32(0.0)))
        ft = np.zeros(3, np.float32)
        if d < 0:
            delta = f32(friction_smoothing)
            a2 = _dot(vt, vt)
            vs = f32(f32(0.5) * a2) if a2 <= f32(delta * delta) else f32(delta * f32(np.sqrt(a2, dtype=np.float32) - f32(f32(0.5) * delta)))
            if vs > 0:
                fr = (vt / vs).astype(np.float32)
                ft = (fr * min(f32(kf * vs), f32(f32(-mu) * f32(fn + fd)))).astype(np.float32)
        f_total = (n * f32(fn + fd) + ft).astype(np.float32)
        if ba >= 0:
            body_f[ba, :3] = (body_f[ba, :3] - f_total).astype(np.float32)
            body_f[ba, 3:] = (body_f[ba, 3:] - _cross(r_a, f_total)).astype(np.float32)
        if bb >= 0:
            body_f[bb, :3] = (body_f[bb, :3] + f_total).astype(np.float32)
            body_f[bb, 3:] = (body_f[bb, 3:] + _cross(r_b, f_total)).astype(np.float32)
    return body_f
