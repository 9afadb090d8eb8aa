"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's frame-to-frame contact matching on Newton's flat, key-sorted
contact arrays (newton/_src/geometry/contact_match.py): _match_contacts_kernel :266-354 (pair range by binary search on the
sort key with the sub-key bits masked, closest midpoint within pos_threshold whose normal passes the dot threshold, packed
atomic_min claim = (distance, low 32 key bits)) and _resolve_claims_kernel :357-391 (losers become MATCH_BROKEN).  float32."""
from __future__ import annotations

import numpy as np

f32 = np.float32
MATCH_NOT_FOUND, MATCH_BROKEN = -1, -2


def sort_key(shape_a, shape_b, sub_key):
    """make_contact_sort_key (contact_data.py:60-90): [62:43] shape_a, [42:23] shape_b, [22:0] sub key."""
    return (int(shape_a) << 43) | (int(shape_b) << 23) | (int(sub_key) & 0x7FFFFF)


def _q_rot(q, v):
    qv, w = q[:3].astype(f32), f32(q[3])
    return (v * f32(f32(f32(2.0) * w) * w - f32(1.0)) + np.cross(qv, v).astype(f32) * w * f32(2.0) + qv * f32(np.dot(qv, v)) * f32(2.0)).astype(f32)


def midpoints(body_q, shape_body, shape0, shape1, point0, point1):
    out = np.zeros((len(shape0), 3), dtype=f32)
    for i in range(len(shape0)):
        p = []
        for s, pt in ((shape0[i], point0[i]), (shape1[i], point1[i])):
            b = shape_body[s]
            pt = np.asarray(pt, dtype=f32)
            p.append(pt if b < 0 else (body_q[b, :3].astype(f32) + _q_rot(body_q[b, 3:], pt)).astype(f32))
        out[i] = f32(0.5) * (p[0] + p[1])
    return out


def match(new_keys, new_pos, new_normal, prev_keys, prev_pos, prev_normal, pos_threshold=0.0005, normal_dot_threshold=0.995):
    n_new, n_old = len(new_keys), len(prev_keys)
    out = np.full(n_new, MATCH_NOT_FOUND, dtype=np.int32)
    if n_old == 0:
        return out
    claim = {}
    best_d = np.zeros(n_new, dtype=f32)
    pk = np.asarray(prev_keys, dtype=np.int64)
    for t in range(n_new):
        prefix = int(new_keys[t]) & ~0x7FFFFF
        lo = int(np.searchsorted(pk, prefix, side="left"))
        hi = int(np.searchsorted(pk, prefix + 0x800000, side="left"))
        if lo >= hi:
            continue
        best, bd = -1, f32(pos_threshold * pos_threshold)
        for j in range(lo, hi):
            d = (new_pos[t] - prev_pos[j]).astype(f32)
            ds = f32(np.dot(d, d))
            if ds <= bd and f32(np.dot(new_normal[t], prev_normal[j])) >= f32(normal_dot_threshold):
                bd, best = ds, j
        if best < 0:
            out[t] = MATCH_BROKEN
            continue
        out[t], best_d[t] = best, bd
        c = (float(bd), int(new_keys[t]) & 0xFFFFFFFF)
        if best not in claim or c < claim[best]:
            claim[best] = c
    for t in range(n_new):
        if out[t] >= 0 and claim[int(out[t])][1] != (int(new_keys[t]) & 0xFFFFFFFF):
            out[t] = MATCH_BROKEN
    return out
