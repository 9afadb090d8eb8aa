// nt_primitives.hpp -- analytic primitive contact generation for the gfx950 collide kernel.
// Behaviour contract (what, not how): newton/_src/geometry/collision_primitive.py:48-683,1176-1232
// and the dispatch order of narrow_phase_primitive_kernel (newton/_src/geometry/narrow_phase.py:642-865).
// Normal points from shape A into shape B; position is the midpoint between the surfaces; distance < 0 is
// penetration; unused contact slots carry NT_MAXVAL.
#pragma once
#include "nt_math.hpp"

namespace nt {

constexpr float NT_MAXVAL = 1e10f;  // newton/_src/core/types.py:71-72
constexpr float NT_MINVAL = 1e-15f;

enum GeoType : int {
    GEO_NONE = 0, GEO_PLANE = 1, GEO_HFIELD = 2, GEO_SPHERE = 3, GEO_CAPSULE = 4, GEO_ELLIPSOID = 5,
    GEO_CYLINDER = 6, GEO_BOX = 7, GEO_MESH = 8, GEO_CONE = 9, GEO_CONVEX_MESH = 10
};

struct Contacts4 {
    float d0, d1, d2, d3;
    vec3 p0, p1, p2, p3;
    vec3 normal;
    NT_DI Contacts4() : d0(NT_MAXVAL), d1(NT_MAXVAL), d2(NT_MAXVAL), d3(NT_MAXVAL) {}
    // (every field assigned through a value select: an if-chain of stores is merged by the compiler into ONE store through a
    // selected address, which again pins the struct in scratch memory)
    NT_DI void set(int i, float d, vec3 p) {
        d0 = i == 0 ? d : d0; p0 = vsel(i == 0, p, p0);
        d1 = i == 1 ? d : d1; p1 = vsel(i == 1, p, p1);
        d2 = i == 2 ? d : d2; p2 = vsel(i == 2, p, p2);
        d3 = i > 2 || i < 0 ? d : d3; p3 = vsel(i > 2 || i < 0, p, p3);
    }
    NT_DI float dist(int i) const { return fsel(i == 0, d0, fsel(i == 1, d1, fsel(i == 2, d2, d3))); }
    NT_DI vec3 pos(int i) const { return vsel(i == 0, p0, vsel(i == 1, p1, vsel(i == 2, p2, p3))); }
    // slots 1..3 = the kept ones of three candidates, in order.  Same result as `if (keep) set(n++, ...)`, but written with
    // selects: a run-time slot index makes the compiler keep the whole struct in scratch memory (measured: 166 scratch
    // instructions in the fused rollout, the pair phase waits on every one of them)
    NT_DI void append3(bool k0, float e0, vec3 q0, bool k1, float e1, vec3 q1, bool k2, float e2, vec3 q2) {
        const vec3 z;
        const bool two = (k0 && k1) || ((k0 != k1) && k2);
        d1 = k0 ? e0 : (k1 ? e1 : (k2 ? e2 : NT_MAXVAL));
        p1 = vsel(k0, q0, vsel(k1, q1, vsel(k2, q2, z)));
        d2 = two ? ((k0 && k1) ? e1 : e2) : NT_MAXVAL;
        p2 = vsel(two, vsel(k0 && k1, q1, q2), z);
        d3 = (k0 && k1 && k2) ? e2 : NT_MAXVAL;
        p3 = vsel(k0 && k1 && k2, q2, z);
    }
};

NT_DI vec3 closest_segment_point(vec3 a, vec3 b, vec3 pt) {
    vec3 ab = b - a;
    float t = dot(pt - a, ab) / (dot(ab, ab) + 1e-6f);
    return a + clampf(t, 0.0f, 1.0f) * ab;
}

NT_DI void plane_sphere(vec3 n, vec3 plane_pos, vec3 c, float r, float& dist, vec3& pos) {
    dist = dot(c - plane_pos, n) - r;
    pos = c - n * (r + 0.5f * dist);
}

NT_DI void sphere_sphere(vec3 p1, float r1, vec3 p2, float r2, float& dist, vec3& pos, vec3& n) {
    vec3 dir = p2 - p1;
    dist = length(dir);
    if (dist == 0.0f) n = vec3(1.0f, 0.0f, 0.0f);
    else n = dir / dist;
    dist = dist - (r1 + r2);
    pos = p1 + n * (r1 + 0.5f * dist);
}

NT_DI void capsule_capsule(vec3 c1, vec3 ax1, float r1, float h1, vec3 c2, vec3 ax2, float r2, float h2, Contacts4& out) {
    vec3 axis1 = ax1 * h1;
    vec3 axis2 = ax2 * h2;
    vec3 dif = c1 - c2;
    float ma = dot(axis1, axis1);
    float mb = -dot(axis1, axis2);
    float mc = dot(axis2, axis2);
    float u = -dot(axis1, dif);
    float v = dot(axis2, dif);
    float det = ma * mc - mb * mb;
    if (fabsf(det) >= NT_MINVAL) {
        float inv_det = 1.0f / det;
        float x1 = (mc * u - mb * v) * inv_det;
        float x2 = (ma * v - mb * u) * inv_det;
        if (x1 > 1.0f) { x1 = 1.0f; x2 = (v - mb) / mc; }
        else if (x1 < -1.0f) { x1 = -1.0f; x2 = (v + mb) / mc; }
        if (x2 > 1.0f) { x2 = 1.0f; x1 = clampf((u - mb) / ma, -1.0f, 1.0f); }
        else if (x2 < -1.0f) { x2 = -1.0f; x1 = clampf((u + mb) / ma, -1.0f, 1.0f); }
        vec3 v1 = c1 + axis1 * x1;
        vec3 v2 = c2 + axis2 * x2;
        sphere_sphere(v1, r1, v2, r2, out.d0, out.p0, out.normal);
    } else {
        vec3 v1 = c1 + axis1;
        float x2 = clampf((v - mb) / mc, -1.0f, 1.0f);
        vec3 v2 = c2 + axis2 * x2;
        sphere_sphere(v1, r1, v2, r2, out.d0, out.p0, out.normal);
        v1 = c1 - axis1;
        x2 = clampf((v + mb) / mc, -1.0f, 1.0f);
        v2 = c2 + axis2 * x2;
        vec3 n_unused;
        sphere_sphere(v1, r1, v2, r2, out.d1, out.p1, n_unused);
    }
}

NT_DI void plane_ellipsoid(vec3 n, vec3 plane_pos, vec3 c, const mat33& rot, vec3 size, Contacts4& out) {
    vec3 s = -normalize(cw_mul(transpose(rot) * n, size));
    vec3 pos = c + rot * cw_mul(s, size);
    float dist = dot(n, pos - plane_pos);
    pos = pos - n * dist * 0.5f;
    out.d0 = dist;
    out.p0 = pos;
    out.normal = n;
}

// keeps the (up to) 4 deepest corners, tracking the worst kept contact by index
NT_DI void plane_box(vec3 n, vec3 plane_pos, vec3 c, const mat33& rot, vec3 size, float margin, Contacts4& out) {
    float center_dist = dot(c - plane_pos, n);
    // the kept set lives directly in `out` (select-style set / dist keep it in registers; a dynamically indexed
    // private array would go to scratch memory)
    out = Contacts4();
    int ncontact = 0, worst_idx = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        vec3 corner((i & 1) ? size.x : -size.x, (i & 2) ? size.y : -size.y, (i & 4) ? size.z : -size.z);
        corner = rot * corner;
        float ldist = dot(n, corner);
        float cdist = center_dist + ldist;
        if (cdist > margin) continue;
        vec3 cpos = corner + c - 0.5f * n * cdist;
        if (ncontact < 4) {
            out.set(ncontact, cdist, cpos);
            if (ncontact == 0 || cdist > out.dist(worst_idx)) worst_idx = ncontact;
            ncontact += 1;
        } else if (cdist < out.dist(worst_idx)) {
            out.set(worst_idx, cdist, cpos);
            worst_idx = 0;
            if (out.d1 > out.dist(worst_idx)) worst_idx = 1;
            if (out.d2 > out.dist(worst_idx)) worst_idx = 2;
            if (out.d3 > out.dist(worst_idx)) worst_idx = 3;
        }
    }
    out.normal = n;
}

NT_DI void sphere_cylinder(vec3 sp, float sr, vec3 cp, vec3 axis, float cr, float ch, float& dist, vec3& pos, vec3& n) {
    vec3 vec = sp - cp;
    float x = dot(vec, axis);
    vec3 a_proj = axis * x;
    vec3 p_proj = vec - a_proj;
    float p_proj_sqr = dot(p_proj, p_proj);
    bool collide_side = fabsf(x) < ch;
    bool collide_cap = p_proj_sqr < (cr * cr);
    if (collide_side && collide_cap) {
        float dist_cap = ch - fabsf(x);
        float dist_radius = cr - sqrtf(p_proj_sqr);
        if (dist_cap < dist_radius) collide_side = false;
        else collide_cap = false;
    }
    if (collide_side) {
        sphere_sphere(sp, sr, cp + a_proj, cr, dist, pos, n);
    } else if (collide_cap) {
        vec3 pos_cap, pn;
        if (x > 0.0f) { pos_cap = cp + axis * ch; pn = axis; }
        else { pos_cap = cp - axis * ch; pn = -axis; }
        plane_sphere(pn, pos_cap, sp, sr, dist, pos);
        n = -pn;
    } else {
        float l = sqrtf(p_proj_sqr);
        float inv_len = 1.0f / (l != 0.0f ? l : 1e-15f);
        p_proj = p_proj * (cr * inv_len);
        vec3 cap_offset = axis * (signf(x) * ch);
        sphere_sphere(sp, sr, cp + cap_offset + p_proj, 0.0f, dist, pos, n);
    }
}

// near-upright: fixed tripod on the near cap + deepest rim point; otherwise rolling mode
NT_DI void plane_cylinder(vec3 n, vec3 plane_pos, vec3 cp, vec3 cyl_axis, float cr, float ch, Contacts4& out) {
    vec3 axis = cyl_axis;
    float dot_na = dot(n, axis);
    if (dot_na > 0.0f) { axis = -axis; dot_na = -dot_na; }
    vec3 cap_center = cp + axis * ch;
    vec3 perp_align = -n + axis * dot_na;
    float pl2 = dot(perp_align, perp_align);
    bool has_align = pl2 > 1e-10f;
    if (has_align) perp_align = perp_align * (1.0f / sqrtf(pl2));
    float abs_dot = -dot_na;
    bool flat_mode = abs_dot >= 0.9238795325112867f;  // cos(22.5 deg)
    vec3 perp_fixed;
    if (flat_mode || !has_align) {
        vec3 ref(1.0f, 0.0f, 0.0f);
        if (fabsf(dot(axis, ref)) > 0.9f) ref = vec3(0.0f, 1.0f, 0.0f);
        perp_fixed = ref - axis * dot(axis, ref);
        perp_fixed = normalize(perp_fixed);
    }
    vec3 deepest_perp = vsel(has_align, perp_align, perp_fixed);
    vec3 deepest_pt = cap_center + deepest_perp * cr;
    float deepest_d = dot(deepest_pt - plane_pos, n);
    vec3 deepest_pos = deepest_pt - n * (deepest_d * 0.5f);
    out.d0 = deepest_d;
    out.p0 = deepest_pos;
    float mt = 0.01f * fmaxw(cr, ch);
    float mt2 = mt * mt;
    if (flat_mode) {
        vec3 u_fixed = perp_fixed * cr;
        vec3 v_fixed = cross(axis, perp_fixed) * cr;
        const float c120 = -0.5f, s120 = 0.8660254f;
        vec3 pt0 = cap_center + u_fixed;
        float d0 = dot(pt0 - plane_pos, n);
        vec3 pos0 = pt0 - n * (d0 * 0.5f);
        vec3 pt1 = cap_center + c120 * u_fixed + s120 * v_fixed;
        float d1 = dot(pt1 - plane_pos, n);
        vec3 pos1 = pt1 - n * (d1 * 0.5f);
        vec3 pt2 = cap_center + c120 * u_fixed - s120 * v_fixed;
        float d2 = dot(pt2 - plane_pos, n);
        vec3 pos2 = pt2 - n * (d2 * 0.5f);
        out.append3(length_sq(pos0 - deepest_pos) > mt2, d0, pos0, length_sq(pos1 - deepest_pos) > mt2, d1, pos1,
                    length_sq(pos2 - deepest_pos) > mt2, d2, pos2);
    } else {
        vec3 perp_roll = vsel(has_align, perp_align, perp_fixed);
        vec3 u = perp_roll * cr;
        vec3 v = cross(axis, perp_roll) * cr;
        vec3 pt = cp - axis * ch + u;
        float d = dot(pt - plane_pos, n);
        vec3 pos = pt - n * (d * 0.5f);
        vec3 pt_pos_v = cap_center + v;
        float d_pos_v = dot(pt_pos_v - plane_pos, n);
        vec3 pt_neg_v = cap_center - v;
        float d_neg_v = dot(pt_neg_v - plane_pos, n);
        bool use_pos_v = d_pos_v <= d_neg_v;
        vec3 ptv = vsel(use_pos_v, pt_pos_v, pt_neg_v);
        float dv = use_pos_v ? d_pos_v : d_neg_v;
        vec3 posv = ptv - n * (dv * 0.5f);
        out.append3(length_sq(pos - deepest_pos) > mt2, d, pos, length_sq(posv - deepest_pos) > mt2, dv, posv, false, 0.0f, vec3());
    }
    out.normal = n;
}

NT_DI void sphere_box(vec3 sp, float sr, vec3 bp, const mat33& rot, vec3 size, float& cdist, vec3& cpos, vec3& cn) {
    vec3 center = transpose(rot) * (sp - bp);
    vec3 clamped = vmax(-size, vmin(size, center));
    vec3 diff = clamped - center;
    float dist = length(diff);
    vec3 dir = dist == 0.0f ? diff : diff / dist;
    vec3 pos;
    if (dist <= 1e-6f) {
        float closest = 2.0f * (size.x + size.y + size.z);
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float fd = fabsf(((i % 2) ? 1.0f : -1.0f) * vget(size, i / 2) - vget(center, i / 2));
            if (closest > fd) { closest = fd; k = i; }
        }
        vec3 nearest(0.0f);
        vset(nearest, k / 2, (k % 2) ? -1.0f : 1.0f);
        pos = center + nearest * (sr - closest) / 2.0f;
        cn = rot * nearest;
        cdist = -closest - sr;
    } else {
        vec3 deepest = center + dir * sr;
        pos = 0.5f * (clamped + deepest);
        cn = rot * dir;
        cdist = dist - sr;
    }
    cpos = bp + rot * pos;
}

// Dispatch for a type-sorted pair (type_a <= type_b).  Returns true if the pair has an analytic path
// (even if it produced no contact); false => the pair belongs to the convex (MPR/GJK) path.
NT_DI bool primitive_pair(int type_a, int type_b, const xform& Xa, const xform& Xb, vec3 sa, vec3 sb, float box_margin,
                          Contacts4& out) {
    bool plane_a = type_a == GEO_PLANE;
    bool sphere_a = type_a == GEO_SPHERE, sphere_b = type_b == GEO_SPHERE;
    bool capsule_a = type_a == GEO_CAPSULE, capsule_b = type_b == GEO_CAPSULE;
    bool ellipsoid_b = type_b == GEO_ELLIPSOID, cylinder_b = type_b == GEO_CYLINDER, box_b = type_b == GEO_BOX;
    const vec3 ez(0.0f, 0.0f, 1.0f);

    bool use_plane_cylinder = plane_a && cylinder_b;
    if (use_plane_cylinder && sb.z > 0.0f) {
        vec3 pn = quat_rotate_ez(Xa.q);
        vec3 ca = quat_rotate_ez(Xb.q);
        use_plane_cylinder = fabsf(dot(pn, ca)) * sb.z >= sb.y;
    }
    if (plane_a && sphere_b) {
        vec3 pn = quat_rotate_ez(Xa.q);
        plane_sphere(pn, Xa.p, Xb.p, sb.x, out.d0, out.p0);
        out.normal = pn;
    } else if (plane_a && ellipsoid_b) {
        plane_ellipsoid(quat_rotate_ez(Xa.q), Xa.p, Xb.p, quat_to_matrix(Xb.q), sb, out);
    } else if (plane_a && box_b) {
        plane_box(quat_rotate_ez(Xa.q), Xa.p, Xb.p, quat_to_matrix(Xb.q), sb, box_margin, out);
    } else if (sphere_a && sphere_b) {
        sphere_sphere(Xa.p, sa.x, Xb.p, sb.x, out.d0, out.p0, out.normal);
    } else if (plane_a && capsule_b) {
        vec3 pn = quat_rotate_ez(Xa.q);
        vec3 seg = quat_rotate_ez(Xb.q) * sb.y;
        plane_sphere(pn, Xa.p, Xb.p + seg, sb.x, out.d0, out.p0);
        plane_sphere(pn, Xa.p, Xb.p - seg, sb.x, out.d1, out.p1);
        out.normal = pn;
    } else if (use_plane_cylinder) {
        plane_cylinder(quat_rotate_ez(Xa.q), Xa.p, Xb.p, quat_rotate_ez(Xb.q), sb.x, sb.y, out);
    } else if (sphere_a && capsule_b) {
        vec3 seg = quat_rotate_ez(Xb.q) * sb.y;
        vec3 pt = closest_segment_point(Xb.p - seg, Xb.p + seg, Xa.p);
        sphere_sphere(Xa.p, sa.x, pt, sb.x, out.d0, out.p0, out.normal);
    } else if (capsule_a && capsule_b) {
        capsule_capsule(Xa.p, quat_rotate_ez(Xa.q), sa.x, sa.y, Xb.p, quat_rotate_ez(Xb.q), sb.x, sb.y, out);
    } else if (sphere_a && cylinder_b && sb.z == 0.0f) {
        sphere_cylinder(Xa.p, sa.x, Xb.p, quat_rotate_ez(Xb.q), sb.x, sb.y, out.d0, out.p0, out.normal);
    } else if (sphere_a && box_b) {
        sphere_box(Xa.p, sa.x, Xb.p, quat_to_matrix(Xb.q), sb, out.d0, out.p0, out.normal);
    }
    return (plane_a && (sphere_b || capsule_b || ellipsoid_b || use_plane_cylinder || box_b)) ||
           (sphere_a && (sphere_b || capsule_b || (cylinder_b && sb.z == 0.0f) || box_b)) || (capsule_a && capsule_b);
}

}  // namespace nt
