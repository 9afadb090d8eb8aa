// nt_collide.hpp -- collide phases: compute_shape_aabbs, per-env broad phase test, narrow phase routing (analytic primitives /
// MPR-GJK manifold), contact writer.
// Included by nt_kernels.hip inside its anonymous namespace, in this order: nt_layout.hpp, nt_collide.hpp, nt_xpbd.hpp,
// nt_semi_implicit.hpp, nt_featherstone.hpp (one translation unit; the split is for reading, not for separate compilation).
#pragma once

// ------------------------------------------------------------------------------------------------
// collide: compute_shape_aabbs (collide.py:283-472)
// ------------------------------------------------------------------------------------------------
NT_DI void shape_aabb(int geo_type, const xform& X, vec3 scale, float effective_gap, const float* mesh_bounds, vec3& lo,
                      vec3& hi) {
    vec3 pos = X.p;
    quat q = X.q;
    vec3 mv(effective_gap, effective_gap, effective_gap);
    bool infinite_plane = (geo_type == GEO_PLANE) && (scale.x == 0.0f && scale.y == 0.0f);
    if (infinite_plane) {
        vec3 normal = quat_rotate_ez(q);
        const float H = 1.0e6f;
        vec3 he(H, H, H);
        lo = pos - he - mv;
        hi = pos + he + mv;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float n_i = vget(normal, i);
            if (fabsf(n_i) > 0.5f) {
                float lateral = fabsf(vget(normal, (i + 1) % 3)) + fabsf(vget(normal, (i + 2) % 3));
                float rise = lateral * H / fabsf(n_i);
                if (n_i > 0.0f) vset(hi, i, fminw(vget(hi, i), vget(pos, i) + rise + effective_gap));
                else vset(lo, i, fmaxw(vget(lo, i), vget(pos, i) - rise - effective_gap));
            }
        }
        return;
    }
    vec3 he;
    if (geo_type == GEO_SPHERE) {
        he = vec3(scale.x, scale.x, scale.x);
    } else if (geo_type == GEO_BOX) {
        vec3 r0 = quat_rotate_ex(q);
        vec3 r1 = quat_rotate_ey(q);
        vec3 r2 = quat_rotate_ez(q);
        he = vec3(fabsf(r0.x) * scale.x + fabsf(r1.x) * scale.y + fabsf(r2.x) * scale.z,
                  fabsf(r0.y) * scale.x + fabsf(r1.y) * scale.y + fabsf(r2.y) * scale.z,
                  fabsf(r0.z) * scale.x + fabsf(r1.z) * scale.y + fabsf(r2.z) * scale.z);
    } else if (geo_type == GEO_CAPSULE) {
        vec3 axis = quat_rotate_ez(q);
        he = vec3(scale.x, scale.x, scale.x) + vabs(axis) * scale.y;
    } else if (geo_type == GEO_CYLINDER) {
        float radius = scale.x, hh = scale.y, barrel = scale.z;
        if (barrel >= hh && barrel > 0.0f) radius += (hh * hh) / (barrel + sqrtf(barrel * barrel - hh * hh));
        vec3 r0 = quat_rotate_ex(q);
        vec3 r1 = quat_rotate_ey(q);
        vec3 r2 = quat_rotate_ez(q);
        he = vec3(radius * sqrtf(r0.x * r0.x + r1.x * r1.x) + hh * fabsf(r2.x),
                  radius * sqrtf(r0.y * r0.y + r1.y * r1.y) + hh * fabsf(r2.y),
                  radius * sqrtf(r0.z * r0.z + r1.z * r1.z) + hh * fabsf(r2.z));
    } else if (geo_type == GEO_CONVEX_MESH) {
        // pre-computed local AABB (scale baked in) rotated to the world frame (collide.py:421-445)
        vec3 a = cw_mul(vec3(mesh_bounds[0], mesh_bounds[1], mesh_bounds[2]), scale);
        vec3 b = cw_mul(vec3(mesh_bounds[3], mesh_bounds[4], mesh_bounds[5]), scale);
        vec3 local_lo = vmin(a, b), local_hi = vmax(a, b);
        vec3 center = (local_lo + local_hi) * 0.5f;
        vec3 half = (local_hi - local_lo) * 0.5f;
        vec3 world_center = quat_rotate(q, center) + pos;
        vec3 r0 = quat_rotate_ex(q);
        vec3 r1 = quat_rotate_ey(q);
        vec3 r2 = quat_rotate_ez(q);
        vec3 world_half(fabsf(r0.x) * half.x + fabsf(r1.x) * half.y + fabsf(r2.x) * half.z,
                        fabsf(r0.y) * half.x + fabsf(r1.y) * half.y + fabsf(r2.y) * half.z,
                        fabsf(r0.z) * half.x + fabsf(r1.z) * half.y + fabsf(r2.z) * half.z);
        lo = world_center - world_half - mv;
        hi = world_center + world_half + mv;
        return;
    } else if (geo_type == GEO_PLANE) {
        // finite plane: compute_tight_aabb_from_support on the rectangle with half extents scale / 2 (collide.py:448-465);
        // its own branch so that the generic support map is not inlined six more times into every kernel
        mat33 Rt = transpose(quat_to_matrix(q));
        vec3 local_x(Rt.m00, Rt.m10, Rt.m20), local_y(Rt.m01, Rt.m11, Rt.m21), local_z(Rt.m02, Rt.m12, Rt.m22);
        vec3 half(scale.x * 0.5f, scale.y * 0.5f, 0.0f);
        float max_x = dot(local_x, support_map_plane(half, local_x));
        float max_y = dot(local_y, support_map_plane(half, local_y));
        float max_z = dot(local_z, support_map_plane(half, local_z));
        float min_x = dot(local_x, support_map_plane(half, -local_x));
        float min_y = dot(local_y, support_map_plane(half, -local_y));
        float min_z = dot(local_z, support_map_plane(half, -local_z));
        lo = vec3(min_x, min_y, min_z) + pos - mv;
        hi = vec3(max_x, max_y, max_z) + pos + mv;
        return;
    } else if (geo_type == GEO_ELLIPSOID || geo_type == GEO_CONE) {
        // compute_tight_aabb_from_support (collision_core.py:454-547): six support evaluations in local space
        mat33 Rt = transpose(quat_to_matrix(q));
        vec3 local_x(Rt.m00, Rt.m10, Rt.m20), local_y(Rt.m01, Rt.m11, Rt.m21), local_z(Rt.m02, Rt.m12, Rt.m22);
        Geom g;
        g.type = geo_type;
        g.scale = scale;
        float max_x = dot(local_x, support_map(g, local_x));
        float max_y = dot(local_y, support_map(g, local_y));
        float max_z = dot(local_z, support_map(g, local_z));
        float min_x = dot(local_x, support_map(g, -local_x));
        float min_y = dot(local_y, support_map(g, -local_y));
        float min_z = dot(local_z, support_map(g, -local_z));
        lo = vec3(min_x, min_y, min_z) + pos - mv;
        hi = vec3(max_x, max_y, max_z) + pos + mv;
        return;
    }
    lo = pos - he - mv;
    hi = pos + he + mv;
}

template <int EPB>
NT_DI void shape_item(const Ctx<EPB>& c, const int s);
NT_DI void store_shape_world(const nt_contacts& ct, int gid, const xform& X, vec3 lo, vec3 hi);
template <int EPB>
NT_DI void store_global_shapes_world(const Ctx<EPB>& c);
template <int EPB>
NT_DI void phase_shapes(const Ctx<EPB>& c) {
    store_global_shapes_world(c);
    if (!c.valid) return;
    const bool out = c.a.ct.world_xform != nullptr;  // only the stand-alone collide launch exports (never the fused rollouts)
    for (int s = c.slot; s < c.a.m.ns; s += c.nslot) {
        shape_item(c, s);
        if (out)
            store_shape_world(c.a.ct, c.newton_shape_id(s), c.lxf(c.L.sx, 0, c.a.m.ns, s), c.lv3(c.L.sa, 0, c.a.m.ns, s),
                              c.lv3(c.L.sa, 3, c.a.m.ns, s));
    }
}
template <int EPB>
NT_DI void shape_item(const Ctx<EPB>& c, const int s) {
    const nt_model& m = c.a.m;
    {
        int body = c.T.shape_body[s];
        xform X = c.shape_local_xform(s);
        if (body >= 0) X = c.body_q(body) * X;
        vec3 lo, hi;
        shape_aabb(c.T.shape_type[s], X, c.shape_scale(s), c.shape_f(s, SP_MARGIN) + c.shape_f(s, SP_GAP),
                   m.shape_mesh_bounds + 6 * s, lo, hi);
        c.st_lxf(c.L.sx, m.ns, s, X);
        c.st_lv3(c.L.sa, 0, m.ns, s, lo);
        c.st_lv3(c.L.sa, 3, m.ns, s, hi);
    }
}
// geom_xform / aabb_lower / aabb_upper of compute_shape_aabbs for the stages outside the tiles (nt_contacts.world_*)
NT_DI void store_shape_world(const nt_contacts& ct, int gid, const xform& X, vec3 lo, vec3 hi) {
    float* x = ct.world_xform + 7 * (size_t)gid;
    x[0] = X.p.x; x[1] = X.p.y; x[2] = X.p.z; x[3] = X.q.x; x[4] = X.q.y; x[5] = X.q.z; x[6] = X.q.w;
    float* l = ct.world_aabb_lower + 3 * (size_t)gid;
    float* u = ct.world_aabb_upper + 3 * (size_t)gid;
    l[0] = lo.x; l[1] = lo.y; l[2] = lo.z;
    u[0] = hi.x; u[1] = hi.y; u[2] = hi.z;
}
// the static global (world -1) shapes: written once per launch by the first workgroup
template <int EPB>
NT_DI void store_global_shapes_world(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    if (!c.a.ct.world_xform || blockIdx.x != 0) return;
    for (int s = m.ns + (int)threadIdx.x; s < m.ns + m.ng; s += blockDim.x) {
        xform X = c.shape_local_xform(s);
        vec3 lo, hi;
        shape_aabb(c.T.shape_type[s], X, c.shape_scale(s), c.shape_f(s, SP_MARGIN) + c.shape_f(s, SP_GAP),
                   m.shape_mesh_bounds + 6 * s, lo, hi);
        store_shape_world(c.a.ct, c.newton_shape_id(s), X, lo, hi);
    }
}

// The global (world -1) shapes are static: their world transform and gap-widened AABB (an infinite plane's costs three IEEE
// divisions) are computed once per launch by the workgroup's last threads into the block-shared T.gworld, one barrier before the
// first pair phase reads them -- instead of by every pair lane in every substep.  Same function, same bits.
template <int EPB>
NT_DI void stage_global_world(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    // (workgroup-strided: a scene may carry more global shapes than the workgroup has lanes -- 64 in the narrow Featherstone tiles)
    for (int k = (int)blockDim.x - 1 - (int)threadIdx.x; k < m.ng; k += (int)blockDim.x) {
        const int s = m.ns + k;
        xform X = c.shape_local_xform(s);
        vec3 lo, hi;
        shape_aabb(c.T.shape_type[s], X, c.shape_scale(s), c.shape_f(s, SP_MARGIN) + c.shape_f(s, SP_GAP), m.shape_mesh_bounds + 6 * s, lo, hi);
        float* g = c.T.gworld + 13 * k;
        g[0] = X.p.x; g[1] = X.p.y; g[2] = X.p.z; g[3] = X.q.x; g[4] = X.q.y; g[5] = X.q.z; g[6] = X.q.w;
        g[7] = lo.x; g[8] = lo.y; g[9] = lo.z; g[10] = hi.x; g[11] = hi.y; g[12] = hi.z;
    }
}
template <int EPB>
NT_DI void shape_world(const Ctx<EPB>& c, int s, xform& X, vec3& lo, vec3& hi) {
    const nt_model& m = c.a.m;
    if (s < m.ns) {
        X = c.lxf(c.L.sx, 0, m.ns, s);
        lo = c.lv3(c.L.sa, 0, m.ns, s);
        hi = c.lv3(c.L.sa, 3, m.ns, s);
    } else if (c.gworld_ready) {
        const float* g = c.T.gworld + 13 * (s - m.ns);
        X = xform(vec3(g[0], g[1], g[2]), quat(g[3], g[4], g[5], g[6]));
        lo = vec3(g[7], g[8], g[9]);
        hi = vec3(g[10], g[11], g[12]);
    } else {
        X = c.shape_local_xform(s);  // global shapes are static (shape_body == -1)
        shape_aabb(c.T.shape_type[s], X, c.shape_scale(s), c.shape_f(s, SP_MARGIN) + c.shape_f(s, SP_GAP),
                   m.shape_mesh_bounds + 6 * s, lo, hi);
    }
}

// broad phase test (broad_phase_common.py:20-38, cutoff 0: AABBs are pre-expanded) + narrow phase primitive
// dispatch (narrow_phase.py:458-1014) + contact writer (collide.py:166-254).
// One lane per CONTACT SLOT (pair p = slot / cpp, sub-contact k = slot % cpp): the cpp lanes of a pair evaluate the
// same analytic pair redundantly (they are otherwise idle) and lane k writes the k-th admitted contact, so the
// world->body conversion and the 19 stores per contact run in parallel instead of 4-deep in one thread.
// Writes one contact record (world -> body frames, collide.py:166-204) into fixed slot `slot`.
template <int EPB>
NT_DI void write_contact_slot(const Ctx<EPB>& c, int slot, int sa, int sb, vec3 center, vec3 n, float dist, float ra, float rb,
                              float margin_a, float margin_b) {
    const nt_contacts& ct = c.a.ct;
    const int ncs = c.a.m.np * c.a.m.cpp;
    int ba = c.T.shape_body[sa], bb = c.T.shape_body[sb];
    xform Xbw_a = ba < 0 ? xform() : xform_inverse(c.body_q_in(ba));
    xform Xbw_b = bb < 0 ? xform() : xform_inverse(c.body_q_in(bb));
    float off_a = ra + margin_a, off_b = rb + margin_b;
    vec3 aw = center - n * (0.5f * dist + ra);
    vec3 bw = center + n * (0.5f * dist + rb);
    vec3 p0 = xform_point(Xbw_a, aw), p1 = xform_point(Xbw_b, bw);
    vec3 o0 = xform_vector(Xbw_a, off_a * n), o1 = xform_vector(Xbw_b, -off_b * n);
    if (c.big && c.aos_records) {  // the pair-heavy rollout's own copy: one line per slot, read back by its contact phases
        float* o = ct.cr + ((size_t)c.env * ncs + slot) * NT_CR_STRIDE;
        o[CD_POINT0] = p0.x; o[CD_POINT0 + 1] = p0.y; o[CD_POINT0 + 2] = p0.z;
        o[CD_POINT1] = p1.x; o[CD_POINT1 + 1] = p1.y; o[CD_POINT1 + 2] = p1.z;
        o[CD_OFFSET0] = o0.x; o[CD_OFFSET0 + 1] = o0.y; o[CD_OFFSET0 + 2] = o0.z;
        o[CD_OFFSET1] = o1.x; o[CD_OFFSET1 + 1] = o1.y; o[CD_OFFSET1 + 2] = o1.z;
        o[CD_NORMAL] = n.x; o[CD_NORMAL + 1] = n.y; o[CD_NORMAL + 2] = n.z;
        o[CD_MARGIN0] = off_a;
        o[CD_MARGIN1] = off_b;
        if (!c.hbm_out) return;  // (the Contacts buffers get the last substep's contacts only)
    }
    size_t gi = (size_t)slot * c.ES + c.env;
    ct.shape0[gi] = c.newton_shape_id(sa);
    ct.shape1[gi] = c.newton_shape_id(sb);
    float* D = ct.data;
    D[c.g(CD_POINT0 + 0, ncs, slot)] = p0.x; D[c.g(CD_POINT0 + 1, ncs, slot)] = p0.y; D[c.g(CD_POINT0 + 2, ncs, slot)] = p0.z;
    D[c.g(CD_POINT1 + 0, ncs, slot)] = p1.x; D[c.g(CD_POINT1 + 1, ncs, slot)] = p1.y; D[c.g(CD_POINT1 + 2, ncs, slot)] = p1.z;
    D[c.g(CD_OFFSET0 + 0, ncs, slot)] = o0.x; D[c.g(CD_OFFSET0 + 1, ncs, slot)] = o0.y; D[c.g(CD_OFFSET0 + 2, ncs, slot)] = o0.z;
    D[c.g(CD_OFFSET1 + 0, ncs, slot)] = o1.x; D[c.g(CD_OFFSET1 + 1, ncs, slot)] = o1.y; D[c.g(CD_OFFSET1 + 2, ncs, slot)] = o1.z;
    D[c.g(CD_NORMAL + 0, ncs, slot)] = n.x; D[c.g(CD_NORMAL + 1, ncs, slot)] = n.y; D[c.g(CD_NORMAL + 2, ncs, slot)] = n.z;
    D[c.g(CD_MARGIN0, ncs, slot)] = off_a;
    D[c.g(CD_MARGIN1, ncs, slot)] = off_b;
}

// ------------------------------------------------------------------------------------------------
// Staged variant of the pair phase (every tile except the pair-heavy one-environment-per-workgroup mode): ONE lane per
// pair runs the broad-phase test + the analytic primitive pair + the admission test and parks the admitted candidates in
// LDS (L.st: normal[3], then (center[3], dist) x 4 per pair); after a barrier one lane per contact SLOT turns its candidate
// into the body-frame record (19 stores).  Same arithmetic and emission order as a one-lane-per-slot evaluation -- the per-slot
// variant evaluated the pair once per slot lane (4x redundantly).  Convex pairs keep their single lane (it writes its
// slots itself) and skip the second stage.
// ------------------------------------------------------------------------------------------------
constexpr int ST_FLOATS = 19;
// the candidate test of one pair (broad_phase_common.py:20-38) on the staged shape transforms / AABBs
template <int EPB>
NT_DI bool pair_aabb_hit(const Ctx<EPB>& c, const int p) {
    xform Xa, Xb;
    vec3 loa, hia, lob, hib;
    shape_world(c, c.T.pair_a[p], Xa, loa, hia);
    shape_world(c, c.T.pair_b[p], Xb, lob, hib);
    return loa.x <= hib.x && hia.x >= lob.x && loa.y <= hib.y && hia.y >= lob.y && loa.z <= hib.z && hia.z >= lob.z;
}
// STAGED: admitted analytic candidates go to LDS (L.st) for the per-slot record stage; otherwise (pair-heavy tile) the
// pair lane writes its analytic records itself.  KNOWN_HIT: the pair comes from the compacted candidate list.
template <int EPB, bool CVX, bool STAGED = true, bool KNOWN_HIT = false>
NT_DI void pair_eval_item(const Ctx<EPB>& c, const int p) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp;
    int sa = c.T.pair_a[p], sb = c.T.pair_b[p];
    xform Xa, Xb;
    vec3 loa, hia, lob, hib;
    shape_world(c, sa, Xa, loa, hia);
    shape_world(c, sb, Xb, lob, hib);
    bool hit = KNOWN_HIT || (loa.x <= hib.x && hia.x >= lob.x && loa.y <= hib.y && hia.y >= lob.y && loa.z <= hib.z && hia.z >= lob.z);
    if (!KNOWN_HIT && c.hbm_out) ct.pair_hit[(size_t)p * c.ES + c.env] = hit ? 1 : 0;
    int nvalid = 0;
    if (hit) {
        int ta = c.T.shape_type[sa], tb = c.T.shape_type[sb];
        if (ta > tb) {  // sort by type (narrow_phase.py:525-528)
            int t = sa; sa = sb; sb = t;
            t = ta; ta = tb; tb = t;
            xform X = Xa; Xa = Xb; Xb = X;
            vec3 v = loa; loa = lob; lob = v;
            v = hia; hia = hib; hib = v;
        }
        vec3 scale_a = c.shape_scale(sa), scale_b = c.shape_scale(sb);
        float margin_a = c.shape_f(sa, SP_MARGIN), margin_b = c.shape_f(sb, SP_MARGIN);
        float gap_sum = c.shape_f(sa, SP_GAP) + c.shape_f(sb, SP_GAP);
        bool to_gjk = ta >= GEO_ELLIPSOID || tb == GEO_CONE || (ta == GEO_CAPSULE && tb > GEO_CAPSULE);
        bool barrel_on_cap = false;
        if constexpr (CVX) {
            // barrel cylinders (scale.z = radius of the side arc): sphere pairs always take MPR / GJK (narrow_phase.py:847,999), plane pairs
            // unless the cylinder rests on an end cap (narrow_phase.py:682-686) -- decided per environment and substep.  The host stores
            // such pairs with the convex ones (they own manifold slots); both outcomes are written by the convex branch below
            if (tb == GEO_CYLINDER && scale_b.z != 0.0f && p >= m.np_analytic && (ta == GEO_SPHERE || ta == GEO_PLANE)) {
                to_gjk = true;
                if (ta == GEO_PLANE) {
                    barrel_on_cap = true;
                    if (scale_b.z > 0.0f)
                        barrel_on_cap = fabsf(dot(quat_rotate(Xa.q, vec3(0.0f, 0.0f, 1.0f)), quat_rotate(Xb.q, vec3(0.0f, 0.0f, 1.0f)))) * scale_b.z >= scale_b.y;
                }
            }
        }
        if (!to_gjk) {
            float ra = (ta == GEO_SPHERE || ta == GEO_CAPSULE) ? scale_a.x : 0.0f;
            float rb = (tb == GEO_SPHERE || tb == GEO_CAPSULE) ? scale_b.x : 0.0f;
            Contacts4 k4;
            primitive_pair(ta, tb, Xa, Xb, scale_a, scale_b, gap_sum + margin_a + margin_b, k4);
            float total_sep = ra + rb + margin_a + margin_b;
            vec3 n = normalize(k4.normal);
            if constexpr (STAGED) c.st_lv3(c.L.st, 0, m.np, p, n);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float dist = k4.dist(i);
                bool ok = dist < NT_MAXVAL;
                if (ok) {
                    vec3 center = k4.pos(i);
                    vec3 aw = center - n * (0.5f * dist + ra);
                    vec3 bw = center + n * (0.5f * dist + rb);
                    float d = dot(bw - aw, n) - total_sep;
                    ok = d <= gap_sum;
                    if (ok) {  // the nvalid-th admitted candidate
                        if constexpr (STAGED) {
                            c.st_lv3(c.L.st, 3 + 4 * nvalid, m.np, p, center);
                            c.l(c.L.st, 6 + 4 * nvalid, m.np, p) = dist;
                        } else {
                            write_contact_slot(c, p * cpp + nvalid, sa, sb, center, n, dist, ra, rb, margin_a, margin_b);
                        }
                    }
                }
                nvalid += ok ? 1 : 0;
            }
            if constexpr (!STAGED)
                for (int i = nvalid; (!c.big || c.hbm_out) && i < cpp; ++i) {
                    size_t gi = (size_t)(p * cpp + i) * c.ES + c.env;
                    ct.shape0[gi] = -1;
                    ct.shape1[gi] = -1;
                }
        }
        if constexpr (CVX) {
            if (barrel_on_cap) {
                // a barrel cylinder resting on an end cap: the analytic plane-cylinder routine, admitted like the primitive kernel admits
                // (narrow_phase.py:791-797,872-955), written by this lane into the pair's manifold slots.  Kept apart from the convex
                // branch below (one rolled loop, one writer): sharing its ConvexContacts record cost the convex rollouts 7 %
                Contacts4 k4;
                plane_cylinder(quat_rotate(Xa.q, vec3(0.0f, 0.0f, 1.0f)), Xa.p, Xb.p, quat_rotate(Xb.q, vec3(0.0f, 0.0f, 1.0f)), scale_b.x, scale_b.y, k4);
                const vec3 n = normalize(k4.normal);
#pragma nounroll
                for (int i = 0; i < 4; ++i) {
                    const float dist = k4.dist(i);
                    if (!(dist < NT_MAXVAL)) continue;
                    const vec3 center = k4.pos(i);
                    const vec3 aw = center - n * (0.5f * dist + 0.0f), bw = center + n * (0.5f * dist + 0.0f);
                    if (!(dot(bw - aw, n) - (0.0f + 0.0f + margin_a + margin_b) <= gap_sum)) continue;
                    write_contact_slot(c, p * cpp + nvalid, sa, sb, center, n, dist, 0.0f, 0.0f, margin_a, margin_b);
                    nvalid += 1;
                }
                for (int i = nvalid; (!c.big || c.hbm_out) && i < cpp; ++i) {
                    size_t gi = (size_t)(p * cpp + i) * c.ES + c.env;
                    ct.shape0[gi] = -1;
                    ct.shape1[gi] = -1;
                }
            }
            if (p >= m.np_analytic && !barrel_on_cap) {
                ConvexContacts cc;
                Geom ga, gb;
                ga.type = ta; ga.scale = scale_a;
                gb.type = tb; gb.scale = scale_b;
                if (ta == GEO_PLANE) ga.scale = vec3(scale_a.x * 0.5f, scale_a.y * 0.5f, 0.0f);
                if (tb == GEO_PLANE) gb.scale = vec3(scale_b.x * 0.5f, scale_b.y * 0.5f, 0.0f);
                if (ta == GEO_CONVEX_MESH) {
                    ga.points = m.mesh_points + 3 * c.T.shape_mesh_start[sa];
                    ga.count = c.T.shape_mesh_count[sa];
                    const float* mb = m.shape_mesh_bounds + 6 * sa;
                    ga.center = 0.5f * (vmin(cw_mul(vec3(mb[0], mb[1], mb[2]), scale_a), cw_mul(vec3(mb[3], mb[4], mb[5]), scale_a)) +
                                        vmax(cw_mul(vec3(mb[0], mb[1], mb[2]), scale_a), cw_mul(vec3(mb[3], mb[4], mb[5]), scale_a)));
                }
                if (tb == GEO_CONVEX_MESH) {
                    gb.points = m.mesh_points + 3 * c.T.shape_mesh_start[sb];
                    gb.count = c.T.shape_mesh_count[sb];
                    const float* mb = m.shape_mesh_bounds + 6 * sb;
                    gb.center = 0.5f * (vmin(cw_mul(vec3(mb[0], mb[1], mb[2]), scale_b), cw_mul(vec3(mb[3], mb[4], mb[5]), scale_b)) +
                                        vmax(cw_mul(vec3(mb[0], mb[1], mb[2]), scale_b), cw_mul(vec3(mb[3], mb[4], mb[5]), scale_b)));
                }
                PolyRef poly;  // manifold polygon scratch: per convex pair, or (pair-heavy tile) per lane
                poly.base = &c.lds[(c.L.poly + 20 * (c.big ? c.slot : p - m.np_analytic)) * Ctx<EPB>::N + c.e];
                poly.stride = Ctx<EPB>::N;
                convex_pair(ga, gb, Xa, Xb, margin_a, margin_b, gap_sum, lob, hib, poly, cc);
                float ra = (ta == GEO_SPHERE || ta == GEO_CAPSULE) ? scale_a.x : 0.0f;
                float rb = (tb == GEO_SPHERE || tb == GEO_CAPSULE) ? scale_b.x : 0.0f;
                vec3 n = normalize(cc.normal);
                nvalid = cc.count < cpp ? cc.count : cpp;
                for (int i = 0; i < cpp; ++i) {
                    if (i < nvalid) {
                        write_contact_slot(c, p * cpp + i, sa, sb, cc.center(i), n, cc.distance(i), ra, rb, margin_a, margin_b);
                    } else if (!c.big || c.hbm_out) {
                        size_t gi = (size_t)(p * cpp + i) * c.ES + c.env;
                        ct.shape0[gi] = -1;
                        ct.shape1[gi] = -1;
                    }
                }
            }
        }
    } else if (CVX && p >= m.np_analytic) {
        for (int i = 0; i < cpp; ++i) {
            size_t gi = (size_t)(p * cpp + i) * c.ES + c.env;
            ct.shape0[gi] = -1;
            ct.shape1[gi] = -1;
        }
    }
    c.l(c.L.pc, 0, m.np, p) = (float)nvalid;
    c.l(c.L.pm, 0, m.np, p) = (float)nvalid;
    if (STAGED && c.lds_records && nvalid > 0) {
        // LDS-record tiles: the pair lane appends its live contacts to the environment's list itself (one LDS atomic), so no prefix
        // pass exists.  The ORDER of the list varies with the waves' timing and does not matter: every contact owns its slot's record
        // (L.cr / L.cw) and the body lanes sum in slot order.
        const int base = atomicAdd(reinterpret_cast<int*>(&c.l(c.L.lc, 0, 1, 0)), nvalid);
        for (int k = 0; k < nvalid; ++k) *reinterpret_cast<int*>(&c.l(c.L.lt, 0, 1, base + k)) = (p << 4) | k;
    }
}
// LDS-record tiles, second stage: lane i converts the environment's i-th LIVE contact (L.lt) into the body-frame record of its slot in
// L.cr -- one pass over the live contacts (16 for the standing quadruped) instead of two over all np * cpp slots, no HBM store and no
// prefix lanes.  Same arithmetic as write_contact_slot.
template <int EPB>
NT_DI void contact_record_item_lds(const Ctx<EPB>& c, const int entry) {
    const nt_model& m = c.a.m;
    const int cpp = m.cpp, ncs = m.np * cpp;
    const int p = entry >> 4, k = entry & 15, slot = p * cpp + k;
    const int* d = c.T.pair_desc + 4 * p;  // type-sorted shapes + their bodies
    const int sa = d[0], sb = d[1], ba = d[2], bb = (d[3] << 2) >> 2;
    const int ta = c.T.shape_type[sa], tb = c.T.shape_type[sb];
    const float ra = (ta == GEO_SPHERE || ta == GEO_CAPSULE) ? c.shape_f(sa, SP_SCALE) : 0.0f;
    const float rb = (tb == GEO_SPHERE || tb == GEO_CAPSULE) ? c.shape_f(sb, SP_SCALE) : 0.0f;
    const vec3 n = c.lv3(c.L.st, 0, m.np, p);
    const vec3 center = c.lv3(c.L.st, 3 + 4 * k, m.np, p);
    const float dist = c.l(c.L.st, 6 + 4 * k, m.np, p);
    const xform Xbw_a = ba < 0 ? xform() : xform_inverse(c.body_q_in(ba));
    const xform Xbw_b = bb < 0 ? xform() : xform_inverse(c.body_q_in(bb));
    const float off_a = ra + c.shape_f(sa, SP_MARGIN), off_b = rb + c.shape_f(sb, SP_MARGIN);
    const vec3 aw = center - n * (0.5f * dist + ra);
    const vec3 bw = center + n * (0.5f * dist + rb);
    c.st_lv3(c.L.cr, CD_POINT0, ncs, slot, xform_point(Xbw_a, aw));
    c.st_lv3(c.L.cr, CD_POINT1, ncs, slot, xform_point(Xbw_b, bw));
    c.st_lv3(c.L.cr, CD_OFFSET0, ncs, slot, xform_vector(Xbw_a, off_a * n));
    c.st_lv3(c.L.cr, CD_OFFSET1, ncs, slot, xform_vector(Xbw_b, -off_b * n));
    c.st_lv3(c.L.cr, CD_NORMAL, ncs, slot, n);
    c.l(c.L.cr, CD_MARGIN0, ncs, slot) = off_a;
    c.l(c.L.cr, CD_MARGIN1, ncs, slot) = off_b;
}
// ... and at the last substep of the launch the Contacts buffers in HBM receive them (one lane per slot: ids + record, or -1 ids)
template <int EPB>
NT_DI void contact_export_item_lds(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp;
    const int p = slot / cpp, k = slot - p * cpp;
    const size_t gi = (size_t)slot * c.ES + c.env;
    if (k < (int)c.l(c.L.pm, 0, m.np, p)) {
        const int* d = c.T.pair_desc + 4 * p;
        ct.shape0[gi] = c.newton_shape_id(d[0]);
        ct.shape1[gi] = c.newton_shape_id(d[1]);
        float v[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) v[i] = c.l(c.L.cr, i, ncs, slot);
#pragma unroll
        for (int i = 0; i < 17; ++i) ct.data[c.g(i, ncs, slot)] = v[i];
    } else {
        ct.shape0[gi] = -1;
        ct.shape1[gi] = -1;
    }
}
// second stage, analytic pairs only: contact slot (p, k) <- the pair's k-th admitted candidate
template <int EPB>
NT_DI void contact_write_item(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const int cpp = m.cpp;
    const int p = slot / cpp, k = slot - p * cpp;
    if (p >= m.np_analytic) return;  // convex pairs wrote their slots in the first stage
    if (k < (int)c.l(c.L.pm, 0, m.np, p)) {
        int sa = c.T.pair_a[p], sb = c.T.pair_b[p];
        int ta = c.T.shape_type[sa], tb = c.T.shape_type[sb];
        if (ta > tb) {
            int t = sa; sa = sb; sb = t;
            t = ta; ta = tb; tb = t;
        }
        float ra = (ta == GEO_SPHERE || ta == GEO_CAPSULE) ? c.shape_f(sa, SP_SCALE) : 0.0f;
        float rb = (tb == GEO_SPHERE || tb == GEO_CAPSULE) ? c.shape_f(sb, SP_SCALE) : 0.0f;
        vec3 n = c.lv3(c.L.st, 0, m.np, p);
        vec3 center = c.lv3(c.L.st, 3 + 4 * k, m.np, p);
        float dist = c.l(c.L.st, 6 + 4 * k, m.np, p);
        write_contact_slot(c, slot, sa, sb, center, n, dist, ra, rb, c.shape_f(sa, SP_MARGIN), c.shape_f(sb, SP_MARGIN));
    } else {
        size_t gi = (size_t)slot * c.ES + c.env;
        c.a.ct.shape0[gi] = -1;
        c.a.ct.shape1[gi] = -1;
    }
}
// Pair-heavy tile (one environment per workgroup, contact records in HBM): stage 1 tests every candidate pair's AABBs
// (one lane per pair) and appends the hits to a block-shared list (LDS atomic counter; the order of the list is
// irrelevant: every pair owns its fixed contact slots), misses clear their slots; stage 2 deals the hits densely to the
// lanes, so the narrow phase runs with full waves on ~250 live candidates instead of 2 336 mostly-empty pair lanes.
template <int EPB>
NT_DI void phase_pairs_big_broad(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    if (threadIdx.x == 0) *c.T.hit_count = 0;
    __syncthreads();
    if (c.valid)
        for (int p = c.slot; p < m.np; p += c.nslot) {
            const bool hit = pair_aabb_hit(c, p);
            if (c.hbm_out) ct.pair_hit[(size_t)p * c.ES + c.env] = hit ? 1 : 0;
            if (hit) {
                c.T.hit_list[atomicAdd(c.T.hit_count, 1)] = p;
            } else {
                for (int i = 0; c.hbm_out && i < m.cpp; ++i) {
                    size_t gi = (size_t)(p * m.cpp + i) * c.ES + c.env;
                    ct.shape0[gi] = -1;
                    ct.shape1[gi] = -1;
                }
                c.l(c.L.pc, 0, m.np, p) = 0.0f;
                c.l(c.L.pm, 0, m.np, p) = 0.0f;
            }
        }
    __syncthreads();
}
template <int EPB, bool CVX>
NT_DI void phase_pairs_big_narrow(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int nhit = *c.T.hit_count;
    for (int i = c.slot; i < nhit; i += c.nslot) pair_eval_item<EPB, CVX, false, true>(c, c.T.hit_list[i]);
}

template <int EPB, bool CVX>
NT_DI void phase_pair_eval(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int p = c.tslot; p < c.a.m.np; p += c.nslot) pair_eval_item<EPB, CVX>(c, p);
}
// Staged tiles with more candidate pairs than slot lanes (e.g. the 8-box stacks: 36 pairs, 32 lanes, ~8 AABB hits): the same
// two-stage split per ENVIRONMENT.  Stage 1: one lane per pair tests the AABBs and appends the hits to the environment's list
// in LDS (atomic counter, zeroed during the shape phase; the order of the list is irrelevant, every pair owns its slots);
// misses retire here.  Stage 2 deals the hits densely to the lanes: one pass of the narrow phase instead of ceil(np / lanes).
template <int EPB>
NT_DI bool pairs_compacted(const Ctx<EPB>& c) { return !c.big && c.a.m.np > c.nslot; }
template <int EPB>
NT_DI void phase_pair_broad_staged(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    int* count = reinterpret_cast<int*>(&c.l(c.L.hc, 0, 1, 0));
    for (int p = c.slot; p < m.np; p += c.nslot) {
        const bool hit = pair_aabb_hit(c, p);
        if (c.hbm_out) ct.pair_hit[(size_t)p * c.ES + c.env] = hit ? 1 : 0;
        if (hit) {
            *reinterpret_cast<int*>(&c.l(c.L.hl, 0, m.np, atomicAdd(count, 1))) = p;
        } else {
            if (p >= m.np_analytic)  // (analytic pairs: the record stage clears the slots of a pair without candidates)
                for (int i = 0; i < m.cpp; ++i) {
                    size_t gi = (size_t)(p * m.cpp + i) * c.ES + c.env;
                    ct.shape0[gi] = -1;
                    ct.shape1[gi] = -1;
                }
            c.l(c.L.pc, 0, m.np, p) = 0.0f;
            c.l(c.L.pm, 0, m.np, p) = 0.0f;
        }
    }
}
template <int EPB, bool CVX>
NT_DI void phase_pair_narrow_staged(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int nhit = *reinterpret_cast<const int*>(&c.l(c.L.hc, 0, 1, 0));
    for (int i = c.slot; i < nhit; i += c.nslot)
        pair_eval_item<EPB, CVX, true, true>(c, *reinterpret_cast<const int*>(&c.l(c.L.hl, 0, c.a.m.np, i)));
}
template <int EPB>
NT_DI void phase_contact_write(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int nas = c.a.m.np_analytic * c.a.m.cpp;
    for (int s = c.slot; s < nas; s += c.nslot) contact_write_item(c, s);
}

// Exclusive prefix of the per-pair live-contact counts (L.pm -> L.px[0..np]) so that the fused solver phases can hand the
// i-th LIVE contact of an environment to lane i instead of visiting every fixed slot (pair-heavy scenes have ~250 live
// contacts in 11 680 slots).  Two-level: up to 16 lanes of the environment sum one chunk of pairs each, then each turns
// its chunk into prefixes.  `partial` = 8 scratch rows that are free between the pair phase and the solver phases (the shape
// transforms of the collide scratch: 13 rows per shape, dead once the pairs are done).
constexpr int NT_PREFIX_LANES = 8;
template <int EPB>
NT_DI void phase_pair_prefix_partials(const Ctx<EPB>& c, int partial) {
    const int np = c.a.m.np;
    if (!c.valid || c.slot >= NT_PREFIX_LANES) return;
    const int lanes = c.nslot < NT_PREFIX_LANES ? c.nslot : NT_PREFIX_LANES;
    const int chunk = (np + lanes - 1) / lanes;
    if (c.slot >= lanes) return;
    int sum = 0;
    for (int p = c.slot * chunk; p < np && p < (c.slot + 1) * chunk; ++p) sum += (int)c.l(c.L.pm, 0, np, p);
    c.lds[(partial + c.slot) * Ctx<EPB>::N + c.e] = (float)sum;
}
// prefix lane `lane` (0 .. NT_PREFIX_LANES-1) of the calling lane's environment
template <int EPB>
NT_DI void prefix_lane(const Ctx<EPB>& c, int lane, int partial, bool store_env_count, bool one_level) {
    const int np = c.a.m.np;
    const int lanes = c.nslot < NT_PREFIX_LANES ? c.nslot : NT_PREFIX_LANES;
    const int chunk = (np + lanes - 1) / lanes;
    if (lane >= lanes) return;
    int acc = 0;
    if (one_level) {  // few pairs: every lane sums the counts in front of its chunk itself (no partials, one barrier less)
        for (int p = 0; p < np && p < lane * chunk; ++p) acc += (int)c.l(c.L.pm, 0, np, p);
    } else {
        for (int s = 0; s < lane; ++s) acc += (int)c.lds[(partial + s) * Ctx<EPB>::N + c.e];
    }
    for (int p = lane * chunk; p < np && p < (lane + 1) * chunk; ++p) {
        c.l(c.L.px, 0, 1, p) = (float)acc;
        const int live = (int)c.l(c.L.pm, 0, np, p);
        if (c.L.has_lt)  // the compacted live-contact list of the fused contact phases (entry: pair << 4 | sub-contact)
            for (int k = 0; k < live; ++k) *reinterpret_cast<int*>(&c.l(c.L.lt, 0, 1, acc + k)) = (p << 4) | k;
        acc += live;
    }
    if (lane == lanes - 1) {  // the last chunk ends at np (chunks past np are empty, their running sum is the total too)
        c.l(c.L.px, 0, 1, np) = (float)acc;
        if (store_env_count) c.a.ct.env_count[c.env] = acc;  // per-env totals: an API-boundary output
    }
}
template <int EPB>
NT_DI void phase_pair_prefix_scan(const Ctx<EPB>& c, int partial, bool store_env_count, bool one_level) {
    if (!c.valid || c.slot >= NT_PREFIX_LANES) return;
    prefix_lane(c, c.slot, partial, store_env_count, one_level);
}
