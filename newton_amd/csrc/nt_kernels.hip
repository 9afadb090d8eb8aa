// nt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of libnewton_hip.so.
//
// Design (DESIGN.md has the long form):
//  * env-major SoA in HBM: base[(comp * nslot + slot) * ES + env]; a wave reads 64 consecutive envs of one
//    (component, slot) = one 256 B coalesced request.
//  * one workgroup owns EPB environments for a whole substep (or a whole rollout): thread -> (env, slot) with
//    env = blockIdx * EPB + tid % EPB and slot = tid / EPB.  A "slot" thread plays body `slot`, joint `slot`,
//    shape `slot` and candidate pair `slot` in the respective phases, so with EPB == 64 every branch on joint
//    or shape type is wave-uniform.
//  * body state lives in LDS ([comp][slot][EPB], conflict-free: lanes of a wave differ in env first);
//    constraint threads publish their per-joint / per-pair corrections in LDS and the owning body thread sums
//    them in ascending joint / pair order through a CSR incidence list -- no float atomics, deterministic,
//    and the same order a serial ascending-tid Warp-CPU launch uses for wp.atomic_add.
//  * no MFMA: the largest dense object on this path is a 3x3 inertia.
//
// Reference behaviour (file:line under /root/reference) is cited per phase.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"
#include "nt_math.hpp"
#include "nt_primitives.hpp"

using namespace nt;

namespace {

enum JointType : int { JT_PRISMATIC = 0, JT_REVOLUTE = 1, JT_BALL = 2, JT_FIXED = 3, JT_FREE = 4, JT_DISTANCE = 5, JT_D6 = 6, JT_ROD = 7 };
constexpr int BODY_KINEMATIC = 2;

// body_param component indices
constexpr int BP_COM = 0, BP_INV_MASS = 3, BP_INERTIA = 4, BP_INV_INERTIA = 13, BP_MASS = 22;
// dof_param
constexpr int DP_AXIS = 0, DP_LIMIT_LOWER = 3, DP_LIMIT_UPPER = 4, DP_TARGET_KE = 5, DP_TARGET_KD = 6;
// shape_param
constexpr int SP_XFORM = 0, SP_SCALE = 7, SP_MARGIN = 10, SP_GAP = 11, SP_MU = 12, SP_MU_TORSIONAL = 13, SP_MU_ROLLING = 14;
// contact data
constexpr int CD_POINT0 = 0, CD_POINT1 = 3, CD_OFFSET0 = 6, CD_OFFSET1 = 9, CD_NORMAL = 12, CD_MARGIN0 = 15, CD_MARGIN1 = 16;

struct LdsLayout {
    int bq, bqd;      // persistent: body_q [7][nb], body_qd [6][nb]
    int u;            // union region
    int jw, pw, bf;   // step view of u: joint wrench/delta [12][nj], pair delta [14][np], body_f_tmp [6][nb] (aliases pw)
    int sx, sa;       // collide view of u: shape world xform [7][ns], shape aabb [6][ns]
    int pc;           // collide: per-pair contact count [np] (after sa)
    int floats_per_env;
};

__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }

__host__ __device__ inline LdsLayout make_layout(int nb, int nj, int np, int ns) {
    LdsLayout L;
    L.bq = 0;
    L.bqd = 7 * nb;
    L.u = 13 * nb;
    L.jw = L.u;
    L.pw = L.u + 12 * nj;
    L.bf = L.pw;
    L.sx = L.u;
    L.sa = L.u + 7 * ns;
    L.pc = L.u + 13 * ns;
    int step_sz = 12 * nj + imax(14 * np, 6 * nb);
    int coll_sz = 13 * ns + np;
    L.floats_per_env = 13 * nb + imax(step_sz, coll_sz);
    return L;
}

struct KArgs {
    nt_model m;
    nt_state s_in, s_out;
    nt_control c;
    nt_contacts ct;
    nt_xpbd_params p;
    float dt;
    int substeps;
    int has_contacts;
    int nslot;  // slot-threads per environment
};

template <int EPB>
struct Ctx {
    const KArgs& a;
    float* lds;
    LdsLayout L;
    int e, slot, env, nslot;
    int ES;
    bool valid;

    NT_DI Ctx(const KArgs& a_, float* lds_) : a(a_), lds(lds_) {
        L = make_layout(a.m.nb, a.m.nj, a.m.np, a.m.ns);
        e = threadIdx.x % EPB;
        slot = threadIdx.x / EPB;
        nslot = a.nslot;
        env = blockIdx.x * EPB + e;
        ES = a.m.env_stride;
        valid = env < a.m.env_count && slot < nslot;
    }
    // LDS element (field offset, component, slots in field, slot)
    NT_DI float& l(int off, int comp, int nslot, int s) const { return lds[(off + comp * nslot + s) * EPB + e]; }
    // global SoA element
    NT_DI size_t g(int comp, int nslot, int s) const { return (size_t)(comp * nslot + s) * ES + env; }

    NT_DI xform lds_xform(int off, int nslot, int s) const {
        return xform(vec3(l(off, 0, nslot, s), l(off, 1, nslot, s), l(off, 2, nslot, s)),
                     quat(l(off, 3, nslot, s), l(off, 4, nslot, s), l(off, 5, nslot, s), l(off, 6, nslot, s)));
    }
    NT_DI void st_lds_xform(int off, int nslot, int s, const xform& t) const {
        l(off, 0, nslot, s) = t.p.x; l(off, 1, nslot, s) = t.p.y; l(off, 2, nslot, s) = t.p.z;
        l(off, 3, nslot, s) = t.q.x; l(off, 4, nslot, s) = t.q.y; l(off, 5, nslot, s) = t.q.z; l(off, 6, nslot, s) = t.q.w;
    }
    NT_DI vec3 lds_vec3(int off, int comp0, int nslot, int s) const {
        return vec3(l(off, comp0, nslot, s), l(off, comp0 + 1, nslot, s), l(off, comp0 + 2, nslot, s));
    }
    NT_DI void st_lds_vec3(int off, int comp0, int nslot, int s, vec3 v) const {
        l(off, comp0, nslot, s) = v.x; l(off, comp0 + 1, nslot, s) = v.y; l(off, comp0 + 2, nslot, s) = v.z;
    }
    NT_DI vec3 g_vec3(const float* base, int comp0, int nslot, int s) const {
        return vec3(base[g(comp0, nslot, s)], base[g(comp0 + 1, nslot, s)], base[g(comp0 + 2, nslot, s)]);
    }
    NT_DI mat33 g_mat33(const float* base, int comp0, int nslot, int s) const {
        return mat33(base[g(comp0, nslot, s)], base[g(comp0 + 1, nslot, s)], base[g(comp0 + 2, nslot, s)],
                     base[g(comp0 + 3, nslot, s)], base[g(comp0 + 4, nslot, s)], base[g(comp0 + 5, nslot, s)],
                     base[g(comp0 + 6, nslot, s)], base[g(comp0 + 7, nslot, s)], base[g(comp0 + 8, nslot, s)]);
    }
    NT_DI xform g_xform(const float* base, int comp0, int nslot, int s) const {
        return xform(g_vec3(base, comp0, nslot, s), quat(base[g(comp0 + 3, nslot, s)], base[g(comp0 + 4, nslot, s)],
                                                          base[g(comp0 + 5, nslot, s)], base[g(comp0 + 6, nslot, s)]));
    }
    NT_DI xform body_q(int b) const { return lds_xform(L.bq, a.m.nb, b); }
    NT_DI spatial body_qd(int b) const {
        return spatial(lds_vec3(L.bqd, 0, a.m.nb, b), lds_vec3(L.bqd, 3, a.m.nb, b));
    }
    // effective inverse mass / inertia: zero for kinematic bodies (solver.py:173-187)
    NT_DI float inv_mass(int b) const {
        if (a.m.body_flags[b] & BODY_KINEMATIC) return 0.0f;
        return a.m.body_param[g(BP_INV_MASS, a.m.nb, b)];
    }
    NT_DI mat33 inv_inertia(int b) const {
        if (a.m.body_flags[b] & BODY_KINEMATIC) return mat33();
        return g_mat33(a.m.body_param, BP_INV_INERTIA, a.m.nb, b);
    }
    NT_DI vec3 com(int b) const { return g_vec3(a.m.body_param, BP_COM, a.m.nb, b); }

    // shape accessors: s < ns local (per-env params), otherwise global table
    NT_DI float shape_f(int s, int comp) const {
        if (s < a.m.ns) return a.m.shape_param[g(comp, a.m.ns, s)];
        return a.m.gshape_param[(s - a.m.ns) * NT_SHAPE_PARAM_FLOATS + comp];
    }
    NT_DI vec3 shape_scale(int s) const { return vec3(shape_f(s, SP_SCALE), shape_f(s, SP_SCALE + 1), shape_f(s, SP_SCALE + 2)); }
    NT_DI xform shape_local_xform(int s) const {
        return xform(vec3(shape_f(s, 0), shape_f(s, 1), shape_f(s, 2)), quat(shape_f(s, 3), shape_f(s, 4), shape_f(s, 5), shape_f(s, 6)));
    }
    NT_DI int newton_shape_id(int s) const {  // flat Newton shape index (world-major locals, then globals)
        return s < a.m.ns ? env * a.m.ns + s : a.m.env_count * a.m.ns + (s - a.m.ns);
    }
    NT_DI int local_shape_id(int gid) const {
        int eg = a.m.env_count * a.m.ns;
        return gid >= eg ? a.m.ns + (gid - eg) : gid - env * a.m.ns;
    }
};

// ------------------------------------------------------------------------------------------------
// state load / store
// ------------------------------------------------------------------------------------------------
template <int EPB>
NT_DI void load_state(const Ctx<EPB>& c, const nt_state& s) {
    const int nb = c.a.m.nb;
    if (!c.valid) return;
    for (int b = c.slot; b < nb; b += c.nslot) {
#pragma unroll
        for (int k = 0; k < 7; ++k) c.l(c.L.bq, k, nb, b) = s.body_q[c.g(k, nb, b)];
#pragma unroll
        for (int k = 0; k < 6; ++k) c.l(c.L.bqd, k, nb, b) = s.body_qd[c.g(k, nb, b)];
    }
}
template <int EPB>
NT_DI void store_state(const Ctx<EPB>& c, const nt_state& s) {
    const int nb = c.a.m.nb;
    if (!c.valid) return;
    for (int b = c.slot; b < nb; b += c.nslot) {
#pragma unroll
        for (int k = 0; k < 7; ++k) s.body_q[c.g(k, nb, b)] = c.l(c.L.bq, k, nb, b);
#pragma unroll
        for (int k = 0; k < 6; ++k) s.body_qd[c.g(k, nb, b)] = c.l(c.L.bqd, k, nb, b);
    }
}

// ------------------------------------------------------------------------------------------------
// collide: compute_shape_aabbs (collide.py:283-472)
// ------------------------------------------------------------------------------------------------
NT_DI void shape_aabb(int geo_type, const xform& X, vec3 scale, float effective_gap, vec3& lo, vec3& hi) {
    vec3 pos = X.p;
    quat q = X.q;
    vec3 mv(effective_gap, effective_gap, effective_gap);
    bool infinite_plane = (geo_type == GEO_PLANE) && (scale.x == 0.0f && scale.y == 0.0f);
    if (infinite_plane) {
        vec3 normal = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
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
        vec3 r0 = quat_rotate(q, vec3(1.0f, 0.0f, 0.0f));
        vec3 r1 = quat_rotate(q, vec3(0.0f, 1.0f, 0.0f));
        vec3 r2 = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
        he = vec3(fabsf(r0.x) * scale.x + fabsf(r1.x) * scale.y + fabsf(r2.x) * scale.z,
                  fabsf(r0.y) * scale.x + fabsf(r1.y) * scale.y + fabsf(r2.y) * scale.z,
                  fabsf(r0.z) * scale.x + fabsf(r1.z) * scale.y + fabsf(r2.z) * scale.z);
    } else if (geo_type == GEO_CAPSULE) {
        vec3 axis = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
        he = vec3(scale.x, scale.x, scale.x) + vabs(axis) * scale.y;
    } else if (geo_type == GEO_CYLINDER) {
        float radius = scale.x, hh = scale.y, barrel = scale.z;
        if (barrel >= hh && barrel > 0.0f) radius += (hh * hh) / (barrel + __fsqrt_rn(barrel * barrel - hh * hh));
        vec3 r0 = quat_rotate(q, vec3(1.0f, 0.0f, 0.0f));
        vec3 r1 = quat_rotate(q, vec3(0.0f, 1.0f, 0.0f));
        vec3 r2 = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
        he = vec3(radius * __fsqrt_rn(r0.x * r0.x + r1.x * r1.x) + hh * fabsf(r2.x),
                  radius * __fsqrt_rn(r0.y * r0.y + r1.y * r1.y) + hh * fabsf(r2.y),
                  radius * __fsqrt_rn(r0.z * r0.z + r1.z * r1.z) + hh * fabsf(r2.z));
    } else if (geo_type == GEO_ELLIPSOID) {
        mat33 R = quat_to_matrix(q);
        he = vec3(length(vec3(R.m00 * scale.x, R.m01 * scale.y, R.m02 * scale.z)),
                  length(vec3(R.m10 * scale.x, R.m11 * scale.y, R.m12 * scale.z)),
                  length(vec3(R.m20 * scale.x, R.m21 * scale.y, R.m22 * scale.z)));
    } else {
        // finite planes / cones: conservative bounding sphere
        float r = (geo_type == GEO_PLANE) ? 0.5f * __fsqrt_rn(scale.x * scale.x + scale.y * scale.y) : scale.x + scale.y;
        he = vec3(r, r, r);
    }
    lo = pos - he - mv;
    hi = pos + he + mv;
}

template <int EPB>
NT_DI void phase_shapes(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    if (!c.valid) return;
    for (int s = c.slot; s < m.ns; s += c.nslot) {
        int body = m.shape_body[s];
        xform X = c.shape_local_xform(s);
        if (body >= 0) X = c.body_q(body) * X;
        vec3 lo, hi;
        shape_aabb(m.shape_type[s], X, c.shape_scale(s), c.shape_f(s, SP_MARGIN) + c.shape_f(s, SP_GAP), lo, hi);
        c.st_lds_xform(c.L.sx, m.ns, s, X);
        c.st_lds_vec3(c.L.sa, 0, m.ns, s, lo);
        c.st_lds_vec3(c.L.sa, 3, m.ns, s, hi);
    }
}

template <int EPB>
NT_DI void shape_world(const Ctx<EPB>& c, int s, xform& X, vec3& lo, vec3& hi) {
    const nt_model& m = c.a.m;
    if (s < m.ns) {
        X = c.lds_xform(c.L.sx, m.ns, s);
        lo = c.lds_vec3(c.L.sa, 0, m.ns, s);
        hi = c.lds_vec3(c.L.sa, 3, m.ns, s);
    } else {
        X = c.shape_local_xform(s);  // global shapes are static (shape_body == -1)
        shape_aabb(m.shape_type[s], X, c.shape_scale(s), c.shape_f(s, SP_MARGIN) + c.shape_f(s, SP_GAP), lo, hi);
    }
}

// broad phase test (broad_phase_common.py:20-38, cutoff 0: AABBs are pre-expanded) + narrow phase primitive
// dispatch (narrow_phase.py:458-1014) + contact writer (collide.py:166-254)
template <int EPB>
NT_DI void pair_item(const Ctx<EPB>& c, const int p) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp;
    const int ncs = m.np * cpp;
    int sa = m.pair_a[p], sb = m.pair_b[p];
    xform Xa, Xb;
    vec3 loa, hia, lob, hib;
    shape_world(c, sa, Xa, loa, hia);
    shape_world(c, sb, Xb, lob, hib);
    bool hit = loa.x <= hib.x && hia.x >= lob.x && loa.y <= hib.y && hia.y >= lob.y && loa.z <= hib.z && hia.z >= lob.z;
    ct.pair_hit[(size_t)p * c.ES + c.env] = hit ? 1 : 0;

    int nvalid = 0;
    if (hit) {
        int ta = m.shape_type[sa], tb = m.shape_type[sb];
        if (ta > tb) {  // sort by type (narrow_phase.py:525-528)
            int t = sa; sa = sb; sb = t;
            t = ta; ta = tb; tb = t;
            xform X = Xa; Xa = Xb; Xb = X;
        }
        vec3 scale_a = c.shape_scale(sa), scale_b = c.shape_scale(sb);
        float margin_a = c.shape_f(sa, SP_MARGIN), margin_b = c.shape_f(sb, SP_MARGIN);
        float gap_sum = c.shape_f(sa, SP_GAP) + c.shape_f(sb, SP_GAP);
        bool to_gjk = ta >= GEO_ELLIPSOID || tb == GEO_CONE || (ta == GEO_CAPSULE && tb > GEO_CAPSULE);
        if (!to_gjk) {
            float ra = (ta == GEO_SPHERE || ta == GEO_CAPSULE) ? scale_a.x : 0.0f;
            float rb = (tb == GEO_SPHERE || tb == GEO_CAPSULE) ? scale_b.x : 0.0f;
            Contacts4 k4;
            primitive_pair(ta, tb, Xa, Xb, scale_a, scale_b, gap_sum + margin_a + margin_b, k4);
            float total_sep = ra + rb + margin_a + margin_b;
            vec3 n = normalize(k4.normal);
            int ba = m.shape_body[sa], bb = m.shape_body[sb];
            xform Xbw_a = ba < 0 ? xform() : xform_inverse(c.body_q(ba));
            xform Xbw_b = bb < 0 ? xform() : xform_inverse(c.body_q(bb));
            float off_a = ra + margin_a, off_b = rb + margin_b;
            int gid_a = c.newton_shape_id(sa), gid_b = c.newton_shape_id(sb);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dist = k4.dist(k);
                if (!(dist < NT_MAXVAL)) continue;
                vec3 center = k4.pos(k);
                vec3 aw = center - n * (0.5f * dist + ra);
                vec3 bw = center + n * (0.5f * dist + rb);
                float d = dot(bw - aw, n) - total_sep;
                if (!(d <= gap_sum)) continue;
                int slot = p * cpp + nvalid;
                size_t gi = (size_t)slot * c.ES + c.env;
                ct.shape0[gi] = gid_a;
                ct.shape1[gi] = gid_b;
                float* D = ct.data;
                vec3 p0 = xform_point(Xbw_a, aw), p1 = xform_point(Xbw_b, bw);
                vec3 o0 = xform_vector(Xbw_a, off_a * n), o1 = xform_vector(Xbw_b, -off_b * n);
                D[c.g(CD_POINT0 + 0, ncs, slot)] = p0.x; D[c.g(CD_POINT0 + 1, ncs, slot)] = p0.y; D[c.g(CD_POINT0 + 2, ncs, slot)] = p0.z;
                D[c.g(CD_POINT1 + 0, ncs, slot)] = p1.x; D[c.g(CD_POINT1 + 1, ncs, slot)] = p1.y; D[c.g(CD_POINT1 + 2, ncs, slot)] = p1.z;
                D[c.g(CD_OFFSET0 + 0, ncs, slot)] = o0.x; D[c.g(CD_OFFSET0 + 1, ncs, slot)] = o0.y; D[c.g(CD_OFFSET0 + 2, ncs, slot)] = o0.z;
                D[c.g(CD_OFFSET1 + 0, ncs, slot)] = o1.x; D[c.g(CD_OFFSET1 + 1, ncs, slot)] = o1.y; D[c.g(CD_OFFSET1 + 2, ncs, slot)] = o1.z;
                D[c.g(CD_NORMAL + 0, ncs, slot)] = n.x; D[c.g(CD_NORMAL + 1, ncs, slot)] = n.y; D[c.g(CD_NORMAL + 2, ncs, slot)] = n.z;
                D[c.g(CD_MARGIN0, ncs, slot)] = off_a;
                D[c.g(CD_MARGIN1, ncs, slot)] = off_b;
                nvalid += 1;
            }
        }
        // pairs routed to the convex (MPR/GJK) path are handled by nt_convex (next round): no contacts yet
    }
    for (int k = nvalid; k < cpp; ++k) {
        size_t gi = (size_t)(p * cpp + k) * c.ES + c.env;
        ct.shape0[gi] = -1;
        ct.shape1[gi] = -1;
    }
    c.l(c.L.pc, 0, m.np, p) = (float)nvalid;
}
template <int EPB>
NT_DI void phase_pairs(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int p = c.slot; p < c.a.m.np; p += c.nslot) pair_item(c, p);
}

template <int EPB>
NT_DI void phase_contact_count(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    if (c.slot == 0 && c.valid) {
        int n = 0;
        for (int p = 0; p < m.np; ++p) n += (int)c.l(c.L.pc, 0, m.np, p);
        c.a.ct.env_count[c.env] = n;
    }
}

// ------------------------------------------------------------------------------------------------
// XPBD: apply_joint_forces (xpbd/kernels.py:945-1075)
// ------------------------------------------------------------------------------------------------
template <int EPB>
NT_DI void phase_joint_forces(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    const int nb = m.nb, nj = m.nj;
    // body threads seed body_f_tmp with state_in.body_f (solver_xpbd.py:423: wp.clone)
    if (!c.valid) return;
    for (int b = c.slot; b < nb; b += c.nslot) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c.l(c.L.bf, k, nb, b) = c.a.s_in.body_f[c.g(k, nb, b)];
    }
    for (int j = c.slot; j < nj; j += c.nslot) {
        vec3 fp, tp, fc, tc;  // parent wrench (subtracted), child wrench (added)
        int type = m.joint_type[j];
        if (m.joint_enabled[j] && type != JT_FIXED && type != JT_ROD) {
            int id_c = m.joint_child[j], id_p = m.joint_parent[j];
            xform X_pj = c.g_xform(m.joint_param, 0, nj, j);
            xform X_cj = c.g_xform(m.joint_param, 7, nj, j);
            xform X_wp = X_pj, pose_p = X_pj;
            vec3 com_p(0.0f);
            if (id_p >= 0) {
                pose_p = c.body_q(id_p);
                X_wp = pose_p * X_wp;
                com_p = c.com(id_p);
            }
            vec3 r_p = X_wp.p - xform_point(pose_p, com_p);
            xform pose_c = c.body_q(id_c);
            xform X_wc = pose_c * X_cj;
            vec3 r_c = X_wc.p - xform_point(pose_c, c.com(id_c));
            int qd_start = m.joint_qd_start[j];
            int lin = m.joint_lin_count[j], ang = m.joint_ang_count[j];
            const float* JF = c.a.c.joint_f;
            vec3 f_total, t_total;
            if (type == JT_FREE || type == JT_DISTANCE) {
                f_total = vec3(JF[c.g(0, 1, qd_start)], JF[c.g(0, 1, qd_start + 1)], JF[c.g(0, 1, qd_start + 2)]);
                t_total = vec3(JF[c.g(0, 1, qd_start + 3)], JF[c.g(0, 1, qd_start + 4)], JF[c.g(0, 1, qd_start + 5)]);
                fc = f_total; tc = t_total;
                fp = f_total; tp = t_total;
            } else {
                if (type == JT_BALL) {
                    t_total = vec3(JF[c.g(0, 1, qd_start)], JF[c.g(0, 1, qd_start + 1)], JF[c.g(0, 1, qd_start + 2)]);
                } else if (type == JT_REVOLUTE || type == JT_PRISMATIC || type == JT_D6) {
                    for (int k = 0; k < 3; ++k)
                        if (lin > k) {
                            vec3 axis = c.g_vec3(m.dof_param, DP_AXIS, m.nd, qd_start + k);
                            f_total += JF[c.g(0, 1, qd_start + k)] * xform_vector(X_wp, axis);
                        }
                    for (int k = 0; k < 3; ++k)
                        if (ang > k) {
                            vec3 axis = c.g_vec3(m.dof_param, DP_AXIS, m.nd, qd_start + lin + k);
                            t_total += JF[c.g(0, 1, qd_start + lin + k)] * xform_vector(X_wp, axis);
                        }
                }
                fc = f_total; tc = t_total + cross(r_c, f_total);
                fp = f_total; tp = t_total + cross(r_p, f_total);
            }
        }
        c.st_lds_vec3(c.L.jw, 0, nj, j, fp);
        c.st_lds_vec3(c.L.jw, 3, nj, j, tp);
        c.st_lds_vec3(c.L.jw, 6, nj, j, fc);
        c.st_lds_vec3(c.L.jw, 9, nj, j, tc);
    }
}

// body thread: fold joint wrenches into body_f_tmp in ascending-joint order, then integrate_bodies
// (solver.py:63-170)
template <int EPB>
NT_DI void integrate_item(const Ctx<EPB>& c, const int b) {
    const nt_model& m = c.a.m;
    const int nb = m.nb, nj = m.nj;
    vec3 f0 = c.lds_vec3(c.L.bf, 0, nb, b), t0 = c.lds_vec3(c.L.bf, 3, nb, b);
    for (int i = m.body_joint_start[b]; i < m.body_joint_start[b + 1]; ++i) {
        int code = m.body_joint_list[i];
        int j = code >> 1;
        if (code & 1) {
            f0 += c.lds_vec3(c.L.jw, 6, nj, j);
            t0 += c.lds_vec3(c.L.jw, 9, nj, j);
        } else {
            f0 -= c.lds_vec3(c.L.jw, 0, nj, j);
            t0 -= c.lds_vec3(c.L.jw, 3, nj, j);
        }
    }
    if (m.body_flags[b] & BODY_KINEMATIC) return;  // pass through unchanged

    xform q = c.body_q(b);
    spatial qd = c.body_qd(b);
    float inv_mass = m.body_param[c.g(BP_INV_MASS, nb, b)];
    mat33 inertia = c.g_mat33(m.body_param, BP_INERTIA, nb, b);
    mat33 inv_inertia = c.g_mat33(m.body_param, BP_INV_INERTIA, nb, b);
    vec3 com = c.com(b);
    vec3 gravity(m.gravity[c.g(0, 1, 0)], m.gravity[c.g(1, 1, 0)], m.gravity[c.g(2, 1, 0)]);
    const float dt = c.a.dt;

    vec3 x0 = q.p;
    quat r0 = q.q;
    vec3 w0 = qd.bottom, v0 = qd.top;
    vec3 x_com = x0 + quat_rotate(r0, com);
    vec3 v1 = v0 + (f0 * inv_mass + gravity * nonzero(inv_mass)) * dt;
    vec3 x1 = x_com + v1 * dt;
    vec3 wb = quat_rotate_inv(r0, w0);
    vec3 tb = quat_rotate_inv(r0, t0) - cross(wb, inertia * wb);
    vec3 w1 = quat_rotate(r0, wb + inv_inertia * tb * dt);
    quat r1 = normalize(r0 + quat(w1, 0.0f) * r0 * 0.5f * dt);
    w1 *= 1.0f - c.a.p.angular_damping * dt;
    c.st_lds_xform(c.L.bq, nb, b, xform(x1 - quat_rotate(r1, com), r1));
    c.st_lds_vec3(c.L.bqd, 0, nb, b, v1);
    c.st_lds_vec3(c.L.bqd, 3, nb, b, w1);
}
template <int EPB>
NT_DI void phase_integrate(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int b = c.slot; b < c.a.m.nb; b += c.nslot) integrate_item(c, b);
}

// ------------------------------------------------------------------------------------------------
// XPBD constraint helpers (xpbd/kernels.py:2047-2161)
// ------------------------------------------------------------------------------------------------
NT_DI float contact_constraint_delta(float err, quat qa, quat qb, float m_inv_a, float m_inv_b, const mat33& I_inv_a,
                                     const mat33& I_inv_b, vec3 lin_a, vec3 lin_b, vec3 ang_a, vec3 ang_b,
                                     float relaxation, float dt) {
    float denom = 0.0f;
    denom += length_sq(lin_a) * m_inv_a;
    denom += length_sq(lin_b) * m_inv_b;
    vec3 ra = quat_rotate_inv(qa, ang_a);
    vec3 rb = quat_rotate_inv(qb, ang_b);
    denom += dot(ra, I_inv_a * ra);
    denom += dot(rb, I_inv_b * rb);
    float delta_lambda = -err;
    if (denom > 0.0f) delta_lambda /= dt * denom;
    return delta_lambda * relaxation;
}

NT_DI float positional_correction(float err, float derr, quat qa, quat qb, float m_inv_a, float m_inv_b,
                                  const mat33& I_inv_a, const mat33& I_inv_b, vec3 lin_a, vec3 lin_b, vec3 ang_a,
                                  vec3 ang_b, float lambda_in, float compliance, float damping, float dt) {
    float denom = 0.0f;
    denom += length_sq(lin_a) * m_inv_a;
    denom += length_sq(lin_b) * m_inv_b;
    vec3 ra = quat_rotate_inv(qa, ang_a);
    vec3 rb = quat_rotate_inv(qb, ang_b);
    denom += dot(ra, I_inv_a * ra);
    denom += dot(rb, I_inv_b * rb);
    float alpha = compliance;
    float gamma = compliance * damping;
    float delta_lambda = -(err + alpha * lambda_in + gamma * derr);
    if (denom + alpha > 0.0f) delta_lambda /= (dt + gamma) * denom + alpha / dt;
    return delta_lambda;
}

NT_DI float angular_correction(float err, float derr, quat qa, quat qb, const mat33& I_inv_a, const mat33& I_inv_b,
                               vec3 ang_a, vec3 ang_b, float lambda_in, float compliance, float damping, float dt) {
    float denom = 0.0f;
    vec3 ra = quat_rotate_inv(qa, ang_a);
    vec3 rb = quat_rotate_inv(qb, ang_b);
    denom += dot(ra, I_inv_a * ra);
    denom += dot(rb, I_inv_b * rb);
    float alpha = compliance;
    float gamma = compliance * damping;
    float delta_lambda = -(err + alpha * lambda_in + gamma * derr);
    if (denom + alpha > 0.0f) delta_lambda /= (dt + gamma) * denom + alpha / dt;
    return delta_lambda;
}

// ------------------------------------------------------------------------------------------------
// XPBD: solve_body_contact_positions (xpbd/kernels.py:2164-2399); one thread per candidate pair walks the
// pair's contact slots in order and publishes the two bodies' summed corrections + active-contact counts.
// ------------------------------------------------------------------------------------------------
template <int EPB>
NT_DI void contacts_item(const Ctx<EPB>& c, const int p) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp;
    const float dt = c.a.dt, relaxation = c.a.p.rigid_contact_relaxation;
    const int pa = m.pair_a[p];
    vec3 dl0, da0, dl1, da1;  // side 0 = body of pair_a's shape, side 1 = body of pair_b's shape
    float cnt0 = 0.0f, cnt1 = 0.0f;
    const float* D = ct.data;

    for (int k = 0; k < cpp; ++k) {
        int slot = p * cpp + k;
        size_t gi = (size_t)slot * c.ES + c.env;
        int gid_a = ct.shape0[gi], gid_b = ct.shape1[gi];
        if (gid_a == gid_b) continue;
        int shape_a = gid_a >= 0 ? c.local_shape_id(gid_a) : -1;
        int shape_b = gid_b >= 0 ? c.local_shape_id(gid_b) : -1;
        int body_a = shape_a >= 0 ? m.shape_body[shape_a] : -1;
        int body_b = shape_b >= 0 ? m.shape_body[shape_b] : -1;
        if (body_a == body_b) continue;

        xform X_wb_a, X_wb_b;
        if (body_a >= 0) X_wb_a = c.body_q(body_a);
        if (body_b >= 0) X_wb_b = c.body_q(body_b);
        vec3 point0 = c.g_vec3(D, CD_POINT0, ncs, slot), point1 = c.g_vec3(D, CD_POINT1, ncs, slot);
        vec3 bx_a = xform_point(X_wb_a, point0);
        vec3 bx_b = xform_point(X_wb_b, point1);
        vec3 n = c.g_vec3(D, CD_NORMAL, ncs, slot);
        float d = dot(n, bx_b - bx_a) - (D[c.g(CD_MARGIN0, ncs, slot)] + D[c.g(CD_MARGIN1, ncs, slot)]);
        if (d >= 0.0f) continue;

        float m_inv_a = 0.0f, m_inv_b = 0.0f;
        mat33 I_inv_a, I_inv_b;
        vec3 com_a(0.0f), com_b(0.0f), omega_a(0.0f), omega_b(0.0f);
        if (body_a >= 0) {
            com_a = c.com(body_a);
            m_inv_a = c.inv_mass(body_a);
            I_inv_a = c.inv_inertia(body_a);
            omega_a = c.lds_vec3(c.L.bqd, 3, m.nb, body_a);
        }
        if (body_b >= 0) {
            com_b = c.com(body_b);
            m_inv_b = c.inv_mass(body_b);
            I_inv_b = c.inv_inertia(body_b);
            omega_b = c.lds_vec3(c.L.bqd, 3, m.nb, body_b);
        }
        int mat_nonzero = 0;
        float mu = 0.0f, mu_torsional = 0.0f, mu_rolling = 0.0f;
        if (shape_a >= 0) {
            mat_nonzero += 1;
            mu += c.shape_f(shape_a, SP_MU);
            mu_torsional += c.shape_f(shape_a, SP_MU_TORSIONAL);
            mu_rolling += c.shape_f(shape_a, SP_MU_ROLLING);
        }
        if (shape_b >= 0) {
            mat_nonzero += 1;
            mu += c.shape_f(shape_b, SP_MU);
            mu_torsional += c.shape_f(shape_b, SP_MU_TORSIONAL);
            mu_rolling += c.shape_f(shape_b, SP_MU_ROLLING);
        }
        if (mat_nonzero > 0) {
            mu /= float(mat_nonzero);
            mu_torsional /= float(mat_nonzero);
            mu_rolling /= float(mat_nonzero);
        }
        vec3 r_a = bx_a - xform_point(X_wb_a, com_a);
        vec3 r_b = bx_b - xform_point(X_wb_b, com_b);
        vec3 angular_a = -cross(r_a, n);
        vec3 angular_b = cross(r_b, n);

        float lambda_n = contact_constraint_delta(d, X_wb_a.q, X_wb_b.q, m_inv_a, m_inv_b, I_inv_a, I_inv_b, -n, n,
                                                  angular_a, angular_b, relaxation, dt);
        vec3 lin_delta_a = -n * lambda_n;
        vec3 lin_delta_b = n * lambda_n;
        vec3 ang_delta_a = angular_a * lambda_n;
        vec3 ang_delta_b = angular_b * lambda_n;

        if (mu > 0.0f) {
            vec3 offset_a = c.g_vec3(D, CD_OFFSET0, ncs, slot), offset_b = c.g_vec3(D, CD_OFFSET1, ncs, slot);
            bx_a = xform_point(X_wb_a, point0 + offset_a);
            bx_b = xform_point(X_wb_b, point1 + offset_b);
            vec3 delta = bx_b - bx_a;
            vec3 friction_delta = delta - dot(n, delta) * n;
            r_a = bx_a - xform_point(X_wb_a, com_a);
            r_b = bx_b - xform_point(X_wb_b, com_b);
            vec3 rel_v_kin_t(0.0f);
            if (body_a >= 0 && (m.body_flags[body_a] & BODY_KINEMATIC) != 0) {
                vec3 v_a = velocity_at_point(c.body_qd(body_a), r_a);
                rel_v_kin_t = rel_v_kin_t - (v_a - dot(n, v_a) * n);
            }
            if (body_b >= 0 && (m.body_flags[body_b] & BODY_KINEMATIC) != 0) {
                vec3 v_b = velocity_at_point(c.body_qd(body_b), r_b);
                rel_v_kin_t = rel_v_kin_t + (v_b - dot(n, v_b) * n);
            }
            friction_delta += rel_v_kin_t * dt;
            vec3 perp = normalize(friction_delta);
            angular_a = -cross(r_a, perp);
            angular_b = cross(r_b, perp);
            float err = length(friction_delta);
            if (err > 0.0f) {
                float lambda_fr = contact_constraint_delta(err, X_wb_a.q, X_wb_b.q, m_inv_a, m_inv_b, I_inv_a, I_inv_b,
                                                           -perp, perp, angular_a, angular_b, relaxation, dt);
                lambda_fr = fmaxw(lambda_fr, -lambda_n * mu);
                lin_delta_a -= perp * lambda_fr;
                lin_delta_b += perp * lambda_fr;
                ang_delta_a += angular_a * lambda_fr;
                ang_delta_b += angular_b * lambda_fr;
            }
        }
        vec3 delta_omega = omega_b - omega_a;
        if (mu_torsional > 0.0f) {
            float err = dot(delta_omega, n) * dt;
            if (fabsf(err) > 0.0f) {
                vec3 lin(0.0f);
                float lt = contact_constraint_delta(err, X_wb_a.q, X_wb_b.q, m_inv_a, m_inv_b, I_inv_a, I_inv_b, lin, lin,
                                                    -n, n, relaxation, dt);
                lt = clampf(lt, -lambda_n * mu_torsional, lambda_n * mu_torsional);
                ang_delta_a -= n * lt;
                ang_delta_b += n * lt;
            }
        }
        if (mu_rolling > 0.0f) {
            delta_omega -= dot(n, delta_omega) * n;
            float err = length(delta_omega) * dt;
            if (err > 0.0f) {
                vec3 lin(0.0f);
                vec3 roll_n = normalize(delta_omega);
                float lr = contact_constraint_delta(err, X_wb_a.q, X_wb_b.q, m_inv_a, m_inv_b, I_inv_a, I_inv_b, lin, lin,
                                                    -roll_n, roll_n, relaxation, dt);
                lr = fmaxw(lr, -lambda_n * mu_rolling);
                ang_delta_a -= roll_n * lr;
                ang_delta_b += roll_n * lr;
            }
        }
        // shape0 is the type-sorted first shape; map back to the pair's (a, b) sides
        if (shape_a == pa) {
            if (body_a >= 0) { dl0 += lin_delta_a; da0 += ang_delta_a; cnt0 += 1.0f; }
            if (body_b >= 0) { dl1 += lin_delta_b; da1 += ang_delta_b; cnt1 += 1.0f; }
        } else {
            if (body_a >= 0) { dl1 += lin_delta_a; da1 += ang_delta_a; cnt1 += 1.0f; }
            if (body_b >= 0) { dl0 += lin_delta_b; da0 += ang_delta_b; cnt0 += 1.0f; }
        }
    }
    c.st_lds_vec3(c.L.pw, 0, m.np, p, dl0);
    c.st_lds_vec3(c.L.pw, 3, m.np, p, da0);
    c.st_lds_vec3(c.L.pw, 6, m.np, p, dl1);
    c.st_lds_vec3(c.L.pw, 9, m.np, p, da1);
    c.l(c.L.pw, 12, m.np, p) = cnt0;
    c.l(c.L.pw, 13, m.np, p) = cnt1;
}
template <int EPB>
NT_DI void phase_contacts(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int p = c.slot; p < c.a.m.np; p += c.nslot) contacts_item(c, p);
}

// ------------------------------------------------------------------------------------------------
// XPBD: apply_body_deltas (xpbd/kernels.py:864-933).  FROM_PAIRS: sum pair corrections (+ contact counts),
// otherwise sum joint corrections.
// ------------------------------------------------------------------------------------------------
template <int EPB, bool FROM_PAIRS>
NT_DI void apply_item(const Ctx<EPB>& c, const int b) {
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    float inv_m = c.inv_mass(b);
    if (inv_m == 0.0f) return;  // pass-through

    vec3 dlin, dang;
    float inv_weight = 0.0f;
    if (FROM_PAIRS) {
        for (int i = m.body_pair_start[b]; i < m.body_pair_start[b + 1]; ++i) {
            int code = m.body_pair_list[i];
            int p = code >> 1, side = code & 1;
            dlin += c.lds_vec3(c.L.pw, side * 6, m.np, p);
            dang += c.lds_vec3(c.L.pw, side * 6 + 3, m.np, p);
            inv_weight += c.l(c.L.pw, 12 + side, m.np, p);
        }
    } else {
        for (int i = m.body_joint_start[b]; i < m.body_joint_start[b + 1]; ++i) {
            int code = m.body_joint_list[i];
            int j = code >> 1, side = code & 1;
            dlin += c.lds_vec3(c.L.jw, side * 6, m.nj, j);
            dang += c.lds_vec3(c.L.jw, side * 6 + 3, m.nj, j);
        }
    }
    mat33 inv_I = c.inv_inertia(b);
    mat33 body_I = c.g_mat33(m.body_param, BP_INERTIA, nb, b);
    xform tf = c.body_q(b);
    spatial qd = c.body_qd(b);
    const float dt = c.a.dt;
    vec3 v0 = qd.top, w0 = qd.bottom;
    vec3 p0 = tf.p;
    quat q0 = tf.q;
    float weight = 1.0f;
    if (FROM_PAIRS && c.a.p.rigid_contact_con_weighting) {
        if (inv_weight > 0.0f) weight = 1.0f / inv_weight;
    }
    vec3 dp = dlin * (inv_m * weight);
    vec3 dq = dang * weight;
    vec3 wb = quat_rotate_inv(q0, w0);
    vec3 dwb = inv_I * quat_rotate_inv(q0, dq);
    vec3 tb = cross(dwb, body_I * (wb + dwb)) + cross(wb, body_I * dwb);
    vec3 dw1 = quat_rotate(q0, dwb - (dt * inv_I) * tb);
    quat q1 = q0 + 0.5f * quat(dw1 * dt, 0.0f) * q0;
    q1 = normalize(q1);
    vec3 com = c.com(b);
    vec3 x_com = p0 + quat_rotate(q0, com);
    vec3 p1 = x_com + dp * dt;
    p1 -= quat_rotate(q1, com);
    c.st_lds_xform(c.L.bq, nb, b, xform(p1, q1));
    vec3 v1 = v0 + dp;
    vec3 w1 = w0 + dw1;
    if (length(v1) < 1e-4f) v1 = vec3(0.0f);
    if (length(w1) < 1e-4f) w1 = vec3(0.0f);
    c.st_lds_vec3(c.L.bqd, 0, nb, b, v1);
    c.st_lds_vec3(c.L.bqd, 3, nb, b, w1);
}
template <int EPB, bool FROM_PAIRS>
NT_DI void phase_apply(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int b = c.slot; b < c.a.m.nb; b += c.nslot) apply_item<EPB, FROM_PAIRS>(c, b);
}

// ------------------------------------------------------------------------------------------------
// XPBD: solve_body_joints (xpbd/kernels.py:1513-2044)
// ------------------------------------------------------------------------------------------------
struct AxisData {
    vec3 lower, upper, target_pos, stiffness, target_vel, damping;
};

template <int EPB>
NT_DI AxisData gather_axes(const Ctx<EPB>& c, int count, int axis_idx0, int target_idx0) {
    const nt_model& m = c.a.m;
    AxisData A;
    vec3 tp, ke_w, tv, kd_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (count > k) {
            int ai = axis_idx0 + k, ti = target_idx0 + k;
            vec3 axis = c.g_vec3(m.dof_param, DP_AXIS, m.nd, ai);
            float lower = m.dof_param[c.g(DP_LIMIT_LOWER, m.nd, ai)];
            float upper = m.dof_param[c.g(DP_LIMIT_UPPER, m.nd, ai)];
            vec3 lo_t = axis * lower, up_t = axis * upper;
            vec3 lo = vmin(lo_t, up_t), up = vmax(lo_t, up_t);
            if (k == 0) { A.lower = lo; A.upper = up; }
            else { A.lower = vmin(A.lower, lo); A.upper = vmax(A.upper, up); }
            float ke = m.dof_param[c.g(DP_TARGET_KE, m.nd, ai)];
            float kd = m.dof_param[c.g(DP_TARGET_KD, m.nd, ai)];
            float target_pos = c.a.c.joint_target_q[c.g(0, 1, ti)];
            float target_vel = c.a.c.joint_target_qd[c.g(0, 1, ai)];
            if (ke > 0.0f) {
                vec3 wa = axis * ke;
                tp += wa * target_pos;
                ke_w += vabs(wa);
            }
            if (kd > 0.0f) {
                vec3 wa = axis * kd;
                tv += wa * target_vel;
                kd_w += vabs(wa);
            }
        }
    }
    if (ke_w.x > 0.0f) tp.x /= ke_w.x;
    if (ke_w.y > 0.0f) tp.y /= ke_w.y;
    if (ke_w.z > 0.0f) tp.z /= ke_w.z;
    if (kd_w.x > 0.0f) tv.x /= kd_w.x;
    if (kd_w.y > 0.0f) tv.y /= kd_w.y;
    if (kd_w.z > 0.0f) tv.z /= kd_w.z;
    A.target_pos = tp; A.stiffness = ke_w; A.target_vel = tv; A.damping = kd_w;
    return A;
}

template <int EPB>
NT_DI void joints_item(const Ctx<EPB>& c, const int j) {
    const nt_model& m = c.a.m;
    const int nj = m.nj;
    const nt_xpbd_params& P = c.a.p;
    const float dt = c.a.dt;
    vec3 lin_delta_p, ang_delta_p, lin_delta_c, ang_delta_c;

    const int type = m.joint_type[j];
    bool active = m.joint_enabled[j] && type != JT_FREE;
    if (active) {
        int id_c = m.joint_child[j], id_p = m.joint_parent[j];
        xform X_pj = c.g_xform(m.joint_param, 0, nj, j);
        xform X_cj = c.g_xform(m.joint_param, 7, nj, j);
        xform X_wp = X_pj, pose_p = X_pj;
        float m_inv_p = 0.0f;
        mat33 I_inv_p;
        vec3 com_p(0.0f), vel_p(0.0f), omega_p(0.0f);
        if (id_p >= 0) {
            pose_p = c.body_q(id_p);
            X_wp = pose_p * X_wp;
            com_p = c.com(id_p);
            m_inv_p = c.inv_mass(id_p);
            I_inv_p = c.inv_inertia(id_p);
            spatial qd = c.body_qd(id_p);
            vel_p = qd.top;
            omega_p = qd.bottom;
        }
        xform pose_c = c.body_q(id_c);
        xform X_wc = pose_c * X_cj;
        vec3 com_c = c.com(id_c);
        float m_inv_c = c.inv_mass(id_c);
        mat33 I_inv_c = c.inv_inertia(id_c);
        spatial qdc = c.body_qd(id_c);
        vec3 vel_c = qdc.top, omega_c = qdc.bottom;

        if (!(m_inv_p == 0.0f && m_inv_c == 0.0f)) {
            xform rel_pose = xform_inverse(X_wp) * X_wc;
            vec3 rel_p = rel_pose.p;
            vec3 x_p = X_wp.p, x_c = X_wc.p;
            int axis_start = m.joint_qd_start[j];
            int target_axis_start = m.joint_tq_start[j];
            int lin_count = m.joint_lin_count[j], ang_count = m.joint_ang_count[j];
            vec3 world_com_p = xform_point(pose_p, com_p);
            vec3 world_com_c = xform_point(pose_c, com_c);
            bool early_out = false;

            if (type == JT_DISTANCE) {
                vec3 r_p = x_p - world_com_p, r_c = x_c - world_com_c;
                float lower = m.dof_param[c.g(DP_LIMIT_LOWER, m.nd, axis_start)];
                float upper = m.dof_param[c.g(DP_LIMIT_UPPER, m.nd, axis_start)];
                if (lower < 0.0f && upper < 0.0f) {
                    early_out = true;
                } else {
                    vec3 anchor_delta = x_c - x_p;
                    float d = length(anchor_delta);
                    float err = 0.0f;
                    if (lower >= 0.0f && d < lower) err = d - lower;
                    else if (upper >= 0.0f && d > upper) err = d - upper;
                    if (fabsf(err) > 1e-9f) {
                        vec3 linear_c;
                        if (d > 1e-9f) {
                            linear_c = anchor_delta / d;
                        } else {
                            vec3 com_delta = world_com_c - world_com_p;
                            if (length_sq(com_delta) > 1e-18f) linear_c = normalize(com_delta);
                            else linear_c = xform_vector(X_wp, vec3(1.0f, 0.0f, 0.0f));
                        }
                        vec3 linear_p = -linear_c;
                        vec3 angular_p = -cross(r_p, linear_c);
                        vec3 angular_c = cross(r_c, linear_c);
                        float derr = dot(linear_p, vel_p) + dot(linear_c, vel_c) + dot(angular_p, omega_p) + dot(angular_c, omega_c);
                        float compliance = P.joint_linear_compliance;
                        float ke = m.dof_param[c.g(DP_TARGET_KE, m.nd, axis_start)];
                        if (ke > 0.0f) compliance = 1.0f / ke;
                        float damping = m.dof_param[c.g(DP_TARGET_KD, m.nd, axis_start)];
                        float d_lambda = positional_correction(err, derr, pose_p.q, pose_c.q, m_inv_p, m_inv_c, I_inv_p, I_inv_c,
                                                               linear_p, linear_c, angular_p, angular_c, 0.0f, compliance, damping, dt);
                        lin_delta_p += linear_p * (d_lambda * P.joint_linear_relaxation);
                        ang_delta_p += angular_p * (d_lambda * P.joint_angular_relaxation);
                        lin_delta_c += linear_c * (d_lambda * P.joint_linear_relaxation);
                        ang_delta_c += angular_c * (d_lambda * P.joint_angular_relaxation);
                    }
                }
            } else {
                AxisData A = gather_axes(c, lin_count, axis_start, target_axis_start);
                vec3 projected_rel_p = rel_p;
#pragma unroll
                for (int dim = 0; dim < 3; ++dim) {
                    float lower = vget(A.lower, dim), upper = vget(A.upper, dim), r = vget(rel_p, dim);
                    if (r < lower) vset(projected_rel_p, dim, lower);
                    else if (r > upper) vset(projected_rel_p, dim, upper);
                    else if (vget(A.stiffness, dim) > 0.0f) vset(projected_rel_p, dim, clampf(vget(A.target_pos, dim), lower, upper));
                }
                mat33 frame_p = quat_to_matrix(X_wp.q);
                vec3 r_p = xform_point(X_wp, projected_rel_p) - world_com_p;
                vec3 r_c = x_c - world_com_c;
#pragma unroll
                for (int dim = 0; dim < 3; ++dim) {
                    float e = vget(rel_p, dim);
                    vec3 linear_c = mat_col(frame_p, dim);
                    vec3 linear_p = -linear_c;
                    vec3 angular_p = -cross(r_p, linear_c);
                    vec3 angular_c = cross(r_c, linear_c);
                    float derr = dot(linear_p, vel_p) + dot(linear_c, vel_c) + dot(angular_p, omega_p) + dot(angular_c, omega_c);
                    float err = 0.0f;
                    float compliance = P.joint_linear_compliance;
                    float damping = 0.0f;
                    float derr_rel = derr - vget(A.target_vel, dim);
                    float lower = vget(A.lower, dim), upper = vget(A.upper, dim);
                    if (e < lower) err = e - lower;
                    else if (e > upper) err = e - upper;
                    else {
                        float target_pos = clampf(vget(A.target_pos, dim), lower, upper);
                        float st = vget(A.stiffness, dim), dm = vget(A.damping, dim);
                        if (st > 0.0f) { err = e - target_pos; compliance = 1.0f / st; damping = dm; }
                        else if (dm > 0.0f) { compliance = 1.0f / dm; damping = dm; }
                    }
                    if (fabsf(err) > 1e-9f || fabsf(derr_rel) > 1e-9f) {
                        float d_lambda = positional_correction(err, derr_rel, pose_p.q, pose_c.q, m_inv_p, m_inv_c, I_inv_p, I_inv_c,
                                                               linear_p, linear_c, angular_p, angular_c, 0.0f, compliance, damping, dt);
                        lin_delta_p += linear_p * (d_lambda * P.joint_linear_relaxation);
                        ang_delta_p += angular_p * (d_lambda * P.joint_angular_relaxation);
                        lin_delta_c += linear_c * (d_lambda * P.joint_linear_relaxation);
                        ang_delta_c += angular_c * (d_lambda * P.joint_angular_relaxation);
                    }
                }
            }

            if (!early_out && (type == JT_FIXED || type == JT_PRISMATIC || type == JT_REVOLUTE || type == JT_D6)) {
                quat q_p = X_wp.q, q_c = X_wc.q;
                if (dot(q_p, q_c) < 0.0f) q_c = q_c * -1.0f;
                quat rel_q = quat_inverse(q_p) * q_c;
                quat qtwist = normalize(quat(rel_q.x, 0.0f, 0.0f, rel_q.w));
                quat qswing = rel_q * quat_inverse(qtwist);
                float s = __fsqrt_rn(rel_q.x * rel_q.x + rel_q.w * rel_q.w);
                float invs = 1.0f / s;
                float invscube = invs * invs * invs;
                float err_0 = 2.0f * asinf(clampf(qtwist.x, -1.0f, 1.0f));
                float err_1 = qswing.y, err_2 = qswing.z;
                quat grad_0(invs - rel_q.x * rel_q.x * invscube, 0.0f, 0.0f, -(rel_q.w * rel_q.x) * invscube);
                quat grad_1(-rel_q.w * (rel_q.w * rel_q.z + rel_q.x * rel_q.y) * invscube, rel_q.w * invs, -rel_q.x * invs,
                            rel_q.x * (rel_q.w * rel_q.z + rel_q.x * rel_q.y) * invscube);
                quat grad_2(rel_q.w * (rel_q.w * rel_q.y - rel_q.x * rel_q.z) * invscube, rel_q.x * invs, rel_q.w * invs,
                            rel_q.x * (rel_q.z * rel_q.x - rel_q.w * rel_q.y) * invscube);
                grad_0 = grad_0 * (2.0f / fabsf(qtwist.w));
                float swing_sq = qswing.w * qswing.w;
                if (swing_sq + 1.0e-4f < 1.0f) {
                    float d = __fsqrt_rn(1.0f - qswing.w * qswing.w);
                    float theta = 2.0f * acosf(clampf(qswing.w, -1.0f, 1.0f));
                    float scale = theta / d;
                    err_1 *= scale;
                    err_2 *= scale;
                    grad_1 = grad_1 * scale;
                    grad_2 = grad_2 * scale;
                }
                AxisData A = gather_axes(c, ang_count, axis_start + lin_count, target_axis_start + lin_count);
#pragma unroll
                for (int dim = 0; dim < 3; ++dim) {
                    float e = dim == 0 ? err_0 : (dim == 1 ? err_1 : err_2);
                    quat grad = dim == 0 ? grad_0 : (dim == 1 ? grad_1 : grad_2);
                    quat quat_c = 0.5f * q_p * grad * quat_inverse(q_c);
                    vec3 angular_c(quat_c.x, quat_c.y, quat_c.z);
                    vec3 angular_p = -angular_c;
                    float derr = dot(angular_p, omega_p) + dot(angular_c, omega_c);
                    float err = 0.0f;
                    float compliance = P.joint_angular_compliance;
                    float damping = 0.0f;
                    float derr_rel = derr - vget(A.target_vel, dim) * length(angular_c);
                    float lower = vget(A.lower, dim), upper = vget(A.upper, dim);
                    if (e < lower) err = e - lower;
                    else if (e > upper) err = e - upper;
                    else {
                        float target_pos = clampf(vget(A.target_pos, dim), lower, upper);
                        float st = vget(A.stiffness, dim), dm = vget(A.damping, dim);
                        if (st > 0.0f) { err = e - target_pos; compliance = 1.0f / st; damping = dm; }
                        else if (dm > 0.0f) { damping = dm; compliance = 1.0f / dm; }
                    }
                    float d_lambda = angular_correction(err, derr_rel, pose_p.q, pose_c.q, I_inv_p, I_inv_c, angular_p, angular_c,
                                                        0.0f, compliance, damping, dt) * P.joint_angular_relaxation;
                    ang_delta_p += angular_p * d_lambda;
                    ang_delta_c += angular_c * d_lambda;
                }
            }
            if (early_out) { lin_delta_p = vec3(); ang_delta_p = vec3(); lin_delta_c = vec3(); ang_delta_c = vec3(); }
        }
    }
    c.st_lds_vec3(c.L.jw, 0, nj, j, lin_delta_p);
    c.st_lds_vec3(c.L.jw, 3, nj, j, ang_delta_p);
    c.st_lds_vec3(c.L.jw, 6, nj, j, lin_delta_c);
    c.st_lds_vec3(c.L.jw, 9, nj, j, ang_delta_c);
}
template <int EPB>
NT_DI void phase_joints(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int j = c.slot; j < c.a.m.nj; j += c.nslot) joints_item(c, j);
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int EPB>
NT_DI void do_collide(const Ctx<EPB>& c) {
    phase_shapes(c);
    __syncthreads();
    phase_pairs(c);
    __syncthreads();
    phase_contact_count(c);
    __syncthreads();
}

// SolverXPBD.step control flow (solver_xpbd.py:329-862), rigid-only model
template <int EPB>
NT_DI void do_xpbd_step(const Ctx<EPB>& c) {
    const nt_model& m = c.a.m;
    phase_joint_forces(c);
    __syncthreads();
    phase_integrate(c);
    __syncthreads();
    for (int it = 0; it < c.a.p.iterations; ++it) {
        if (c.a.has_contacts) {
            phase_contacts(c);
            __syncthreads();
            phase_apply<EPB, true>(c);
            __syncthreads();
        }
        if (m.nj > 0) {
            phase_joints(c);
            __syncthreads();
            phase_apply<EPB, false>(c);
            __syncthreads();
        }
    }
}

template <int EPB>
__global__ void __launch_bounds__(EPB * 16 > 512 ? 1024 : 512) collide_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds);
    load_state(c, a.s_in);
    __syncthreads();
    do_collide(c);
}

template <int EPB>
__global__ void __launch_bounds__(EPB * 16 > 512 ? 1024 : 512) xpbd_step_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds);
    load_state(c, a.s_in);
    __syncthreads();
    do_xpbd_step(c);
    store_state(c, a.s_out);
}

// substeps x { clear_forces; collide; step; swap } with the state resident in LDS across substeps.
// Only the final state is stored (into s0 for an even number of substeps, s1 for odd, like the reference's
// pointer swap); body_f of both states is zeroed as clear_forces would leave it.
template <int EPB>
__global__ void __launch_bounds__(EPB * 16 > 512 ? 1024 : 512) xpbd_rollout_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds);
    const int nb = a.m.nb;
    load_state(c, a.s_in);
    if (c.valid)
        for (int b = c.slot; b < nb; b += c.nslot) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                a.s_in.body_f[c.g(k, nb, b)] = 0.0f;
                a.s_out.body_f[c.g(k, nb, b)] = 0.0f;
            }
        }
    __syncthreads();
    for (int s = 0; s < a.substeps; ++s) {
        do_collide(c);
        do_xpbd_step(c);
    }
    store_state(c, (a.substeps & 1) ? a.s_out : a.s_in);
}

__global__ void clear_forces_kernel(float* body_f, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) body_f[i] = 0.0f;
}

// AoS [E*nslot][ncomp] <-> SoA [ncomp][nslot][ES]
__global__ void pack_kernel(const float* __restrict__ aos, float* __restrict__ soa, int ncomp, int nslot, int E, int ES) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)ncomp * nslot * ES;
    if (i >= n) return;
    int env = i % ES;
    int s = (i / ES) % nslot;
    int comp = i / ((size_t)ES * nslot);
    soa[i] = env < E ? aos[((size_t)env * nslot + s) * ncomp + comp] : 0.0f;
}
__global__ void unpack_kernel(const float* __restrict__ soa, float* __restrict__ aos, int ncomp, int nslot, int E, int ES) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)ncomp * nslot * E;
    if (i >= n) return;
    int comp = i % ncomp;
    int s = (i / ncomp) % nslot;
    int env = i / ((size_t)ncomp * nslot);
    aos[i] = soa[((size_t)comp * nslot + s) * ES + env];
}

// contacts export: exclusive scan of per-env counts (single block), then scatter in (env, pair, k) order
__global__ void contacts_scan_kernel(const int32_t* env_count, int E, int32_t* scan, int32_t* out_count) {
    __shared__ int32_t part[1024];
    int t = threadIdx.x, T = blockDim.x;
    int per = (E + T - 1) / T;
    int beg = t * per, end = beg + per < E ? beg + per : E;
    int sum = 0;
    for (int i = beg; i < end; ++i) sum += env_count[i];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int i = 0; i < T; ++i) { int v = part[i]; part[i] = acc; acc += v; }
        out_count[0] = acc;
        scan[E] = acc;
    }
    __syncthreads();
    int acc = part[t];
    for (int i = beg; i < end; ++i) { scan[i] = acc; acc += env_count[i]; }
}

struct ExportArgs {
    nt_model m;
    nt_contacts c;
    int cap;
    const int32_t* scan;
    int32_t *shape0, *shape1;
    float *point0, *point1, *offset0, *offset1, *normal, *margin0, *margin1;
};
__global__ void contacts_export_kernel(ExportArgs a) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.m.env_count) return;
    const int ES = a.m.env_stride, ncs = a.m.np * a.m.cpp;
    int idx = a.scan[env];
    for (int slot = 0; slot < ncs; ++slot) {
        size_t gi = (size_t)slot * ES + env;
        int s0 = a.c.shape0[gi];
        if (s0 < 0) continue;
        if (idx < a.cap) {
            a.shape0[idx] = s0;
            a.shape1[idx] = a.c.shape1[gi];
            const float* D = a.c.data;
            auto ld = [&](int comp) { return D[((size_t)comp * ncs + slot) * ES + env]; };
            for (int k = 0; k < 3; ++k) {
                a.point0[3 * idx + k] = ld(CD_POINT0 + k);
                a.point1[3 * idx + k] = ld(CD_POINT1 + k);
                a.offset0[3 * idx + k] = ld(CD_OFFSET0 + k);
                a.offset1[3 * idx + k] = ld(CD_OFFSET1 + k);
                a.normal[3 * idx + k] = ld(CD_NORMAL + k);
            }
            a.margin0[idx] = ld(CD_MARGIN0);
            a.margin1[idx] = ld(CD_MARGIN1);
        }
        idx += 1;
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// slot-threads per env: enough for the widest per-env population, capped so a block stays <= 512 threads
// (<= 1024 at EPB 64); phases with more items than slot-threads loop.
int slots_for(const nt_model& m, int epb) {
    int want = imax(imax(m.nb, m.nj), imax(m.ns, m.np));
    int cap = (epb == 64 ? 1024 : 512) / epb;
    return want < cap ? want : cap;
}

int pick_epb(const nt_model& m, int requested) {
    LdsLayout L = make_layout(m.nb, m.nj, m.np, m.ns);
    auto fits = [&](int epb) { return (size_t)L.floats_per_env * 4 * epb <= 160 * 1024; };
    if (requested == 16 || requested == 32 || requested == 64) return fits(requested) ? requested : 0;
    // auto: the widest tile that still gives >= 2 workgroups per CU worth of blocks (256 CUs), else the smallest
    const int cands[3] = {64, 32, 16};
    for (int i = 0; i < 3; ++i) {
        int epb = cands[i];
        if (!fits(epb)) continue;
        int blocks = (m.env_count + epb - 1) / epb;
        if (blocks >= 512 || epb == 16) return epb;
    }
    for (int i = 2; i >= 0; --i)
        if (fits(cands[i])) return cands[i];
    return 0;
}

template <typename K>
nt_status launch(K kernel, KArgs a, int epb, hipStream_t stream) {
    LdsLayout L = make_layout(a.m.nb, a.m.nj, a.m.np, a.m.ns);
    int nslot = slots_for(a.m, epb);
    a.nslot = nslot;
    int threads = ((nslot * epb + 63) / 64) * 64;
    size_t lds_bytes = (size_t)L.floats_per_env * 4 * epb;
    int blocks = (a.m.env_count + epb - 1) / epb;
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

#define NT_DISPATCH_EPB(KERNEL, args, epb, stream)                                \
    ((epb) == 64 ? launch(KERNEL<64>, args, 64, stream)                           \
                 : ((epb) == 32 ? launch(KERNEL<32>, args, 32, stream) : launch(KERNEL<16>, args, 16, stream)))

bool model_ok(const nt_model* m) {
    return m && m->env_count > 0 && m->env_stride >= m->env_count && (m->env_stride % 64) == 0 && m->nb > 0 &&
           (m->cpp == 4 || m->cpp == 5);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* nt_error_string(nt_status s) {
    switch (s) {
        case NT_OK: return "ok";
        case NT_ERR_INVALID_ARG: return "invalid argument";
        case NT_ERR_LAUNCH: return "kernel launch failed";
        case NT_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown";
    }
}

const char* nt_build_info(void) { return "libnewton_hip gfx950 (CDNA4) fp32, -ffp-contract=off, built " __DATE__; }

int32_t nt_lds_bytes_per_env(const nt_model* m) {
    if (!m) return -1;
    return make_layout(m->nb, m->nj, m->np, m->ns).floats_per_env * 4;
}

nt_status nt_clear_forces(const nt_model* m, nt_state* s, void* stream) {
    if (!model_ok(m) || !s || !s->body_f) return NT_ERR_INVALID_ARG;
    size_t n = (size_t)6 * m->nb * m->env_stride;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(clear_forces_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->body_f, n);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_collide(const nt_model* m, const nt_state* s, nt_contacts* c, const nt_collide_params* p, void* stream) {
    if (!model_ok(m) || !s || !c || !s->body_q) return NT_ERR_INVALID_ARG;
    if (m->np == 0) return NT_OK;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s;
    a.ct = *c;
    int epb = pick_epb(*m, p ? p->envs_per_block : 0);
    if (!epb) return NT_ERR_UNSUPPORTED;
    return NT_DISPATCH_EPB(collide_kernel, a, epb, (hipStream_t)stream);
}

nt_status nt_xpbd_step(const nt_model* m, const nt_xpbd_params* p, nt_state* s_in, nt_state* s_out, const nt_control* ctrl,
                       const nt_contacts* c, float dt, int32_t envs_per_block, void* stream) {
    if (!model_ok(m) || !p || !s_in || !s_out || !ctrl) return NT_ERR_INVALID_ARG;
    if (p->enable_restitution) return NT_ERR_UNSUPPORTED;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s_in;
    a.s_out = *s_out;
    a.c = *ctrl;
    if (c) a.ct = *c;
    a.has_contacts = (c != nullptr && m->np > 0) ? 1 : 0;
    a.p = *p;
    a.dt = dt;
    int epb = pick_epb(*m, envs_per_block);
    if (!epb) return NT_ERR_UNSUPPORTED;
    return NT_DISPATCH_EPB(xpbd_step_kernel, a, epb, (hipStream_t)stream);
}

nt_status nt_xpbd_rollout(const nt_model* m, const nt_xpbd_params* p, const nt_collide_params* cp, nt_state* s0, nt_state* s1,
                          const nt_control* ctrl, nt_contacts* c, float dt, int32_t substeps, void* stream) {
    if (!model_ok(m) || !p || !s0 || !s1 || !ctrl || !c || substeps < 1) return NT_ERR_INVALID_ARG;
    if (p->enable_restitution) return NT_ERR_UNSUPPORTED;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s0;
    a.s_out = *s1;
    a.c = *ctrl;
    a.ct = *c;
    a.has_contacts = m->np > 0 ? 1 : 0;
    a.p = *p;
    a.dt = dt;
    a.substeps = substeps;
    int epb = pick_epb(*m, cp ? cp->envs_per_block : 0);
    if (!epb) return NT_ERR_UNSUPPORTED;
    return NT_DISPATCH_EPB(xpbd_rollout_kernel, a, epb, (hipStream_t)stream);
}

nt_status nt_semi_implicit_step(const nt_model*, const nt_semi_implicit_params*, nt_state*, nt_state*, const nt_control*,
                                const nt_contacts*, float, int32_t, void*) {
    return NT_ERR_UNSUPPORTED;
}

nt_status nt_eval_fk(const nt_model*, const float*, const float*, nt_state*, void*) { return NT_ERR_UNSUPPORTED; }

nt_status nt_pack_aos(const float* aos, float* soa, int32_t ncomp, int32_t nslot, int32_t env_count, int32_t env_stride,
                      void* stream) {
    if (!aos || !soa || ncomp <= 0 || nslot <= 0 || env_count <= 0 || env_stride < env_count) return NT_ERR_INVALID_ARG;
    size_t n = (size_t)ncomp * nslot * env_stride;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, aos, soa, ncomp,
                       nslot, env_count, env_stride);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_unpack_aos(const float* soa, float* aos, int32_t ncomp, int32_t nslot, int32_t env_count, int32_t env_stride,
                        void* stream) {
    if (!aos || !soa || ncomp <= 0 || nslot <= 0 || env_count <= 0 || env_stride < env_count) return NT_ERR_INVALID_ARG;
    size_t n = (size_t)ncomp * nslot * env_count;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, soa, aos, ncomp,
                       nslot, env_count, env_stride);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_contacts_export(const nt_model* m, const nt_contacts* c, int32_t cap, int32_t* out_count, int32_t* out_shape0,
                             int32_t* out_shape1, float* out_point0, float* out_point1, float* out_offset0,
                             float* out_offset1, float* out_normal, float* out_margin0, float* out_margin1,
                             int32_t* scan_tmp, void* stream) {
    if (!model_ok(m) || !c || !out_count || !scan_tmp) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(contacts_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, c->env_count, m->env_count, scan_tmp,
                       out_count);
    ExportArgs a;
    a.m = *m;
    a.c = *c;
    a.cap = cap;
    a.scan = scan_tmp;
    a.shape0 = out_shape0; a.shape1 = out_shape1;
    a.point0 = out_point0; a.point1 = out_point1;
    a.offset0 = out_offset0; a.offset1 = out_offset1;
    a.normal = out_normal; a.margin0 = out_margin0; a.margin1 = out_margin1;
    hipLaunchKernelGGL(contacts_export_kernel, dim3((m->env_count + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
