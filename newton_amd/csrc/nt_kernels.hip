// nt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of libnewton_hip.so.
//
// Design (DESIGN.md has the long form):
//  * env-major SoA in HBM: base[(comp * nslot + slot) * ES + env]; a wave reads consecutive envs of one
//    (component, slot) = one coalesced request.
//  * one workgroup owns EPB environments for a whole substep (or a whole rollout): thread -> (env, slot) with
//    env = blockIdx * EPB + tid % EPB and slot = tid / EPB.  A slot-thread plays, phase by phase, body `slot`,
//    shape `slot`, candidate pair `slot`, contact slot `slot`, and joint part `slot` ([0,nj) linear rows,
//    [nj,2nj) angular rows), so every constraint row of every environment is solved by its own lane and lanes of a
//    wave run the same code path.
//  * everything an environment needs during a substep is LDS-resident ([row][EPB], conflict-free because lanes of
//    a wave differ in env first): body state, per-body mass properties (the 3x3 inertia tiles), joint frames, dof
//    limits/gains, shape parameters and the control targets.  HBM is touched once per kernel for state/params
//    and once per substep for the Contacts boundary.
//  * constraint threads publish per-joint / per-contact corrections in LDS and the owning body thread sums
//    them in ascending joint / contact order through a CSR incidence list -- no float atomics, deterministic,
//    and the same order a serial ascending-tid Warp-CPU launch produces for wp.atomic_add.
//  * no MFMA: the largest dense object on this path is a 3x3 inertia.
//
// Reference behaviour (file:line under /root/reference) is cited per phase.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/newton_hip.h"
#include "nt_math.hpp"
#include "nt_primitives.hpp"
#include "nt_convex.hpp"

using namespace nt;

namespace {

enum JointType : int { JT_PRISMATIC = 0, JT_REVOLUTE = 1, JT_BALL = 2, JT_FIXED = 3, JT_FREE = 4, JT_DISTANCE = 5, JT_D6 = 6, JT_ROD = 7 };
constexpr int BODY_KINEMATIC = 2;

// body_param rows
constexpr int BP_COM = 0, BP_INV_MASS = 3, BP_INERTIA = 4, BP_INV_INERTIA = 13, BP_MASS = 22;
// dof_param rows
constexpr int DP_AXIS = 0, DP_LIMIT_LOWER = 3, DP_LIMIT_UPPER = 4, DP_TARGET_KE = 5, DP_TARGET_KD = 6, DP_LIMIT_KE = 7,
              DP_LIMIT_KD = 8, DP_ARMATURE = 9, DP_DAMPING = 10;
// shape_param rows
constexpr int SP_XFORM = 0, SP_SCALE = 7, SP_MARGIN = 10, SP_GAP = 11, SP_MU = 12, SP_MU_TORSIONAL = 13, SP_MU_ROLLING = 14,
              SP_KE = 15, SP_KD = 16, SP_KF = 17, SP_KA = 18, SP_RESTITUTION = 19;
// contact data rows
constexpr int CD_POINT0 = 0, CD_POINT1 = 3, CD_OFFSET0 = 6, CD_OFFSET1 = 9, CD_NORMAL = 12, CD_MARGIN0 = 15, CD_MARGIN1 = 16;
// per-contact correction record in LDS: lin_a, ang_a, lin_b, ang_b, has_a, has_b, shape0_is_pair_a
constexpr int CW_FLOATS = 15;

__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }

// LDS layout, in float rows per environment (each row is EPB floats wide)
struct LdsLayout {
    // persistent
    int bq, bqd;       // body_q [7][nb], body_qd [6][nb]
    int bp;            // body params [23][nb] (inverse mass / inertia already "effective": zero for kinematic bodies)
    int jp;            // joint params [14][nj]
    int dp;            // dof params [10][nd]
    int sp;            // shape params [19][ns]
    int cf, ctq, ctqd; // control: joint_f [nd], joint_target_q [ntq], joint_target_qd [nd]
    int grav;          // gravity [3]
    int bd;            // body-derived [9][nb]: world COM (3) + world-frame inverse inertia R I^-1 R^T (xx xy xz yy yz zz)
    int pm;            // live contacts per pair [np] (written by the collide phase, read by the fused solver phases)
    // scratch union
    int u;
    int sx, sa, pc;    // collide: shape world xform [7][ns], aabb [6][ns], per-pair contact count [np]
    int bf, jf;        // forces: body_f_tmp [6][nb], joint wrenches [12][nj]
    int jl, ja;        // joints: linear-part corrections [12][nj], angular-part child terms [9][nj]
    int cw;            // contacts: per-contact corrections [CW_FLOATS][np*cpp]
    int si_jf, si_cw;  // semi-implicit: joint wrenches + contact wrenches live together with body_f_tmp
    int xi;            // XPBD restitution: pre-step body_q / body_qd [13][nb], behind the XPBD scratch (solver_xpbd.py:414-416)
    int rows_per_env;
};

__host__ __device__ inline LdsLayout make_layout(const nt_model& m) {
    LdsLayout L;
    int o = 0;
    L.bq = o; o += 7 * m.nb;
    L.bqd = o; o += 6 * m.nb;
    L.bp = o; o += NT_BODY_PARAM_FLOATS * m.nb;
    L.jp = o; o += NT_JOINT_PARAM_FLOATS * m.nj;
    L.dp = o; o += NT_DOF_PARAM_FLOATS * m.nd;
    L.sp = o; o += NT_SHAPE_PARAM_FLOATS * m.ns;
    L.cf = o; o += m.nd;
    L.ctq = o; o += m.ntq;
    L.ctqd = o; o += m.nd;
    L.grav = o; o += 3;
    L.bd = o; o += 9 * m.nb;
    L.pm = o; o += m.np;
    L.u = o;
    L.sx = L.u; L.sa = L.sx + 7 * m.ns; L.pc = L.sa + 6 * m.ns;
    int coll = 13 * m.ns + m.np + 20 * (m.np - m.np_analytic);  // + manifold polygon scratch of the convex pairs
    L.bf = L.u; L.jf = L.bf + 6 * m.nb;
    int forces = 6 * m.nb + 12 * m.nj;
    L.jl = L.u; L.ja = L.jl + 12 * m.nj;
    int joints = 21 * m.nj;
    L.cw = L.u;
    int contacts = CW_FLOATS * m.np * m.cpp;
    L.si_jf = L.bf + 6 * m.nb; L.si_cw = L.si_jf + 12 * m.nj;
    int semi = 6 * m.nb + 12 * m.nj + contacts;
    int xpbd = imax(imax(coll, forces), imax(joints, contacts));
    L.xi = L.u + xpbd;
    L.rows_per_env = L.u + imax(xpbd + 13 * m.nb, semi);
    return L;
}

struct KArgs {
    nt_model m;
    nt_state s_in, s_out;
    nt_control c;
    nt_contacts ct;
    nt_xpbd_params p;
    nt_xpbd_report rep;  // optional reporting outputs of nt_xpbd_step (all NULL on the hot path)
    nt_semi_implicit_params sp;
    float angular_damping;  // integrate_bodies damping of the active solver
    float dt;
    int substeps;
    int has_contacts;
    int nslot;       // slot-threads per environment
    int debug_skip;  // ablation bitmask (NT_DEBUG_SKIP env var): 1 collide, 2 forces+integrate, 4 contacts, 8 joints, 16 apply
};

// env-uniform topology, staged once per workgroup into LDS (block-shared ints behind the per-env rows)
struct Topo {
    const int *body_flags, *joint_type, *joint_enabled, *joint_parent, *joint_child, *joint_q_start, *joint_qd_start,
        *joint_tq_start, *joint_lin_count, *joint_ang_count, *shape_body, *shape_type, *shape_flags, *shape_group, *pair_a,
        *pair_b, *body_joint_start, *body_joint_list, *body_pair_start, *body_pair_list, *shape_mesh_start, *shape_mesh_count, *gshape_id;
    const float* gshape;  // [ng][NT_SHAPE_PARAM_FLOATS] parameters of the global (world -1) shapes, block-shared copy
};
__host__ __device__ inline int topo_ints(const nt_model& m) {
    return m.nb + 9 * m.nj + 6 * (m.ns + m.ng) + 2 * m.np + 2 * (m.nb + 1) + 2 * m.nj + 2 * m.np + m.ng +
           NT_SHAPE_PARAM_FLOATS * m.ng;
}

template <int EPB>
struct Ctx {
    const KArgs& a;
    Topo T;
    float* lds;
    LdsLayout L;
    int e, slot, env, nslot;
    int ES;
    bool valid;

    // rows: float rows per env in front of the block-shared topology ints (-1: the XPBD / collide layout)
    NT_DI Ctx(const KArgs& a_, float* lds_, int rows = -1) : a(a_), lds(lds_) {
        L = make_layout(a.m);
        if (rows < 0) rows = L.rows_per_env;
        e = threadIdx.x % EPB;
        slot = threadIdx.x / EPB;
        nslot = a.nslot;
        env = blockIdx.x * EPB + e;
        ES = a.m.env_stride;
        valid = env < a.m.env_count && slot < nslot;
        const nt_model& m = a.m;
        int* ti = reinterpret_cast<int*>(lds + (size_t)rows * EPB);
        int o = 0;
        auto take = [&](const int*& dst, const int32_t* src, int n) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) ti[o + i] = src[i];
            dst = ti + o;
            o += n;
        };
        take(T.body_flags, m.body_flags, m.nb);
        take(T.joint_type, m.joint_type, m.nj);
        take(T.joint_enabled, m.joint_enabled, m.nj);
        take(T.joint_parent, m.joint_parent, m.nj);
        take(T.joint_child, m.joint_child, m.nj);
        take(T.joint_q_start, m.joint_q_start, m.nj);
        take(T.joint_qd_start, m.joint_qd_start, m.nj);
        take(T.joint_tq_start, m.joint_tq_start, m.nj);
        take(T.joint_lin_count, m.joint_lin_count, m.nj);
        take(T.joint_ang_count, m.joint_ang_count, m.nj);
        take(T.shape_body, m.shape_body, m.ns + m.ng);
        take(T.shape_type, m.shape_type, m.ns + m.ng);
        take(T.shape_flags, m.shape_flags, m.ns + m.ng);
        take(T.shape_group, m.shape_group, m.ns + m.ng);
        take(T.pair_a, m.pair_a, m.np);
        take(T.pair_b, m.pair_b, m.np);
        take(T.body_joint_start, m.body_joint_start, m.nb + 1);
        take(T.body_joint_list, m.body_joint_list, 2 * m.nj);  // padded to 2*nj entries by the host
        take(T.body_pair_start, m.body_pair_start, m.nb + 1);
        take(T.body_pair_list, m.body_pair_list, 2 * m.np);    // padded to 2*np entries by the host
        take(T.shape_mesh_start, m.shape_mesh_start, m.ns + m.ng);
        take(T.shape_mesh_count, m.shape_mesh_count, m.ns + m.ng);
        take(T.gshape_id, m.gshape_id, m.ng);
        {
            float* g = reinterpret_cast<float*>(ti + o);
            for (int i = threadIdx.x; i < NT_SHAPE_PARAM_FLOATS * m.ng; i += blockDim.x) g[i] = m.gshape_param[i];
            T.gshape = g;
            o += NT_SHAPE_PARAM_FLOATS * m.ng;
        }
    }
    // LDS element: row = field offset + comp * slots_in_field + slot
    NT_DI float& l(int off, int comp, int n, int s) const { return lds[(off + comp * n + s) * EPB + e]; }
    NT_DI size_t g(int comp, int n, int s) const { return (size_t)(comp * n + s) * ES + env; }

    NT_DI vec3 lv3(int off, int comp0, int n, int s) const {
        return vec3(l(off, comp0, n, s), l(off, comp0 + 1, n, s), l(off, comp0 + 2, n, s));
    }
    NT_DI void st_lv3(int off, int comp0, int n, int s, vec3 v) const {
        l(off, comp0, n, s) = v.x; l(off, comp0 + 1, n, s) = v.y; l(off, comp0 + 2, n, s) = v.z;
    }
    NT_DI xform lxf(int off, int comp0, int n, int s) const {
        return xform(lv3(off, comp0, n, s),
                     quat(l(off, comp0 + 3, n, s), l(off, comp0 + 4, n, s), l(off, comp0 + 5, n, s), l(off, comp0 + 6, n, s)));
    }
    NT_DI void st_lxf(int off, int n, int s, const xform& t) const {
        l(off, 0, n, s) = t.p.x; l(off, 1, n, s) = t.p.y; l(off, 2, n, s) = t.p.z;
        l(off, 3, n, s) = t.q.x; l(off, 4, n, s) = t.q.y; l(off, 5, n, s) = t.q.z; l(off, 6, n, s) = t.q.w;
    }
    NT_DI mat33 lm33(int off, int comp0, int n, int s) const {
        return mat33(l(off, comp0, n, s), l(off, comp0 + 1, n, s), l(off, comp0 + 2, n, s), l(off, comp0 + 3, n, s),
                     l(off, comp0 + 4, n, s), l(off, comp0 + 5, n, s), l(off, comp0 + 6, n, s), l(off, comp0 + 7, n, s),
                     l(off, comp0 + 8, n, s));
    }
    NT_DI vec3 gv3(const float* base, int comp0, int n, int s) const {
        return vec3(base[g(comp0, n, s)], base[g(comp0 + 1, n, s)], base[g(comp0 + 2, n, s)]);
    }

    NT_DI xform body_q(int b) const { return lxf(L.bq, 0, a.m.nb, b); }
    NT_DI quat body_rot(int b) const {
        const int nb = a.m.nb;
        return quat(l(L.bq, 3, nb, b), l(L.bq, 4, nb, b), l(L.bq, 5, nb, b), l(L.bq, 6, nb, b));
    }
    NT_DI vec3 body_v(int b) const { return lv3(L.bqd, 0, a.m.nb, b); }
    NT_DI vec3 body_w(int b) const { return lv3(L.bqd, 3, a.m.nb, b); }
    NT_DI float inv_mass(int b) const { return l(L.bp, BP_INV_MASS, a.m.nb, b); }
    NT_DI mat33 inv_inertia(int b) const { return lm33(L.bp, BP_INV_INERTIA, a.m.nb, b); }
    NT_DI mat33 inertia(int b) const { return lm33(L.bp, BP_INERTIA, a.m.nb, b); }
    NT_DI vec3 com(int b) const { return lv3(L.bp, BP_COM, a.m.nb, b); }
    NT_DI vec3 world_com(int b) const { return lv3(L.bd, 0, a.m.nb, b); }
    // a^T (R I^-1 R^T) a for body b (world-frame inverse inertia, symmetric 6-float tile in LDS)
    NT_DI float w_quad(int b, vec3 v) const {
        const int nb = a.m.nb;
        float xx = l(L.bd, 3, nb, b), xy = l(L.bd, 4, nb, b), xz = l(L.bd, 5, nb, b);
        float yy = l(L.bd, 6, nb, b), yz = l(L.bd, 7, nb, b), zz = l(L.bd, 8, nb, b);
        vec3 wv(xx * v.x + xy * v.y + xz * v.z, xy * v.x + yy * v.y + yz * v.z, xz * v.x + yz * v.y + zz * v.z);
        return dot(v, wv);
    }
    NT_DI void update_body_derived(int b) const {
        const int nb = a.m.nb;
        xform X = body_q(b);
        st_lv3(L.bd, 0, nb, b, xform_point(X, com(b)));
        mat33 R = quat_to_matrix(X.q);
        mat33 Ii = inv_inertia(b);
        // T = I^-1 R^T ; W = R T
        vec3 t0 = Ii * vec3(R.m00, R.m01, R.m02), t1 = Ii * vec3(R.m10, R.m11, R.m12), t2 = Ii * vec3(R.m20, R.m21, R.m22);
        vec3 r0(R.m00, R.m01, R.m02), r1(R.m10, R.m11, R.m12), r2(R.m20, R.m21, R.m22);
        l(L.bd, 3, nb, b) = dot(r0, t0); l(L.bd, 4, nb, b) = dot(r0, t1); l(L.bd, 5, nb, b) = dot(r0, t2);
        l(L.bd, 6, nb, b) = dot(r1, t1); l(L.bd, 7, nb, b) = dot(r1, t2); l(L.bd, 8, nb, b) = dot(r2, t2);
    }
    NT_DI float dof(int row, int d) const { return l(L.dp, row, a.m.nd, d); }
    NT_DI vec3 dof_axis(int d) const { return lv3(L.dp, DP_AXIS, a.m.nd, d); }

    // shape accessors: s < ns local (per-env params in LDS), otherwise the env-uniform global table
    NT_DI float shape_f(int s, int comp) const {
        if (s < a.m.ns) return l(L.sp, comp, a.m.ns, s);
        return T.gshape[(s - a.m.ns) * NT_SHAPE_PARAM_FLOATS + comp];
    }
    NT_DI vec3 shape_scale(int s) const { return vec3(shape_f(s, SP_SCALE), shape_f(s, SP_SCALE + 1), shape_f(s, SP_SCALE + 2)); }
    NT_DI xform shape_local_xform(int s) const {
        return xform(vec3(shape_f(s, 0), shape_f(s, 1), shape_f(s, 2)), quat(shape_f(s, 3), shape_f(s, 4), shape_f(s, 5), shape_f(s, 6)));
    }
    NT_DI int newton_shape_id(int s) const {  // flat Newton shape index
        return s < a.m.ns ? a.m.shape_local0 + env * a.m.ns + s : T.gshape_id[s - a.m.ns];
    }
    NT_DI int local_shape_id(int gid) const {
        int rel = gid - a.m.shape_local0 - env * a.m.ns;
        if (rel >= 0 && rel < a.m.ns) return rel;
        int g = 0;
        for (int k = 0; k < a.m.ng; ++k)
            if (T.gshape_id[k] == gid) g = k;
        return a.m.ns + g;
    }
};

// ------------------------------------------------------------------------------------------------
// HBM <-> LDS staging
// ------------------------------------------------------------------------------------------------
template <int EPB>
NT_DI void stage_rows(const Ctx<EPB>& c, int lds_off, const float* src, int rows) {
    for (int r = c.slot; r < rows; r += c.nslot) c.lds[(lds_off + r) * EPB + c.e] = src[(size_t)r * c.ES + c.env];
}
template <int EPB>
NT_DI void unstage_rows(const Ctx<EPB>& c, int lds_off, float* dst, int rows) {
    for (int r = c.slot; r < rows; r += c.nslot) dst[(size_t)r * c.ES + c.env] = c.lds[(lds_off + r) * EPB + c.e];
}

template <int EPB>
NT_DI void load_state(const Ctx<EPB>& c, const nt_state& s) {
    if (!c.valid) return;
    stage_rows(c, c.L.bq, s.body_q, 7 * c.a.m.nb);
    stage_rows(c, c.L.bqd, s.body_qd, 6 * c.a.m.nb);
}
template <int EPB>
NT_DI void store_state(const Ctx<EPB>& c, const nt_state& s) {
    if (!c.valid) return;
    unstage_rows(c, c.L.bq, s.body_q, 7 * c.a.m.nb);
    unstage_rows(c, c.L.bqd, s.body_qd, 6 * c.a.m.nb);
}
// parameters and controls: read once per kernel
template <int EPB>
NT_DI void load_params(const Ctx<EPB>& c, bool with_control) {
    if (!c.valid) return;
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    // body params, with effective (kinematic => 0) inverse mass / inertia (solver.py:173-187)
    for (int r = c.slot; r < NT_BODY_PARAM_FLOATS * nb; r += c.nslot) {
        int comp = r / nb, b = r - comp * nb;
        float v = m.body_param[(size_t)r * c.ES + c.env];
        bool inv_row = comp == BP_INV_MASS || (comp >= BP_INV_INERTIA && comp < BP_INV_INERTIA + 9);
        if (inv_row && (m.body_flags[b] & BODY_KINEMATIC)) v = 0.0f;  // (global copy: the LDS topology is not published yet)
        c.lds[(c.L.bp + r) * EPB + c.e] = v;
    }
    stage_rows(c, c.L.jp, m.joint_param, NT_JOINT_PARAM_FLOATS * m.nj);
    stage_rows(c, c.L.dp, m.dof_param, NT_DOF_PARAM_FLOATS * m.nd);
    stage_rows(c, c.L.sp, m.shape_param, NT_SHAPE_PARAM_FLOATS * m.ns);
    stage_rows(c, c.L.grav, m.gravity, 3);
    if (with_control) {
        stage_rows(c, c.L.cf, c.a.c.joint_f, m.nd);
        stage_rows(c, c.L.ctq, c.a.c.joint_target_q, m.ntq);
        stage_rows(c, c.L.ctqd, c.a.c.joint_target_qd, m.nd);
    }
}

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
        if (barrel >= hh && barrel > 0.0f) radius += (hh * hh) / (barrel + sqrtf(barrel * barrel - hh * hh));
        vec3 r0 = quat_rotate(q, vec3(1.0f, 0.0f, 0.0f));
        vec3 r1 = quat_rotate(q, vec3(0.0f, 1.0f, 0.0f));
        vec3 r2 = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
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
        vec3 r0 = quat_rotate(q, vec3(1.0f, 0.0f, 0.0f));
        vec3 r1 = quat_rotate(q, vec3(0.0f, 1.0f, 0.0f));
        vec3 r2 = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
        vec3 world_half(fabsf(r0.x) * half.x + fabsf(r1.x) * half.y + fabsf(r2.x) * half.z,
                        fabsf(r0.y) * half.x + fabsf(r1.y) * half.y + fabsf(r2.y) * half.z,
                        fabsf(r0.z) * half.x + fabsf(r1.z) * half.y + fabsf(r2.z) * half.z);
        lo = world_center - world_half - mv;
        hi = world_center + world_half + mv;
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
    } else {
        // finite planes: conservative bounding sphere (rejected by the host for collision)
        float r = 0.5f * sqrtf(scale.x * scale.x + scale.y * scale.y);
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

template <int EPB>
NT_DI void shape_world(const Ctx<EPB>& c, int s, xform& X, vec3& lo, vec3& hi) {
    const nt_model& m = c.a.m;
    if (s < m.ns) {
        X = c.lxf(c.L.sx, 0, m.ns, s);
        lo = c.lv3(c.L.sa, 0, m.ns, s);
        hi = c.lv3(c.L.sa, 3, m.ns, s);
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
    xform Xbw_a = ba < 0 ? xform() : xform_inverse(c.body_q(ba));
    xform Xbw_b = bb < 0 ? xform() : xform_inverse(c.body_q(bb));
    float off_a = ra + margin_a, off_b = rb + margin_b;
    vec3 aw = center - n * (0.5f * dist + ra);
    vec3 bw = center + n * (0.5f * dist + rb);
    size_t gi = (size_t)slot * c.ES + c.env;
    ct.shape0[gi] = c.newton_shape_id(sa);
    ct.shape1[gi] = c.newton_shape_id(sb);
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
}

template <int EPB, bool CVX>
NT_DI void collide_slot_item(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp;
    const int ncs = m.np * cpp;
    const int p = slot / cpp, k = slot - p * cpp;
    int sa = c.T.pair_a[p], sb = c.T.pair_b[p];
    xform Xa, Xb;
    vec3 loa, hia, lob, hib;
    shape_world(c, sa, Xa, loa, hia);
    shape_world(c, sb, Xb, lob, hib);
    bool hit = loa.x <= hib.x && hia.x >= lob.x && loa.y <= hib.y && hia.y >= lob.y && loa.z <= hib.z && hia.z >= lob.z;
    if (k == 0) ct.pair_hit[(size_t)p * c.ES + c.env] = hit ? 1 : 0;

    int nvalid = 0;
    bool wrote = false;
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
        if (!to_gjk) {
            float ra = (ta == GEO_SPHERE || ta == GEO_CAPSULE) ? scale_a.x : 0.0f;
            float rb = (tb == GEO_SPHERE || tb == GEO_CAPSULE) ? scale_b.x : 0.0f;
            Contacts4 k4;
            primitive_pair(ta, tb, Xa, Xb, scale_a, scale_b, gap_sum + margin_a + margin_b, k4);
            float total_sep = ra + rb + margin_a + margin_b;
            vec3 n = normalize(k4.normal);
            // admission test for all four candidates (contact_data.py:139-157); lane k keeps the k-th admitted one
            float my_dist = 0.0f;
            vec3 my_center;
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
                    if (ok && nvalid == k) { my_dist = dist; my_center = center; wrote = true; }
                }
                nvalid += ok ? 1 : 0;
            }
            if (wrote) write_contact_slot(c, slot, sa, sb, my_center, n, my_dist, ra, rb, margin_a, margin_b);
        }
        if constexpr (CVX) {
            // pairs [np_analytic, np): MPR/GJK + manifold. Lane k == 0 of the pair runs the whole (serial, divergent)
            // algorithm and fills the pair's slots in emission order; the other lanes of the pair leave them alone.
            if (p >= m.np_analytic) {
                if (k != 0) return;
                ConvexContacts cc;
                Geom ga, gb;
                ga.type = ta; ga.scale = scale_a;
                gb.type = tb; gb.scale = scale_b;
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
                // polygon scratch: 20 rows per convex pair in the part of the scratch union that the collide phases do not
                // use (behind shape transforms / AABBs / pair counts)
                PolyRef poly;
                poly.base = &c.lds[(c.L.pc + m.np + 20 * (p - m.np_analytic)) * EPB + c.e];
                poly.stride = EPB;
                convex_pair(ga, gb, Xa, Xb, margin_a, margin_b, gap_sum, lob, hib, poly, cc);
                float ra = (ta == GEO_SPHERE || ta == GEO_CAPSULE) ? scale_a.x : 0.0f;
                float rb = (tb == GEO_SPHERE || tb == GEO_CAPSULE) ? scale_b.x : 0.0f;
                vec3 n = normalize(cc.normal);
                nvalid = cc.count < cpp ? cc.count : cpp;
                for (int i = 0; i < cpp; ++i) {
                    if (i < nvalid) {
                        write_contact_slot(c, slot + i, sa, sb, cc.center(i), n, cc.distance(i), ra, rb, margin_a, margin_b);
                    } else {
                        size_t gi = (size_t)(slot + i) * c.ES + c.env;
                        ct.shape0[gi] = -1;
                        ct.shape1[gi] = -1;
                    }
                }
                c.l(c.L.pc, 0, m.np, p) = (float)nvalid;
                c.l(c.L.pm, 0, m.np, p) = (float)nvalid;
                return;
            }
        }
    }
    if (!wrote) {
        size_t gi = (size_t)slot * c.ES + c.env;
        ct.shape0[gi] = -1;
        ct.shape1[gi] = -1;
    }
    if (k == 0) {
        c.l(c.L.pc, 0, m.np, p) = (float)nvalid;
        c.l(c.L.pm, 0, m.np, p) = (float)nvalid;
    }
}
template <int EPB, bool CVX>
NT_DI void phase_pairs(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int ncs = c.a.m.np * c.a.m.cpp;
    for (int s = c.slot; s < ncs; s += c.nslot) collide_slot_item<EPB, CVX>(c, s);
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
NT_DI void phase_joint_forces(const Ctx<EPB>& c, bool forces_are_zero) {
    const nt_model& m = c.a.m;
    const int nb = m.nb, nj = m.nj;
    if (!c.valid) return;
    // body threads seed body_f_tmp with state_in.body_f (solver_xpbd.py:423: wp.clone)
    for (int r = c.slot; r < 6 * nb; r += c.nslot)
        c.lds[(c.L.bf + r) * EPB + c.e] = forces_are_zero ? 0.0f : c.a.s_in.body_f[(size_t)r * c.ES + c.env];
    for (int j = c.slot; j < nj; j += c.nslot) {
        vec3 fp, tp, fc, tc;  // parent wrench (subtracted), child wrench (added)
        int type = c.T.joint_type[j];
        if (c.T.joint_enabled[j] && type != JT_FIXED && type != JT_ROD) {
            int id_c = c.T.joint_child[j], id_p = c.T.joint_parent[j];
            xform X_pj = c.lxf(c.L.jp, 0, nj, j);
            xform X_cj = c.lxf(c.L.jp, 7, nj, j);
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
            int qd_start = c.T.joint_qd_start[j];
            int lin = c.T.joint_lin_count[j], ang = c.T.joint_ang_count[j];
            vec3 f_total, t_total;
            if (type == JT_FREE || type == JT_DISTANCE) {
                // joint_f rows qd_start .. qd_start+5 (n = 1 => comp is the row step)
                f_total = c.lv3(c.L.cf, 0, 1, qd_start);
                t_total = c.lv3(c.L.cf, 0, 1, qd_start + 3);
                fc = f_total; tc = t_total;
                fp = f_total; tp = t_total;
            } else {
                if (type == JT_BALL) {
                    t_total = c.lv3(c.L.cf, 0, 1, qd_start);
                } else if (type == JT_REVOLUTE || type == JT_PRISMATIC || type == JT_D6) {
                    for (int k = 0; k < 3; ++k)
                        if (lin > k) f_total += c.l(c.L.cf, 0, 1, qd_start + k) * xform_vector(X_wp, c.dof_axis(qd_start + k));
                    for (int k = 0; k < 3; ++k)
                        if (ang > k)
                            t_total += c.l(c.L.cf, 0, 1, qd_start + lin + k) * xform_vector(X_wp, c.dof_axis(qd_start + lin + k));
                }
                fc = f_total; tc = t_total + cross(r_c, f_total);
                fp = f_total; tp = t_total + cross(r_p, f_total);
            }
        }
        c.st_lv3(c.L.jf, 0, nj, j, fp);
        c.st_lv3(c.L.jf, 3, nj, j, tp);
        c.st_lv3(c.L.jf, 6, nj, j, fc);
        c.st_lv3(c.L.jf, 9, nj, j, tc);
    }
}

// body thread: fold joint wrenches into body_f_tmp in ascending-joint order, then integrate_bodies
// (solver.py:63-170)
// SEMI = false: XPBD (apply_joint_forces wrenches: parent subtracted, child added).
// SEMI = true : SolverSemiImplicit (eval_body_joints: parent added, child subtracted; then eval_body_contact: shape0's
//               body subtracted, shape1's body added), all in ascending joint / contact order.
template <int EPB, bool SEMI>
NT_DI void integrate_item(const Ctx<EPB>& c, const int b) {
    const nt_model& m = c.a.m;
    const int nb = m.nb, nj = m.nj;
    vec3 f0 = c.lv3(c.L.bf, 0, nb, b), t0 = c.lv3(c.L.bf, 3, nb, b);
    for (int i = c.T.body_joint_start[b]; i < c.T.body_joint_start[b + 1]; ++i) {
        int code = c.T.body_joint_list[i];
        int j = code >> 1;
        bool add = SEMI ? !(code & 1) : (code & 1);
        int row = (code & 1) ? 6 : 0;
        vec3 f = c.lv3(c.L.jf, row, nj, j), t = c.lv3(c.L.jf, row + 3, nj, j);
        if (add) { f0 += f; t0 += t; }
        else { f0 -= f; t0 -= t; }
    }
    if (SEMI && c.a.has_contacts) {
        const int cpp = m.cpp, ncs = m.np * cpp;
        for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
            int code = c.T.body_pair_list[i];
            int p = code >> 1, side = code & 1;
            for (int k = 0; k < cpp; ++k) {
                int slot = p * cpp + k;
                bool is_a = (side == 0) == (c.l(c.L.si_cw, 14, ncs, slot) != 0.0f);
                if (c.l(c.L.si_cw, is_a ? 12 : 13, ncs, slot) != 0.0f) {
                    vec3 f = c.lv3(c.L.si_cw, is_a ? 0 : 6, ncs, slot), t = c.lv3(c.L.si_cw, is_a ? 3 : 9, ncs, slot);
                    if (is_a) { f0 -= f; t0 -= t; }
                    else { f0 += f; t0 += t; }
                }
            }
        }
    }
    if (c.T.body_flags[b] & BODY_KINEMATIC) return;  // pass through unchanged

    xform q = c.body_q(b);
    vec3 v0 = c.body_v(b), w0 = c.body_w(b);
    // integrate_bodies uses the raw model inverse mass/inertia; for non-kinematic bodies raw == effective
    float inv_mass = c.inv_mass(b);
    mat33 inertia = c.inertia(b);
    mat33 inv_inertia = c.inv_inertia(b);
    vec3 com = c.com(b);
    vec3 gravity(c.lds[(c.L.grav + 0) * EPB + c.e], c.lds[(c.L.grav + 1) * EPB + c.e], c.lds[(c.L.grav + 2) * EPB + c.e]);
    const float dt = c.a.dt;

    vec3 x0 = q.p;
    quat r0 = q.q;
    vec3 x_com = x0 + quat_rotate(r0, com);
    vec3 v1 = v0 + (f0 * inv_mass + gravity * nonzero(inv_mass)) * dt;
    vec3 x1 = x_com + v1 * dt;
    vec3 wb = quat_rotate_inv(r0, w0);
    vec3 tb = quat_rotate_inv(r0, t0) - cross(wb, inertia * wb);
    vec3 w1 = quat_rotate(r0, wb + inv_inertia * tb * dt);
    quat r1 = normalize(r0 + quat(w1, 0.0f) * r0 * 0.5f * dt);
    w1 *= 1.0f - c.a.angular_damping * dt;
    c.st_lxf(c.L.bq, nb, b, xform(x1 - quat_rotate(r1, com), r1));
    c.st_lv3(c.L.bqd, 0, nb, b, v1);
    c.st_lv3(c.L.bqd, 3, nb, b, w1);
    c.update_body_derived(b);
}
template <int EPB>
NT_DI void phase_body_derived(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int b = c.slot; b < c.a.m.nb; b += c.nslot) c.update_body_derived(b);
}
template <int EPB, bool SEMI>
NT_DI void phase_integrate(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int b = c.slot; b < c.a.m.nb; b += c.nslot) integrate_item<EPB, SEMI>(c, b);
}

// ------------------------------------------------------------------------------------------------
// XPBD constraint helpers (xpbd/kernels.py:2047-2161)
// ------------------------------------------------------------------------------------------------
// Generalised inverse mass of a constraint row: sum |lin|^2 m^-1 + ang^T (R I^-1 R^T) ang.  The reference rotates `ang`
// into the body frame and applies the body-frame inverse inertia (xpbd/kernels.py:2064-2077); here the body thread has
// already rotated the inverse inertia into the world frame (Ctx::update_body_derived), which is the same quantity up to
// fp32 rounding and saves two quaternion rotations + a full 3x3 product per row.  wq_a / wq_b are those angular terms.
NT_DI float contact_constraint_delta(float err, float m_inv_a, float m_inv_b, vec3 lin_a, vec3 lin_b, float wq_a, float wq_b,
                                     float relaxation, float dt) {
    float denom = 0.0f;
    denom += length_sq(lin_a) * m_inv_a;
    denom += length_sq(lin_b) * m_inv_b;
    denom += wq_a;
    denom += wq_b;
    float delta_lambda = -err;
    if (denom > 0.0f) delta_lambda /= dt * denom;
    return delta_lambda * relaxation;
}

NT_DI float positional_correction(float err, float derr, float m_inv_a, float m_inv_b, vec3 lin_a, vec3 lin_b, float wq_a,
                                  float wq_b, float lambda_in, float compliance, float damping, float dt) {
    float denom = 0.0f;
    denom += length_sq(lin_a) * m_inv_a;
    denom += length_sq(lin_b) * m_inv_b;
    denom += wq_a;
    denom += wq_b;
    float alpha = compliance;
    float gamma = compliance * damping;
    float delta_lambda = -(err + alpha * lambda_in + gamma * derr);
    if (denom + alpha > 0.0f) delta_lambda /= (dt + gamma) * denom + alpha / dt;
    return delta_lambda;
}

NT_DI float angular_correction(float err, float derr, float wq_a, float wq_b, float lambda_in, float compliance,
                               float damping, float dt) {
    float denom = 0.0f;
    denom += wq_a;
    denom += wq_b;
    float alpha = compliance;
    float gamma = compliance * damping;
    float delta_lambda = -(err + alpha * lambda_in + gamma * derr);
    if (denom + alpha > 0.0f) delta_lambda /= (dt + gamma) * denom + alpha / dt;
    return delta_lambda;
}

// ------------------------------------------------------------------------------------------------
// XPBD: solve_body_contact_positions (xpbd/kernels.py:2164-2399); one lane per contact slot.
// ------------------------------------------------------------------------------------------------
// FUSED: the collide phase of the same kernel left the live-contact count of every pair in LDS, and the (type-sorted)
// shape order of a pair is static, so neither the liveness test nor the shape ids need the global contact arrays.
template <int EPB, bool FUSED>
NT_DI void contact_item(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp;
    const float dt = c.a.dt, relaxation = c.a.p.rigid_contact_relaxation;
    const float* D = ct.data;
    float has_a = 0.0f, has_b = 0.0f, a_is_pair_a = 1.0f;
    vec3 lin_delta_a, ang_delta_a, lin_delta_b, ang_delta_b;

    bool live;
    int shape_a = -1, shape_b = -1, body_a = -1, body_b = -1;
    if (FUSED) {
        const int p = slot / cpp, k = slot - p * cpp;
        live = k < (int)c.l(c.L.pm, 0, m.np, p);
        if (live) {
            shape_a = c.T.pair_a[p];
            shape_b = c.T.pair_b[p];
            if (c.T.shape_type[shape_a] > c.T.shape_type[shape_b]) {  // narrow_phase.py:525-528
                int t = shape_a; shape_a = shape_b; shape_b = t;
            }
        }
    } else {
        size_t gi = (size_t)slot * c.ES + c.env;
        int gid_a = ct.shape0[gi], gid_b = ct.shape1[gi];
        live = gid_a != gid_b;
        if (live) {
            shape_a = gid_a >= 0 ? c.local_shape_id(gid_a) : -1;
            shape_b = gid_b >= 0 ? c.local_shape_id(gid_b) : -1;
        }
    }
    if (live) {
        body_a = shape_a >= 0 ? c.T.shape_body[shape_a] : -1;
        body_b = shape_b >= 0 ? c.T.shape_body[shape_b] : -1;
        live = body_a != body_b;
    }
    if (live) {
        xform X_wb_a, X_wb_b;
        if (body_a >= 0) X_wb_a = c.body_q(body_a);
        if (body_b >= 0) X_wb_b = c.body_q(body_b);
        vec3 point0 = c.gv3(D, CD_POINT0, ncs, slot), point1 = c.gv3(D, CD_POINT1, ncs, slot);
        vec3 bx_a = xform_point(X_wb_a, point0);
        vec3 bx_b = xform_point(X_wb_b, point1);
        vec3 n = c.gv3(D, CD_NORMAL, ncs, slot);
        float d = dot(n, bx_b - bx_a) - (D[c.g(CD_MARGIN0, ncs, slot)] + D[c.g(CD_MARGIN1, ncs, slot)]);
        if (d < 0.0f) {
            float m_inv_a = 0.0f, m_inv_b = 0.0f;
            vec3 wc_a(0.0f), wc_b(0.0f), omega_a(0.0f), omega_b(0.0f);  // world COM (origin for static shapes)
            if (body_a >= 0) {
                wc_a = c.world_com(body_a);
                m_inv_a = c.inv_mass(body_a);
                omega_a = c.body_w(body_a);
            }
            if (body_b >= 0) {
                wc_b = c.world_com(body_b);
                m_inv_b = c.inv_mass(body_b);
                omega_b = c.body_w(body_b);
            }
            auto wq_a = [&](vec3 v) { return body_a >= 0 ? c.w_quad(body_a, v) : 0.0f; };
            auto wq_b = [&](vec3 v) { return body_b >= 0 ? c.w_quad(body_b, v) : 0.0f; };
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
            vec3 r_a = bx_a - wc_a;
            vec3 r_b = bx_b - wc_b;
            vec3 angular_a = -cross(r_a, n);
            vec3 angular_b = cross(r_b, n);

            float lambda_n = contact_constraint_delta(d, m_inv_a, m_inv_b, -n, n, wq_a(angular_a), wq_b(angular_b), relaxation, dt);
            lin_delta_a = -n * lambda_n;
            lin_delta_b = n * lambda_n;
            ang_delta_a = angular_a * lambda_n;
            ang_delta_b = angular_b * lambda_n;

            if (mu > 0.0f) {
                vec3 offset_a = c.gv3(D, CD_OFFSET0, ncs, slot), offset_b = c.gv3(D, CD_OFFSET1, ncs, slot);
                bx_a = xform_point(X_wb_a, point0 + offset_a);
                bx_b = xform_point(X_wb_b, point1 + offset_b);
                vec3 delta = bx_b - bx_a;
                vec3 friction_delta = delta - dot(n, delta) * n;
                r_a = bx_a - wc_a;
                r_b = bx_b - wc_b;
                vec3 rel_v_kin_t(0.0f);
                if (body_a >= 0 && (c.T.body_flags[body_a] & BODY_KINEMATIC) != 0) {
                    vec3 v_a = velocity_at_point(spatial(c.body_v(body_a), omega_a), r_a);
                    rel_v_kin_t = rel_v_kin_t - (v_a - dot(n, v_a) * n);
                }
                if (body_b >= 0 && (c.T.body_flags[body_b] & BODY_KINEMATIC) != 0) {
                    vec3 v_b = velocity_at_point(spatial(c.body_v(body_b), omega_b), r_b);
                    rel_v_kin_t = rel_v_kin_t + (v_b - dot(n, v_b) * n);
                }
                friction_delta += rel_v_kin_t * dt;
                vec3 perp = normalize(friction_delta);
                angular_a = -cross(r_a, perp);
                angular_b = cross(r_b, perp);
                float err = length(friction_delta);
                if (err > 0.0f) {
                    float lambda_fr = contact_constraint_delta(err, m_inv_a, m_inv_b, -perp, perp, wq_a(angular_a),
                                                               wq_b(angular_b), relaxation, dt);
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
                    float lt = contact_constraint_delta(err, m_inv_a, m_inv_b, lin, lin, wq_a(-n), wq_b(n), relaxation, dt);
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
                    float lr = contact_constraint_delta(err, m_inv_a, m_inv_b, lin, lin, wq_a(-roll_n), wq_b(roll_n), relaxation, dt);
                    lr = fmaxw(lr, -lambda_n * mu_rolling);
                    ang_delta_a -= roll_n * lr;
                    ang_delta_b += roll_n * lr;
                }
            }
            has_a = body_a >= 0 ? 1.0f : 0.0f;
            has_b = body_b >= 0 ? 1.0f : 0.0f;
            a_is_pair_a = (shape_a == c.T.pair_a[slot / cpp]) ? 1.0f : 0.0f;
        }
    }
    c.st_lv3(c.L.cw, 0, ncs, slot, lin_delta_a);
    c.st_lv3(c.L.cw, 3, ncs, slot, ang_delta_a);
    c.st_lv3(c.L.cw, 6, ncs, slot, lin_delta_b);
    c.st_lv3(c.L.cw, 9, ncs, slot, ang_delta_b);
    c.l(c.L.cw, 12, ncs, slot) = has_a;
    c.l(c.L.cw, 13, ncs, slot) = has_b;
    c.l(c.L.cw, 14, ncs, slot) = a_is_pair_a;
}
template <int EPB, bool FUSED>
NT_DI void phase_contacts(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int ncs = c.a.m.np * c.a.m.cpp;
    for (int s = c.slot; s < ncs; s += c.nslot) contact_item<EPB, FUSED>(c, s);
}

// ------------------------------------------------------------------------------------------------
// XPBD: apply_body_deltas (xpbd/kernels.py:864-933).  FROM_CONTACTS: sum contact corrections (+ contact counts)
// in ascending contact order; otherwise sum joint corrections in ascending joint order.
// ------------------------------------------------------------------------------------------------
template <int EPB, bool FROM_CONTACTS>
NT_DI void apply_item(const Ctx<EPB>& c, const int b) {
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    float inv_m = c.inv_mass(b);
    if (inv_m == 0.0f) return;  // pass-through

    vec3 dlin, dang;
    float inv_weight = 0.0f;
    if (FROM_CONTACTS) {
        const int cpp = m.cpp, ncs = m.np * cpp;
        for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
            int code = c.T.body_pair_list[i];
            int p = code >> 1, side = code & 1;  // side 0: this body owns pair_a's shape
            for (int k = 0; k < cpp; ++k) {
                int slot = p * cpp + k;
                // this body is the contact's "a" iff (side == 0) == (shape0 is pair_a's shape)
                bool is_a = (side == 0) == (c.l(c.L.cw, 14, ncs, slot) != 0.0f);
                float has = c.l(c.L.cw, is_a ? 12 : 13, ncs, slot);
                if (has != 0.0f) {
                    dlin += c.lv3(c.L.cw, is_a ? 0 : 6, ncs, slot);
                    dang += c.lv3(c.L.cw, is_a ? 3 : 9, ncs, slot);
                    inv_weight += 1.0f;
                }
            }
        }
    } else {
        const int nj = m.nj;
        for (int i = c.T.body_joint_start[b]; i < c.T.body_joint_start[b + 1]; ++i) {
            int code = c.T.body_joint_list[i];
            int j = code >> 1, side = code & 1;  // side 1: this body is the joint's child
            vec3 jl = c.lv3(c.L.jl, side * 6, nj, j);
            vec3 ja = c.lv3(c.L.jl, side * 6 + 3, nj, j);
            vec3 t0 = c.lv3(c.L.ja, 0, nj, j), t1 = c.lv3(c.L.ja, 3, nj, j), t2 = c.lv3(c.L.ja, 6, nj, j);
            if (side == 0) { t0 = -t0; t1 = -t1; t2 = -t2; }  // angular_p = -angular_c
            ja = ((ja + t0) + t1) + t2;
            dlin += jl;
            dang += ja;
        }
    }
    mat33 inv_I = c.inv_inertia(b);
    mat33 body_I = c.inertia(b);
    xform tf = c.body_q(b);
    vec3 v0 = c.body_v(b), w0 = c.body_w(b);
    const float dt = c.a.dt;
    vec3 p0 = tf.p;
    quat q0 = tf.q;
    float weight = 1.0f;
    if (FROM_CONTACTS && c.a.p.rigid_contact_con_weighting) {
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
    c.st_lxf(c.L.bq, nb, b, xform(p1, q1));
    vec3 v1 = v0 + dp;
    vec3 w1 = w0 + dw1;
    if (length(v1) < 1e-4f) v1 = vec3(0.0f);
    if (length(w1) < 1e-4f) w1 = vec3(0.0f);
    c.st_lv3(c.L.bqd, 0, nb, b, v1);
    c.st_lv3(c.L.bqd, 3, nb, b, w1);
    c.update_body_derived(b);
}
template <int EPB, bool FROM_CONTACTS>
NT_DI void phase_apply(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int b = c.slot; b < c.a.m.nb; b += c.nslot) apply_item<EPB, FROM_CONTACTS>(c, b);
}

// ------------------------------------------------------------------------------------------------
// XPBD: solve_body_joints (xpbd/kernels.py:1513-2044), split into a linear-rows lane and an angular-rows lane
// ------------------------------------------------------------------------------------------------
struct AxisData {
    vec3 lower, upper, target_pos, stiffness, target_vel, damping;
};

template <int EPB>
NT_DI AxisData gather_axes(const Ctx<EPB>& c, int count, int axis_idx0, int target_idx0) {
    AxisData A;
    vec3 tp, ke_w, tv, kd_w;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (count > k) {
            int ai = axis_idx0 + k, ti = target_idx0 + k;
            vec3 axis = c.dof_axis(ai);
            float lower = c.dof(DP_LIMIT_LOWER, ai);
            float upper = c.dof(DP_LIMIT_UPPER, ai);
            vec3 lo_t = axis * lower, up_t = axis * upper;
            vec3 lo = vmin(lo_t, up_t), up = vmax(lo_t, up_t);
            if (k == 0) { A.lower = lo; A.upper = up; }
            else { A.lower = vmin(A.lower, lo); A.upper = vmax(A.upper, up); }
            float ke = c.dof(DP_TARGET_KE, ai);
            float kd = c.dof(DP_TARGET_KD, ai);
            float target_pos = c.l(c.L.ctq, 0, 1, ti);
            float target_vel = c.l(c.L.ctqd, 0, 1, ai);
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

// true if the joint is solved at all (enabled, not FREE, not between two immovable bodies)
template <int EPB>
NT_DI bool joint_live(const Ctx<EPB>& c, int j, int& id_p, int& id_c, float& m_inv_p, float& m_inv_c) {
    const nt_model& m = c.a.m;
    const int type = c.T.joint_type[j];
    if (!c.T.joint_enabled[j] || type == JT_FREE) return false;
    id_c = c.T.joint_child[j];
    id_p = c.T.joint_parent[j];
    m_inv_p = id_p >= 0 ? c.inv_mass(id_p) : 0.0f;
    m_inv_c = c.inv_mass(id_c);
    return !(m_inv_p == 0.0f && m_inv_c == 0.0f);
}

template <int EPB>
NT_DI void joint_linear_item(const Ctx<EPB>& c, const int j) {
    const nt_model& m = c.a.m;
    const int nj = m.nj;
    const nt_xpbd_params& P = c.a.p;
    const float dt = c.a.dt;
    vec3 lin_delta_p, ang_delta_p, lin_delta_c, ang_delta_c;
    int id_p, id_c;
    float m_inv_p, m_inv_c;
    if (joint_live(c, j, id_p, id_c, m_inv_p, m_inv_c)) {
        const int type = c.T.joint_type[j];
        xform X_pj = c.lxf(c.L.jp, 0, nj, j);
        xform X_cj = c.lxf(c.L.jp, 7, nj, j);
        xform X_wp = X_pj;
        vec3 world_com_p = X_pj.p;  // transform_point(pose_p = X_pj, com_p = 0) for world-attached joints
        vec3 vel_p(0.0f), omega_p(0.0f);
        if (id_p >= 0) {
            X_wp = c.body_q(id_p) * X_wp;
            world_com_p = c.world_com(id_p);
            vel_p = c.body_v(id_p);
            omega_p = c.body_w(id_p);
        }
        xform X_wc = c.body_q(id_c) * X_cj;
        vec3 world_com_c = c.world_com(id_c);
        vec3 vel_c = c.body_v(id_c), omega_c = c.body_w(id_c);
        auto wq_p = [&](vec3 v) { return id_p >= 0 ? c.w_quad(id_p, v) : 0.0f; };
        auto wq_c = [&](vec3 v) { return c.w_quad(id_c, v); };

        xform rel_pose = xform_inverse(X_wp) * X_wc;
        vec3 rel_p = rel_pose.p;
        vec3 x_p = X_wp.p, x_c = X_wc.p;
        int axis_start = c.T.joint_qd_start[j];
        int target_axis_start = c.T.joint_tq_start[j];
        int lin_count = c.T.joint_lin_count[j];

        if (type == JT_DISTANCE) {
            vec3 r_p = x_p - world_com_p, r_c = x_c - world_com_c;
            float lower = c.dof(DP_LIMIT_LOWER, axis_start);
            float upper = c.dof(DP_LIMIT_UPPER, axis_start);
            if (!(lower < 0.0f && upper < 0.0f)) {
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
                    float ke = c.dof(DP_TARGET_KE, axis_start);
                    if (ke > 0.0f) compliance = 1.0f / ke;
                    float damping = c.dof(DP_TARGET_KD, axis_start);
                    float d_lambda = positional_correction(err, derr, m_inv_p, m_inv_c, linear_p, linear_c, wq_p(angular_p),
                                                           wq_c(angular_c), 0.0f, compliance, damping, dt);
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
                    float d_lambda = positional_correction(err, derr_rel, m_inv_p, m_inv_c, linear_p, linear_c, wq_p(angular_p),
                                                           wq_c(angular_c), 0.0f, compliance, damping, dt);
                    lin_delta_p += linear_p * (d_lambda * P.joint_linear_relaxation);
                    ang_delta_p += angular_p * (d_lambda * P.joint_angular_relaxation);
                    lin_delta_c += linear_c * (d_lambda * P.joint_linear_relaxation);
                    ang_delta_c += angular_c * (d_lambda * P.joint_angular_relaxation);
                }
            }
        }
    }
    c.st_lv3(c.L.jl, 0, nj, j, lin_delta_p);
    c.st_lv3(c.L.jl, 3, nj, j, ang_delta_p);
    c.st_lv3(c.L.jl, 6, nj, j, lin_delta_c);
    c.st_lv3(c.L.jl, 9, nj, j, ang_delta_c);
}

template <int EPB>
NT_DI void joint_angular_item(const Ctx<EPB>& c, const int j) {
    const nt_model& m = c.a.m;
    const int nj = m.nj;
    const nt_xpbd_params& P = c.a.p;
    const float dt = c.a.dt;
    vec3 t0, t1, t2;  // angular_c * d_lambda for the three angular rows (parent gets the negation)
    int id_p, id_c;
    float m_inv_p, m_inv_c;
    const int type = c.T.joint_type[j];
    bool angular_type = type == JT_FIXED || type == JT_PRISMATIC || type == JT_REVOLUTE || type == JT_D6;
    if (angular_type && joint_live(c, j, id_p, id_c, m_inv_p, m_inv_c)) {
        xform X_pj = c.lxf(c.L.jp, 0, nj, j);
        xform X_cj = c.lxf(c.L.jp, 7, nj, j);
        quat q_p = X_pj.q;
        vec3 omega_p(0.0f);
        if (id_p >= 0) {
            q_p = c.body_rot(id_p) * X_pj.q;
            omega_p = c.body_w(id_p);
        }
        quat q_c = c.body_rot(id_c) * X_cj.q;
        vec3 omega_c = c.body_w(id_c);
        int axis_start = c.T.joint_qd_start[j];
        int target_axis_start = c.T.joint_tq_start[j];
        int lin_count = c.T.joint_lin_count[j], ang_count = c.T.joint_ang_count[j];

        if (dot(q_p, q_c) < 0.0f) q_c = q_c * -1.0f;
        quat rel_q = quat_inverse(q_p) * q_c;
        quat qtwist = normalize(quat(rel_q.x, 0.0f, 0.0f, rel_q.w));
        quat qswing = rel_q * quat_inverse(qtwist);
        float s = sqrtf(rel_q.x * rel_q.x + rel_q.w * rel_q.w);
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
            float d = sqrtf(1.0f - qswing.w * qswing.w);
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
            float wqp = id_p >= 0 ? c.w_quad(id_p, angular_p) : 0.0f;
            float d_lambda = angular_correction(err, derr_rel, wqp, c.w_quad(id_c, angular_c), 0.0f, compliance, damping, dt) *
                             P.joint_angular_relaxation;
            vec3 t = angular_c * d_lambda;
            if (dim == 0) t0 = t;
            else if (dim == 1) t1 = t;
            else t2 = t;
        }
    }
    c.st_lv3(c.L.ja, 0, nj, j, t0);
    c.st_lv3(c.L.ja, 3, nj, j, t1);
    c.st_lv3(c.L.ja, 6, nj, j, t2);
}

// apply_rigid_restitution (xpbd/kernels.py:2583-2728) for one contact slot; velocity deltas go to the per-contact record
template <int EPB>
NT_DI void restitution_item(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp, nb = m.nb;
    const float dt = c.a.dt;
    const float* D = ct.data;
    float has_a = 0.0f, has_b = 0.0f, a_is_pair_a = 1.0f;
    vec3 lin_a, ang_a, lin_b, ang_b;
    size_t gi = (size_t)slot * c.ES + c.env;
    int gid_a = ct.shape0[gi], gid_b = ct.shape1[gi];
    if (gid_a != gid_b) {
        int shape_a = gid_a >= 0 ? c.local_shape_id(gid_a) : -1;
        int shape_b = gid_b >= 0 ? c.local_shape_id(gid_b) : -1;
        int body_a = -1, body_b = -1, mat_nonzero = 0;
        float restitution = 0.0f;
        if (shape_a >= 0) {
            mat_nonzero += 1;
            restitution += c.shape_f(shape_a, SP_RESTITUTION);
            body_a = c.T.shape_body[shape_a];
        }
        if (shape_b >= 0) {
            mat_nonzero += 1;
            restitution += c.shape_f(shape_b, SP_RESTITUTION);
            body_b = c.T.shape_body[shape_b];
        }
        if (mat_nonzero > 0) restitution /= float(mat_nonzero);
        if (body_a != body_b) {
            float m_inv_a = 0.0f, m_inv_b = 0.0f;
            mat33 I_inv_a, I_inv_b;
            xform X_a_prev, X_b_prev;
            vec3 com_a(0.0f), com_b(0.0f);
            auto prev_q = [&](int b) { return c.lxf(c.L.xi, 0, nb, b); };
            auto prev_qd = [&](int b) { return spatial(c.lv3(c.L.xi, 7, nb, b), c.lv3(c.L.xi, 10, nb, b)); };
            if (body_a >= 0) {
                X_a_prev = prev_q(body_a);
                m_inv_a = c.inv_mass(body_a);
                I_inv_a = c.inv_inertia(body_a);
                com_a = c.com(body_a);
            }
            if (body_b >= 0) {
                X_b_prev = prev_q(body_b);
                m_inv_b = c.inv_mass(body_b);
                I_inv_b = c.inv_inertia(body_b);
                com_b = c.com(body_b);
            }
            vec3 bx_a = xform_point(X_a_prev, c.gv3(D, CD_POINT0, ncs, slot) + c.gv3(D, CD_OFFSET0, ncs, slot));
            vec3 bx_b = xform_point(X_b_prev, c.gv3(D, CD_POINT1, ncs, slot) + c.gv3(D, CD_OFFSET1, ncs, slot));
            vec3 n = c.gv3(D, CD_NORMAL, ncs, slot);
            float d = dot(n, bx_b - bx_a);
            if (d < 0.0f) {
                vec3 r_a = bx_a - xform_point(X_a_prev, com_a);
                vec3 r_b = bx_b - xform_point(X_b_prev, com_b);
                vec3 gravity(c.lds[(c.L.grav + 0) * EPB + c.e], c.lds[(c.L.grav + 1) * EPB + c.e], c.lds[(c.L.grav + 2) * EPB + c.e]);
                vec3 rxn_a(0.0f), rxn_b(0.0f), v_a(0.0f), v_b(0.0f), v_a_new(0.0f), v_b_new(0.0f);
                float inv_mass = 0.0f;
                if (body_a >= 0) {
                    v_a = velocity_at_point(prev_qd(body_a), r_a) + gravity * dt;
                    v_a_new = velocity_at_point(spatial(c.body_v(body_a), c.body_w(body_a)), r_a);
                    rxn_a = quat_rotate_inv(X_a_prev.q, cross(r_a, n));
                    inv_mass += m_inv_a + dot(rxn_a, I_inv_a * rxn_a);
                }
                if (body_b >= 0) {
                    v_b = velocity_at_point(prev_qd(body_b), r_b) + gravity * dt;
                    v_b_new = velocity_at_point(spatial(c.body_v(body_b), c.body_w(body_b)), r_b);
                    rxn_b = quat_rotate_inv(X_b_prev.q, cross(r_b, n));
                    inv_mass += m_inv_b + dot(rxn_b, I_inv_b * rxn_b);
                }
                float rel_vel_old = dot(n, v_b - v_a);
                float rel_vel_new = dot(n, v_b_new - v_a_new);
                if (inv_mass != 0.0f && rel_vel_old < 0.0f) {
                    float dv = (-rel_vel_new - restitution * rel_vel_old) / inv_mass;
                    if (body_a >= 0) {
                        float dv_a = -dv;
                        lin_a = n * m_inv_a * dv_a;
                        ang_a = quat_rotate(X_a_prev.q, I_inv_a * rxn_a * dv_a);
                        has_a = 1.0f;
                    }
                    if (body_b >= 0) {
                        lin_b = n * m_inv_b * dv;
                        ang_b = quat_rotate(X_b_prev.q, I_inv_b * rxn_b * dv);
                        has_b = 1.0f;
                    }
                    a_is_pair_a = (shape_a == c.T.pair_a[slot / cpp]) ? 1.0f : 0.0f;
                }
            }
        }
    }
    c.st_lv3(c.L.cw, 0, ncs, slot, lin_a);
    c.st_lv3(c.L.cw, 3, ncs, slot, ang_a);
    c.st_lv3(c.L.cw, 6, ncs, slot, lin_b);
    c.st_lv3(c.L.cw, 9, ncs, slot, ang_b);
    c.l(c.L.cw, 12, ncs, slot) = has_a;
    c.l(c.L.cw, 13, ncs, slot) = has_b;
    c.l(c.L.cw, 14, ncs, slot) = a_is_pair_a;
}
// apply_body_delta_velocities (xpbd/kernels.py:936-942): body lane sums its contacts' velocity deltas in contact order
template <int EPB>
NT_DI void restitution_apply_item(const Ctx<EPB>& c, const int b) {
    const nt_model& m = c.a.m;
    const int cpp = m.cpp, ncs = m.np * cpp, nb = m.nb;
    vec3 dv, dw;
    for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
        int code = c.T.body_pair_list[i];
        int p = code >> 1, side = code & 1;
        for (int k = 0; k < cpp; ++k) {
            int slot = p * cpp + k;
            bool is_a = (side == 0) == (c.l(c.L.cw, 14, ncs, slot) != 0.0f);
            if (c.l(c.L.cw, is_a ? 12 : 13, ncs, slot) != 0.0f) {
                dv += c.lv3(c.L.cw, is_a ? 0 : 6, ncs, slot);
                dw += c.lv3(c.L.cw, is_a ? 3 : 9, ncs, slot);
            }
        }
    }
    c.st_lv3(c.L.bqd, 0, nb, b, c.body_v(b) + dv);
    c.st_lv3(c.L.bqd, 3, nb, b, c.body_w(b) + dw);
}

// ------------------------------------------------------------------------------------------------
// optional reporting (never compiled into the fused rollout): per-joint child-side impulse -> State.body_parent_f
// (xpbd/kernels.py:1018-1019,1074-1075,2043-2044,2497-2544) and per-contact weighted impulse -> Contacts.force
// (xpbd/kernels.py:2398-2461).  Accumulators live in HBM (env-major SoA); lane <-> item mapping is the same in every
// phase, so a lane only ever re-reads its own partial sums.
// ------------------------------------------------------------------------------------------------
// after phase_joint_forces: joint_impulse[j] = child_wrench_at_com * dt (initialises the accumulator)
template <int EPB>
NT_DI void report_joint_forces(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int nj = c.a.m.nj;
    float* J = c.a.rep.joint_impulse;
    const float dt = c.a.dt;
    for (int j = c.slot; j < nj; j += c.nslot) {
        vec3 fc = c.lv3(c.L.jf, 6, nj, j) * dt, tc = c.lv3(c.L.jf, 9, nj, j) * dt;
        J[c.g(0, nj, j)] = fc.x; J[c.g(1, nj, j)] = fc.y; J[c.g(2, nj, j)] = fc.z;
        J[c.g(3, nj, j)] = tc.x; J[c.g(4, nj, j)] = tc.y; J[c.g(5, nj, j)] = tc.z;
    }
}
// after phase_joints: joint_impulse[j] += (lin_delta_c, ang_delta_c), the child-side correction of this iteration
template <int EPB>
NT_DI void report_joint_iteration(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int nj = c.a.m.nj;
    float* J = c.a.rep.joint_impulse;
    for (int j = c.slot; j < nj; j += c.nslot) {
        int id_p, id_c;
        float m_inv_p, m_inv_c;
        if (!joint_live(c, j, id_p, id_c, m_inv_p, m_inv_c)) continue;
        vec3 jl = c.lv3(c.L.jl, 6, nj, j);
        vec3 ja = ((c.lv3(c.L.jl, 9, nj, j) + c.lv3(c.L.ja, 0, nj, j)) + c.lv3(c.L.ja, 3, nj, j)) + c.lv3(c.L.ja, 6, nj, j);
        J[c.g(0, nj, j)] += jl.x; J[c.g(1, nj, j)] += jl.y; J[c.g(2, nj, j)] += jl.z;
        J[c.g(3, nj, j)] += ja.x; J[c.g(4, nj, j)] += ja.y; J[c.g(5, nj, j)] += ja.z;
    }
}
// end of step: body_parent_f[b] = sum over enabled non-FREE inbound joints (ascending) of joint_impulse / dt
template <int EPB>
NT_DI void report_parent_f(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const nt_model& m = c.a.m;
    const int nb = m.nb, nj = m.nj;
    const float* J = c.a.rep.joint_impulse;
    float* out = c.a.s_out.body_parent_f;
    const float inv_dt = 1.0f / c.a.dt;
    for (int b = c.slot; b < nb; b += c.nslot) {
        vec3 f, t;
        if (J)
            for (int i = c.T.body_joint_start[b]; i < c.T.body_joint_start[b + 1]; ++i) {
                int code = c.T.body_joint_list[i];
                int j = code >> 1;
                if (!(code & 1) || !c.T.joint_enabled[j] || c.T.joint_type[j] == JT_FREE) continue;
                f += vec3(J[c.g(0, nj, j)], J[c.g(1, nj, j)], J[c.g(2, nj, j)]) * inv_dt;
                t += vec3(J[c.g(3, nj, j)], J[c.g(4, nj, j)], J[c.g(5, nj, j)]) * inv_dt;
            }
        out[c.g(0, nb, b)] = f.x; out[c.g(1, nb, b)] = f.y; out[c.g(2, nb, b)] = f.z;
        out[c.g(3, nb, b)] = t.x; out[c.g(4, nb, b)] = t.y; out[c.g(5, nb, b)] = t.z;
    }
}
// number of active contacts on body b in this iteration (constraint_inv_weight[b], xpbd/kernels.py:2287-2291)
template <int EPB>
NT_DI float report_body_contact_count(const Ctx<EPB>& c, int b) {
    const nt_model& m = c.a.m;
    const int cpp = m.cpp, ncs = m.np * cpp;
    float n = 0.0f;
    for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
        int code = c.T.body_pair_list[i];
        int p = code >> 1, side = code & 1;
        for (int k = 0; k < cpp; ++k) {
            int slot = p * cpp + k;
            bool is_a = (side == 0) == (c.l(c.L.cw, 14, ncs, slot) != 0.0f);
            if (c.l(c.L.cw, is_a ? 12 : 13, ncs, slot) != 0.0f) n += 1.0f;
        }
    }
    return n;
}
// contact_impulse[slot] (+)= (lin_delta_a, ang_delta_a) * weight   (accumulate_weighted_contact_impulse)
template <int EPB>
NT_DI void report_contact_iteration(const Ctx<EPB>& c, bool first) {
    if (!c.valid) return;
    const nt_model& m = c.a.m;
    const int cpp = m.cpp, ncs = m.np * cpp;
    float* I = c.a.rep.contact_impulse;
    for (int slot = c.slot; slot < ncs; slot += c.nslot) {
        float has_a = c.l(c.L.cw, 12, ncs, slot), has_b = c.l(c.L.cw, 13, ncs, slot);
        vec3 lin, ang;
        if (has_a != 0.0f || has_b != 0.0f) {
            float weight = 1.0f;
            if (c.a.p.rigid_contact_con_weighting) {
                const int p = slot / cpp;
                int sa = c.T.pair_a[p], sb = c.T.pair_b[p];
                if (c.l(c.L.cw, 14, ncs, slot) == 0.0f) { int t = sa; sa = sb; sb = t; }
                int body_a = c.T.shape_body[sa], body_b = c.T.shape_body[sb];
                float n_a = body_a >= 0 ? report_body_contact_count(c, body_a) : 0.0f;
                float n_b = body_b >= 0 ? report_body_contact_count(c, body_b) : 0.0f;
                float n_sum = n_a + n_b;
                if (n_sum > 0.0f) {
                    if (n_a == 0.0f) weight = 1.0f / n_b;
                    else if (n_b == 0.0f) weight = 1.0f / n_a;
                    else weight = 2.0f / n_sum;
                }
            }
            lin = c.lv3(c.L.cw, 0, ncs, slot) * weight;
            ang = c.lv3(c.L.cw, 3, ncs, slot) * weight;
        }
        if (first) {
            I[c.g(0, ncs, slot)] = lin.x; I[c.g(1, ncs, slot)] = lin.y; I[c.g(2, ncs, slot)] = lin.z;
            I[c.g(3, ncs, slot)] = ang.x; I[c.g(4, ncs, slot)] = ang.y; I[c.g(5, ncs, slot)] = ang.z;
        } else if (has_a != 0.0f || has_b != 0.0f) {
            I[c.g(0, ncs, slot)] += lin.x; I[c.g(1, ncs, slot)] += lin.y; I[c.g(2, ncs, slot)] += lin.z;
            I[c.g(3, ncs, slot)] += ang.x; I[c.g(4, ncs, slot)] += ang.y; I[c.g(5, ncs, slot)] += ang.z;
        }
    }
}

template <int EPB>
NT_DI void phase_joints(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int nj = c.a.m.nj;
    // linear rows on slots [0, nj), angular rows on slots [A0, A0 + nj) with A0 rounded up to a wave boundary (a wave
    // holds 64 / EPB slots): no wavefront then mixes the two code paths, so the phase costs max(linear, angular)
    // instead of their sum in the wave that used to straddle the boundary
    const int spw = 64 / EPB > 0 ? 64 / EPB : 1;
    const int A0 = ((nj + spw - 1) / spw) * spw;
    for (int i = c.slot; i < A0 + nj; i += c.nslot) {
        if (i < nj) joint_linear_item(c, i);
        else if (i >= A0) joint_angular_item(c, i - A0);
    }
}

// ------------------------------------------------------------------------------------------------
// optional per-phase cycle accounting (-DNT_PHASE_TIMING, tools/phase_timing.py): workgroup 0 / thread 0 accumulates the
// s_memtime delta of every phase; never compiled into the product library
// ------------------------------------------------------------------------------------------------
#ifdef NT_PHASE_TIMING
__device__ unsigned long long nt_phase_clock[32];
#define NT_TICK(slot)                                                                  \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                     \
            unsigned long long now = __builtin_readcyclecounter();                     \
            nt_phase_clock[slot] += now - nt_phase_clock[31];                          \
            nt_phase_clock[31] = now;                                                  \
        }                                                                              \
    } while (0)
#else
#define NT_TICK(slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int EPB, bool CVX>
NT_DI void do_collide(const Ctx<EPB>& c, bool count_contacts) {
    if (c.a.debug_skip & 1) return;
    phase_shapes(c);
    __syncthreads();
    NT_TICK(1);
    phase_pairs<EPB, CVX>(c);
    __syncthreads();  // also publishes the contact records (global memory) to the block's contact lanes
    NT_TICK(2);
    if (count_contacts) {  // per-env totals are an API-boundary output, not needed by the solver
        phase_contact_count(c);
        __syncthreads();
    }
}

// SolverXPBD.step control flow (solver_xpbd.py:329-862), rigid-only model
template <int EPB, bool FUSED>
NT_DI void do_xpbd_step(const Ctx<EPB>& c, bool forces_are_zero) {
    const nt_model& m = c.a.m;
    const int skip = c.a.debug_skip;
    const bool restitution = c.a.p.enable_restitution && c.a.has_contacts;
    if (restitution && c.valid)  // body_q_init / body_qd_init: the state the step starts from
        for (int r = c.slot; r < 13 * m.nb; r += c.nslot) c.lds[(c.L.xi + r) * EPB + c.e] = c.lds[(c.L.bq + r) * EPB + c.e];
    const bool rep_joints = !FUSED && c.a.rep.joint_impulse != nullptr;
    const bool rep_contacts = !FUSED && c.a.rep.contact_impulse != nullptr && c.a.has_contacts;
    if (!(skip & 2)) {
        phase_joint_forces(c, forces_are_zero);
        __syncthreads();
        NT_TICK(3);
        if (rep_joints) report_joint_forces(c);
        phase_integrate<EPB, false>(c);
        __syncthreads();
        NT_TICK(4);
    }
    for (int it = 0; it < c.a.p.iterations; ++it) {
        if (c.a.has_contacts) {
            if (!(skip & 4)) phase_contacts<EPB, FUSED>(c);
            __syncthreads();
            NT_TICK(5);
            if (rep_contacts) report_contact_iteration(c, it == 0);
            if (!(skip & 16)) phase_apply<EPB, true>(c);
            __syncthreads();
            NT_TICK(6);
        }
        if (m.nj > 0) {
            if (!(skip & 8)) phase_joints(c);
            __syncthreads();
            NT_TICK(7);
            if (rep_joints) report_joint_iteration(c);
            if (!(skip & 16)) phase_apply<EPB, false>(c);
            __syncthreads();
            NT_TICK(8);
        }
    }
    if (restitution) {  // solver_xpbd.py:784-858
        if (c.valid)
            for (int s = c.slot; s < m.np * m.cpp; s += c.nslot) restitution_item(c, s);
        __syncthreads();
        if (c.valid)
            for (int b = c.slot; b < m.nb; b += c.nslot)
                if (!(c.T.body_flags[b] & BODY_KINEMATIC)) restitution_apply_item(c, b);
        __syncthreads();
    }
    if (!FUSED && c.a.s_out.body_parent_f) {
        __threadfence_block();  // joint lanes' accumulators -> body lanes
        __syncthreads();
        report_parent_f(c);
    }
}

template <int EPB, bool CVX>
__global__ void __launch_bounds__(EPB <= 8 ? 256 : 512) collide_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds);
    load_state(c, a.s_in);
    load_params(c, false);
    __syncthreads();
    do_collide<EPB, CVX>(c, true);
}

template <int EPB>
__global__ void __launch_bounds__(EPB <= 8 ? 256 : 512) xpbd_step_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds);
    load_state(c, a.s_in);
    load_params(c, true);
    __syncthreads();
    phase_body_derived(c);
    __syncthreads();
    do_xpbd_step<EPB, false>(c, false);
    store_state(c, a.s_out);
}

// substeps x { clear_forces; collide; step; swap } with state and parameters resident in LDS across substeps.
// Only the final state is stored (into s0 for an even number of substeps, s1 for odd, like the reference's
// pointer swap); body_f of both states is zeroed as clear_forces would leave it.
template <int EPB, bool CVX>
__global__ void __launch_bounds__(EPB <= 8 ? 256 : 512) xpbd_rollout_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds);
    const int nb = a.m.nb;
    load_state(c, a.s_in);
    load_params(c, true);
    if (c.valid)
        for (int r = c.slot; r < 6 * nb; r += c.nslot) {
            a.s_in.body_f[(size_t)r * c.ES + c.env] = 0.0f;
            a.s_out.body_f[(size_t)r * c.ES + c.env] = 0.0f;
        }
    __syncthreads();
    phase_body_derived(c);
    __syncthreads();
    NT_TICK(0);
    for (int s = 0; s < a.substeps; ++s) {
        do_collide<EPB, CVX>(c, s == a.substeps - 1);
        do_xpbd_step<EPB, true>(c, true);
    }
    store_state(c, (a.substeps & 1) ? a.s_out : a.s_in);
    NT_TICK(9);
}


// ------------------------------------------------------------------------------------------------
// SolverSemiImplicit (solver_semi_implicit.py:123-217): penalty joints + penalty contacts -> integrate_bodies
// ------------------------------------------------------------------------------------------------
// joint_force (semi_implicit/kernels_body.py:17-52)
NT_DI float si_joint_force(float q, float qd, float target_q, float target_qd, float target_ke, float target_kd,
                           float limit_lower, float limit_upper, float limit_ke, float limit_kd, float damping) {
    float limit_f = 0.0f, damping_f = 0.0f;
    float target_f = target_ke * (target_q - q) + target_kd * (target_qd - qd);
    if (q < limit_lower) {
        limit_f = limit_ke * (limit_lower - q);
        damping_f = -limit_kd * qd;
        target_f = 0.0f;
    } else if (q > limit_upper) {
        limit_f = limit_ke * (limit_upper - q);
        damping_f = -limit_kd * qd;
        target_f = 0.0f;
    }
    float passive_f = -damping * qd;
    return limit_f + damping_f + target_f + passive_f;
}
template <int EPB>
NT_DI float si_dof_force(const Ctx<EPB>& c, int dof, int tq, float q, float qd) {
    return si_joint_force(q, qd, c.l(c.L.ctq, 0, 1, tq), c.l(c.L.ctqd, 0, 1, dof), c.dof(DP_TARGET_KE, dof), c.dof(DP_TARGET_KD, dof),
                          c.dof(DP_LIMIT_LOWER, dof), c.dof(DP_LIMIT_UPPER, dof), c.dof(DP_LIMIT_KE, dof), c.dof(DP_LIMIT_KD, dof),
                          c.dof(DP_DAMPING, dof));
}
// signed twist angle of q about `axis`, wrapped to [-pi, pi] (wp.quat_twist_angle_signed, kernels_body.py:206)
NT_DI float quat_twist_angle_signed(vec3 axis, quat q) {
    const float pi = 3.14159265358979323846f;
    float a = q.x * axis.x + q.y * axis.y + q.z * axis.z;
    float angle = 2.0f * atan2f(a, q.w);
    if (angle > pi) angle -= 2.0f * pi;
    if (angle < -pi) angle += 2.0f * pi;
    return angle;
}

// eval_body_joints (semi_implicit/kernels_body.py:55-520): publishes (f, t_parent) and (f, t_child); the body lane adds
// the parent wrench and subtracts the child wrench.  FREE/DISTANCE joints add joint_f to the child: stored negated.
template <int EPB>
NT_DI void si_joint_item(const Ctx<EPB>& c, const int j) {
    const nt_model& m = c.a.m;
    const int nj = m.nj;
    const float ke_att = c.a.sp.joint_attach_ke, kd_att = c.a.sp.joint_attach_kd;
    vec3 f_total, t_total, r_p, r_c;
    const int type = c.T.joint_type[j];
    if (c.T.joint_enabled[j]) {
        const int c_child = c.T.joint_child[j], c_parent = c.T.joint_parent[j];
        const int qd_start = c.T.joint_qd_start[j], tq_start = c.T.joint_tq_start[j];
        if (type == JT_FREE || type == JT_DISTANCE) {
            f_total = -c.lv3(c.L.cf, 0, 1, qd_start);
            t_total = -c.lv3(c.L.cf, 0, 1, qd_start + 3);
        } else {
            xform X_pj = c.lxf(c.L.jp, 0, nj, j), X_cj = c.lxf(c.L.jp, 7, nj, j);
            xform X_wp = X_pj;
            vec3 w_p, v_p;
            if (c_parent >= 0) {
                xform bq = c.body_q(c_parent);
                X_wp = bq * X_wp;
                r_p = X_wp.p - xform_point(bq, c.com(c_parent));
                w_p = c.body_w(c_parent);
                v_p = c.body_v(c_parent) + cross(w_p, r_p);
            }
            xform bqc = c.body_q(c_child);
            xform X_wc = bqc * X_cj;
            r_c = X_wc.p - xform_point(bqc, c.com(c_child));
            vec3 w_c = c.body_w(c_child);
            vec3 v_c = c.body_v(c_child) + cross(w_c, r_c);
            const int lin = c.T.joint_lin_count[j], ang = c.T.joint_ang_count[j];
            vec3 x_err = X_wc.p - X_wp.p;
            quat r_err = quat_inverse(X_wp.q) * X_wc.q;
            vec3 v_err = v_c - v_p;
            vec3 w_err = w_c - w_p;
            const float ads = 0.01f;  // angular_damping_scale
            if (type == JT_FIXED) {
                vec3 ang_err = normalize(vec3(r_err.x, r_err.y, r_err.z)) * acosf(clampf(r_err.w, -1.0f, 1.0f)) * 2.0f;
                f_total += x_err * ke_att + v_err * kd_att;
                t_total += xform_vector(X_wp, ang_err) * ke_att + w_err * kd_att * ads;
            }
            if (type == JT_PRISMATIC) {
                vec3 axis_p = xform_vector(X_wp, c.dof_axis(qd_start));
                float q = dot(x_err, axis_p), qd = dot(v_err, axis_p);
                f_total = axis_p * (-c.l(c.L.cf, 0, 1, qd_start) - si_dof_force(c, qd_start, tq_start, q, qd));
                vec3 ang_err = normalize(vec3(r_err.x, r_err.y, r_err.z)) * acosf(clampf(r_err.w, -1.0f, 1.0f)) * 2.0f;
                f_total += (x_err - q * axis_p) * ke_att + (v_err - qd * axis_p) * kd_att;
                t_total += xform_vector(X_wp, ang_err) * ke_att + w_err * kd_att * ads;
            }
            if (type == JT_REVOLUTE) {
                vec3 axis = c.dof_axis(qd_start);
                vec3 axis_p = xform_vector(X_wp, axis), axis_c = xform_vector(X_wc, axis);
                float q = quat_twist_angle_signed(axis, r_err);
                float qd = dot(w_err, axis_p);
                t_total = axis_p * (-c.l(c.L.cf, 0, 1, qd_start) - si_dof_force(c, qd_start, tq_start, q, qd));
                vec3 swing_err = cross(axis_p, axis_c);
                f_total += x_err * ke_att + v_err * kd_att;
                t_total += swing_err * ke_att + (w_err - qd * axis_p) * kd_att * ads;
            }
            if (type == JT_BALL) {
                f_total += x_err * ke_att + v_err * kd_att;
                for (int k = 0; k < 3; ++k) {
                    vec3 axis_k = xform_vector(X_wp, c.dof_axis(qd_start + k));
                    t_total += axis_k * (-c.l(c.L.cf, 0, 1, qd_start + k) + c.dof(DP_DAMPING, qd_start + k) * dot(axis_k, w_err));
                }
            }
            if (type == JT_D6) {
                vec3 pos(0.0f), vel(0.0f);
                for (int k = 0; k < 3; ++k) {
                    bool take = (k == 0 && lin >= 1) || (k == 1 && lin >= 2) || (k == 2 && lin == 3);
                    if (!take) continue;
                    vec3 axis_k = xform_vector(X_wp, c.dof_axis(qd_start + k));
                    float qk = dot(x_err, axis_k), qdk = dot(v_err, axis_k);
                    f_total += axis_k * (-c.l(c.L.cf, 0, 1, qd_start + k) - si_dof_force(c, qd_start + k, tq_start + k, qk, qdk));
                    pos += qk * axis_k;
                    vel += qdk * axis_k;
                }
                f_total += (x_err - pos) * ke_att + (v_err - vel) * kd_att;
                if (ang == 0) {
                    vec3 ang_err = normalize(vec3(r_err.x, r_err.y, r_err.z)) * acosf(clampf(r_err.w, -1.0f, 1.0f)) * 2.0f;
                    t_total += xform_vector(X_wp, ang_err) * ke_att + w_err * kd_att * ads;
                }
                if (ang == 1) {
                    int i_0 = lin + qd_start, i_0_q = lin + tq_start;
                    vec3 axis = c.dof_axis(i_0);
                    vec3 axis_p = xform_vector(X_wp, axis), axis_c = xform_vector(X_wc, axis);
                    float q = quat_twist_angle_signed(axis, r_err);
                    float qd = dot(w_err, axis_p);
                    t_total = axis_p * (-c.l(c.L.cf, 0, 1, i_0) - si_dof_force(c, i_0, i_0_q, q, qd));
                    vec3 swing_err = cross(axis_p, axis_c);
                    t_total += swing_err * ke_att + (w_err - qd * axis_p) * kd_att * ads;
                }
                // 2 / 3 angular axes need wp.quat_to_euler: rejected on the host (NotImplementedError)
            }
        }
    }
    c.st_lv3(c.L.si_jf, 0, nj, j, f_total);
    c.st_lv3(c.L.si_jf, 3, nj, j, t_total + cross(r_p, f_total));
    c.st_lv3(c.L.si_jf, 6, nj, j, f_total);
    c.st_lv3(c.L.si_jf, 9, nj, j, t_total + cross(r_c, f_total));
}

// eval_body_contact (semi_implicit/kernels_contact.py:381-556), one lane per contact slot: publishes f_total and the
// torques about both bodies' COMs; the body lane subtracts for shape0's body and adds for shape1's body.
template <int EPB>
NT_DI void si_contact_item(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp;
    const float* D = ct.data;
    float has_a = 0.0f, has_b = 0.0f, a_is_pair_a = 1.0f;
    vec3 f_total, tq_a, tq_b;
    size_t gi = (size_t)slot * c.ES + c.env;
    int gid_a = ct.shape0[gi], gid_b = ct.shape1[gi];
    if (gid_a != gid_b) {
        float ke = 0.0f, kd = 0.0f, kf = 0.0f, ka = 0.0f, mu = 0.0f;
        int mat_nonzero = 0, shape_a = -1, shape_b = -1, body_a = -1, body_b = -1;
        if (gid_a >= 0) {
            shape_a = c.local_shape_id(gid_a);
            mat_nonzero += 1;
            ke += c.shape_f(shape_a, SP_KE); kd += c.shape_f(shape_a, SP_KD); kf += c.shape_f(shape_a, SP_KF);
            ka += c.shape_f(shape_a, SP_KA); mu += c.shape_f(shape_a, SP_MU);
            body_a = c.T.shape_body[shape_a];
        }
        if (gid_b >= 0) {
            shape_b = c.local_shape_id(gid_b);
            mat_nonzero += 1;
            ke += c.shape_f(shape_b, SP_KE); kd += c.shape_f(shape_b, SP_KD); kf += c.shape_f(shape_b, SP_KF);
            ka += c.shape_f(shape_b, SP_KA); mu += c.shape_f(shape_b, SP_MU);
            body_b = c.T.shape_body[shape_b];
        }
        if (mat_nonzero > 0) {
            ke /= float(mat_nonzero); kd /= float(mat_nonzero); kf /= float(mat_nonzero);
            ka /= float(mat_nonzero); mu /= float(mat_nonzero);
        }
        vec3 n = -c.gv3(D, CD_NORMAL, ncs, slot);
        vec3 bx_a = c.gv3(D, CD_POINT0, ncs, slot), bx_b = c.gv3(D, CD_POINT1, ncs, slot);
        float margin_a = D[c.g(CD_MARGIN0, ncs, slot)], margin_b = D[c.g(CD_MARGIN1, ncs, slot)];
        vec3 r_a(0.0f), r_b(0.0f);
        if (body_a >= 0) {
            xform X = c.body_q(body_a);
            bx_a = xform_point(X, bx_a) - margin_a * n;
            r_a = bx_a - xform_point(X, c.com(body_a));
        }
        if (body_b >= 0) {
            xform X = c.body_q(body_b);
            bx_b = xform_point(X, bx_b) + margin_b * n;
            r_b = bx_b - xform_point(X, c.com(body_b));
        }
        float d = dot(n, bx_a - bx_b);
        if (d < ka) {
            vec3 bv_a(0.0f), bv_b(0.0f);
            if (body_a >= 0) bv_a = c.body_v(body_a) + cross(c.body_w(body_a), r_a);
            if (body_b >= 0) bv_b = c.body_v(body_b) + cross(c.body_w(body_b), r_b);
            vec3 v = bv_a - bv_b;
            float vn = dot(n, v);
            vec3 vt = v - n * vn;
            float fn = d * ke;
            float fd = fminw(vn, 0.0f) * kd * (d < 0.0f ? 1.0f : 0.0f);
            vec3 ft(0.0f);
            if (d < 0.0f) {
                float delta = c.a.sp.friction_smoothing;
                float a2 = dot(vt, vt);  // wp.norm_huber
                float vs = a2 <= delta * delta ? 0.5f * a2 : delta * (sqrtf(a2) - 0.5f * delta);
                if (vs > 0.0f) {
                    vec3 fr = vt / vs;
                    ft = fr * fminw(kf * vs, -mu * (fn + fd));
                }
            }
            f_total = n * (fn + fd) + ft;
            tq_a = cross(r_a, f_total);
            tq_b = cross(r_b, f_total);
            has_a = body_a >= 0 ? 1.0f : 0.0f;
            has_b = body_b >= 0 ? 1.0f : 0.0f;
            a_is_pair_a = (shape_a == c.T.pair_a[slot / cpp]) ? 1.0f : 0.0f;
        }
    }
    c.st_lv3(c.L.si_cw, 0, ncs, slot, f_total);
    c.st_lv3(c.L.si_cw, 3, ncs, slot, tq_a);
    c.st_lv3(c.L.si_cw, 6, ncs, slot, f_total);
    c.st_lv3(c.L.si_cw, 9, ncs, slot, tq_b);
    c.l(c.L.si_cw, 12, ncs, slot) = has_a;
    c.l(c.L.si_cw, 13, ncs, slot) = has_b;
    c.l(c.L.si_cw, 14, ncs, slot) = a_is_pair_a;
}

template <int EPB>
__global__ void __launch_bounds__(EPB <= 8 ? 256 : 512) semi_implicit_step_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds);
    load_state(c, a.s_in);
    load_params(c, true);
    __syncthreads();
    if (c.valid) {
        const nt_model& m = a.m;
        for (int r = c.slot; r < 6 * m.nb; r += c.nslot) c.lds[(c.L.bf + r) * EPB + c.e] = a.s_in.body_f[(size_t)r * c.ES + c.env];
        // joints and contacts are independent force evaluations on the input state: one phase
        const int ncs = a.has_contacts ? m.np * m.cpp : 0;
        const int spw = 64 / EPB > 0 ? 64 / EPB : 1;  // contact items start on a wave boundary (no mixed-path wave)
        const int C0 = ((m.nj + spw - 1) / spw) * spw;
        for (int i = c.slot; i < C0 + ncs; i += c.nslot) {
            if (i < m.nj) si_joint_item(c, i);
            else if (i >= C0) si_contact_item(c, i - C0);
        }
    }
    __syncthreads();
    // the integrator reads joint wrenches through L.jf: alias it to the semi-implicit region
    Ctx<EPB> ci = c;
    ci.L.jf = c.L.si_jf;
    phase_integrate<EPB, true>(ci);
    __syncthreads();
    store_state(c, a.s_out);
}

#include "nt_featherstone.hpp"

__global__ void clear_forces_kernel(float* body_f, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) body_f[i] = 0.0f;
}

// 4 B/lane coalesced copy with a known byte count: calibrates the FETCH_SIZE / WRITE_SIZE PMC counters for this
// access pattern (MI355X_MICROARCH.md, HBM section)
__global__ void calibration_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

// masked env-column copy: dst[row][env] = src[row][env] for every env whose mask byte is non-zero (RL-style world reset)
__global__ void masked_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, const uint8_t* __restrict__ mask,
                                   int rows, int E, int ES) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)rows * ES;
    if (i >= n) return;
    int env = i % ES;
    if (env < E && mask[env]) dst[i] = src[i];
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
// scan_tmp layout ([4*(E+1)] int32): scanA[E+1] | scanC[E+1] | cntA[E+1] | cntC[E+1]
__global__ void contacts_count_kernel(nt_model m, nt_contacts c, int32_t* scan_tmp) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int E = m.env_count;
    if (env >= E) return;
    const int ES = m.env_stride, ncs = m.np * m.cpp, nas = m.np_analytic * m.cpp;
    int na = 0, nc = 0;
    for (int slot = 0; slot < ncs; ++slot) {
        if (c.shape0[(size_t)slot * ES + env] < 0) continue;
        if (slot < nas) na += 1;
        else nc += 1;
    }
    scan_tmp[2 * (E + 1) + env] = na;
    scan_tmp[3 * (E + 1) + env] = nc;
}
__global__ void contacts_scan_kernel(int E, int32_t* scan_tmp, int32_t* out_count) {
    __shared__ int32_t partA[1024], partC[1024];
    const int32_t *cntA = scan_tmp + 2 * (E + 1), *cntC = scan_tmp + 3 * (E + 1);
    int32_t *scanA = scan_tmp, *scanC = scan_tmp + (E + 1);
    int t = threadIdx.x, T = blockDim.x;
    int per = (E + T - 1) / T;
    int beg = t * per, end = beg + per < E ? beg + per : E;
    int sumA = 0, sumC = 0;
    for (int i = beg; i < end; ++i) { sumA += cntA[i]; sumC += cntC[i]; }
    partA[t] = sumA;
    partC[t] = sumC;
    __syncthreads();
    if (t == 0) {
        int accA = 0, accC = 0;
        for (int i = 0; i < T; ++i) {
            int v = partA[i]; partA[i] = accA; accA += v;
            v = partC[i]; partC[i] = accC; accC += v;
        }
        out_count[0] = accA + accC;
        scanA[E] = accA;  // = first index of the convex section
        scanC[E] = accC;
    }
    __syncthreads();
    int accA = partA[t], accC = partC[t];
    for (int i = beg; i < end; ++i) {
        scanA[i] = accA; accA += cntA[i];
        scanC[i] = accC; accC += cntC[i];
    }
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
    const int E = a.m.env_count;
    if (env >= E) return;
    const int ES = a.m.env_stride, ncs = a.m.np * a.m.cpp, nas = a.m.np_analytic * a.m.cpp;
    int idxA = a.scan[env];
    int idxC = a.scan[E] + a.scan[(E + 1) + env];
    for (int slot = 0; slot < ncs; ++slot) {
        size_t gi = (size_t)slot * ES + env;
        int s0 = a.c.shape0[gi];
        if (s0 < 0) continue;
        int idx = slot < nas ? idxA++ : idxC++;
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
    }
}

// convert_contact_impulse_to_force in export order; entries beyond the live count are zeroed
__global__ void contacts_export_force_kernel(nt_model m, nt_contacts c, const float* __restrict__ impulse, float inv_dt, int cap,
                                             const int32_t* __restrict__ scan, float* __restrict__ out) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int E = m.env_count;
    if (env >= E) return;
    const int ES = m.env_stride, ncs = m.np * m.cpp, nas = m.np_analytic * m.cpp;
    int idxA = scan[env];
    int idxC = scan[E] + scan[(E + 1) + env];
    for (int slot = 0; slot < ncs; ++slot) {
        size_t gi = (size_t)slot * ES + env;
        if (c.shape0[gi] < 0) continue;
        int idx = slot < nas ? idxA++ : idxC++;
        if (idx < cap)
            for (int k = 0; k < 6; ++k) out[6 * (size_t)idx + k] = impulse[((size_t)k * ncs + slot) * ES + env] * inv_dt;
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
constexpr int MAX_THREADS = 512;  // 256 for EPB <= 8 (see max_threads_for)
inline int max_threads_for(int epb) { return epb <= 8 ? 256 : 512; }
constexpr size_t LDS_BYTES_PER_CU = 160 * 1024;

// slot-threads per env: enough for the widest per-env population (contact slots, joint parts, bodies, shapes,
// pairs), capped by the block size; phases with more items than slot-threads loop.
int slots_for(const nt_model& m, int epb) {
    int want = imax(imax(m.nb, 2 * m.nj), imax(imax(m.ns, m.np), m.np * m.cpp + m.nj));
    int cap = max_threads_for(epb) / epb;
    return want < cap ? want : cap;
}

bool epb_fits(const nt_model& m, int epb) {
    return (size_t)make_layout(m).rows_per_env * 4 * epb + (size_t)topo_ints(m) * 4 <= LDS_BYTES_PER_CU;
}

int pick_epb(const nt_model& m, int requested) {
    if (requested == 1 || requested == 8 || requested == 16 || requested == 32 || requested == 64)
        return epb_fits(m, requested) ? requested : 0;
    // auto: the widest tile (best coalescing) that still yields >= 256 workgroups (one per CU); else the narrowest
    const int cands[4] = {64, 32, 16, 8};
    for (int i = 0; i < 4; ++i) {
        int epb = cands[i];
        if (!epb_fits(m, epb)) continue;
        int blocks = (m.env_count + epb - 1) / epb;
        if (blocks >= 256 || epb == 8) return epb;
    }
    for (int i = 3; i >= 0; --i)
        if (epb_fits(m, cands[i])) return cands[i];
    // scenes too large for 8 environments per workgroup (> 20 KB of LDS each): one environment per workgroup, 256 lanes
    // on its items, several workgroups resident per CU while their LDS fits
    return epb_fits(m, 1) ? 1 : 0;
}

template <typename K>
nt_status launch(K kernel, KArgs a, int epb, hipStream_t stream) {
    LdsLayout L = make_layout(a.m);
    int nslot = slots_for(a.m, epb);
    a.nslot = nslot;
    {
        static int dbg = -1;
        if (dbg < 0) { const char* e = getenv("NT_DEBUG_SKIP"); dbg = e ? atoi(e) : 0; }
        a.debug_skip = dbg;
    }
    int threads = ((nslot * epb + 63) / 64) * 64;
    size_t lds_bytes = (size_t)L.rows_per_env * 4 * epb + (size_t)topo_ints(a.m) * 4;
    int blocks = (a.m.env_count + epb - 1) / epb;
    if (lds_bytes > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

#define NT_DISPATCH_EPB(KERNEL, args, epb, stream)                                      \
    ((epb) == 64 ? launch(KERNEL<64>, args, 64, stream)                                 \
     : (epb) == 32 ? launch(KERNEL<32>, args, 32, stream)                               \
     : (epb) == 16 ? launch(KERNEL<16>, args, 16, stream)                               \
     : (epb) == 8 ? launch(KERNEL<8>, args, 8, stream) : launch(KERNEL<1>, args, 1, stream))

#define NT_DISPATCH_EPB2(KERNEL, B, args, epb, stream)                                  \
    ((epb) == 64 ? launch(KERNEL<64, B>, args, 64, stream)                              \
     : (epb) == 32 ? launch(KERNEL<32, B>, args, 32, stream)                            \
     : (epb) == 16 ? launch(KERNEL<16, B>, args, 16, stream)                            \
     : (epb) == 8 ? launch(KERNEL<8, B>, args, 8, stream) : launch(KERNEL<1, B>, args, 1, stream))
// kernels that collide are compiled twice: the convex (MPR/GJK) code only exists in the variant used by models
// that have convex-routed pairs, so analytic-only models keep their register budget
// the convex variants are only instantiated for 1 / 8 / 16 envs per workgroup (build time): wider tiles fall back to 16
#define NT_DISPATCH_EPB_CVX(KERNEL, m, args, epb, stream)                                              \
    ((m).np_analytic < (m).np                                                                          \
         ? ((epb) >= 16 ? launch(KERNEL<16, true>, args, 16, stream)                                   \
            : (epb) == 8 ? launch(KERNEL<8, true>, args, 8, stream) : launch(KERNEL<1, true>, args, 1, stream)) \
         : NT_DISPATCH_EPB2(KERNEL, false, args, epb, stream))

bool model_ok(const nt_model* m) {
    return m && m->env_count > 0 && m->env_stride >= m->env_count && (m->env_stride % 64) == 0 && m->nb > 0 &&
           (m->cpp == 4 || m->cpp == 5) && m->np_analytic >= 0 && m->np_analytic <= m->np &&
           (m->np_analytic == m->np || m->cpp == 5);
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
    return make_layout(*m).rows_per_env * 4;
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
    return NT_DISPATCH_EPB_CVX(collide_kernel, *m, a, epb, (hipStream_t)stream);
}

nt_status nt_xpbd_step(const nt_model* m, const nt_xpbd_params* p, nt_state* s_in, nt_state* s_out, const nt_control* ctrl,
                       const nt_contacts* c, float dt, int32_t envs_per_block, const nt_xpbd_report* report, void* stream) {
    if (!model_ok(m) || !p || !s_in || !s_out || !ctrl) return NT_ERR_INVALID_ARG;
    if (s_out->body_parent_f && m->nj > 0 && !(report && report->joint_impulse)) return NT_ERR_INVALID_ARG;
    KArgs a = {};
    if (report) a.rep = *report;
    if (!s_out->body_parent_f || m->nj == 0) a.rep.joint_impulse = nullptr;
    a.m = *m;
    a.s_in = *s_in;
    a.s_out = *s_out;
    a.c = *ctrl;
    if (c) a.ct = *c;
    a.has_contacts = (c != nullptr && m->np > 0) ? 1 : 0;
    a.p = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    int epb = pick_epb(*m, envs_per_block);
    if (!epb) return NT_ERR_UNSUPPORTED;
    return NT_DISPATCH_EPB(xpbd_step_kernel, a, epb, (hipStream_t)stream);
}

nt_status nt_xpbd_rollout(const nt_model* m, const nt_xpbd_params* p, const nt_collide_params* cp, nt_state* s0, nt_state* s1,
                          const nt_control* ctrl, nt_contacts* c, float dt, int32_t substeps, void* stream) {
    if (!model_ok(m) || !p || !s0 || !s1 || !ctrl || !c || substeps < 1) return NT_ERR_INVALID_ARG;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s0;
    a.s_out = *s1;
    a.c = *ctrl;
    a.ct = *c;
    a.has_contacts = m->np > 0 ? 1 : 0;
    a.p = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    a.substeps = substeps;
    int epb = pick_epb(*m, cp ? cp->envs_per_block : 0);
    if (!epb) return NT_ERR_UNSUPPORTED;
    return NT_DISPATCH_EPB_CVX(xpbd_rollout_kernel, *m, a, epb, (hipStream_t)stream);
}

nt_status nt_semi_implicit_step(const nt_model* m, const nt_semi_implicit_params* p, nt_state* s_in, nt_state* s_out,
                                const nt_control* ctrl, const nt_contacts* c, float dt, int32_t envs_per_block, void* stream) {
    if (!model_ok(m) || !p || !s_in || !s_out || !ctrl) return NT_ERR_INVALID_ARG;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s_in;
    a.s_out = *s_out;
    a.c = *ctrl;
    if (c) a.ct = *c;
    a.has_contacts = (c != nullptr && m->np > 0) ? 1 : 0;
    a.sp = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    int epb = pick_epb(*m, envs_per_block);
    if (!epb) return NT_ERR_UNSUPPORTED;
    return NT_DISPATCH_EPB(semi_implicit_step_kernel, a, epb, (hipStream_t)stream);
}

// shared launch logic of the Featherstone kernels (step / rollout)
static nt_status fs_launch(const nt_model* m, KArgs& a, int32_t envs_per_block, bool rollout, hipStream_t stream) {
    {
        const char* e = getenv("NT_DEBUG_SKIP");
        a.debug_skip = e ? atoi(e) : 0;
    }
    const FsLayout F = make_fs_layout(*m, make_layout(*m));
    const size_t shared_ints = (size_t)topo_ints(*m) + fs_topo_ints(*m);
    auto fits = [&](int epb) { return (size_t)F.rows * 4 * epb + shared_ints * 4 <= LDS_BYTES_PER_CU; };
    int epb = 0;
    if (envs_per_block == 4 || envs_per_block == 8 || envs_per_block == 16) {
        epb = fits(envs_per_block) ? envs_per_block : 0;
    } else {
        // measured on MI355X (4096 quadrupeds): 4 envs per workgroup (16 cooperating lanes per env in the Cholesky wave,
        // two resident workgroups per CU) beats 8; 16 rarely fits
        const int cands[3] = {4, 8, 16};
        for (int i = 0; i < 3 && !epb; ++i)
            if (fits(cands[i])) epb = cands[i];
    }
    if (!epb) return NT_ERR_UNSUPPORTED;
    const bool cvx = m->np_analytic < m->np;
    if (rollout && cvx && epb == 16) epb = 8;  // the convex rollout is only instantiated for 4 / 8 envs per workgroup
    int want = imax(imax(m->nb, m->nj), imax(m->np * m->cpp, imax(m->nj, m->nd) * m->max_art_dofs));
    int cap = 256 / epb;
    a.nslot = want < cap ? want : cap;
    int threads = ((a.nslot * epb + 63) / 64) * 64;
    size_t lds_bytes = (size_t)F.rows * 4 * epb + shared_ints * 4;
    int blocks = (m->env_count + epb - 1) / epb;
    auto go = [&](auto kernel) -> nt_status {
        if (lds_bytes > 48 * 1024 &&
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NT_ERR_LAUNCH;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, a);
        return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
    };
    if (!rollout) {
        if (epb == 16) return go(featherstone_step_kernel<16>);
        if (epb == 8) return go(featherstone_step_kernel<8>);
        return go(featherstone_step_kernel<4>);
    }
    if (cvx) return epb == 8 ? go(featherstone_rollout_kernel<8, true>) : go(featherstone_rollout_kernel<4, true>);
    if (epb == 16) return go(featherstone_rollout_kernel<16, false>);
    if (epb == 8) return go(featherstone_rollout_kernel<8, false>);
    return go(featherstone_rollout_kernel<4, false>);
}

static bool fs_state_ok(const nt_state* s) { return s && s->joint_q && s->joint_qd && s->body_q && s->body_qd; }

nt_status nt_featherstone_step(const nt_model* m, const nt_featherstone_params* p, nt_state* s_in, nt_state* s_out,
                                const nt_control* ctrl, const nt_contacts* c, float dt, int32_t envs_per_block, void* stream) {
    if (!model_ok(m) || !p || !ctrl || !fs_state_ok(s_in) || !fs_state_ok(s_out)) return NT_ERR_INVALID_ARG;
    if (m->nj <= 0 || m->na <= 0 || m->max_art_dofs <= 0 || !m->art_start) return NT_ERR_UNSUPPORTED;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s_in;
    a.s_out = *s_out;
    a.c = *ctrl;
    if (c) a.ct = *c;
    a.has_contacts = (c != nullptr && m->np > 0) ? 1 : 0;
    a.sp.friction_smoothing = p->friction_smoothing;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    return fs_launch(m, a, envs_per_block, false, (hipStream_t)stream);
}

nt_status nt_featherstone_rollout(const nt_model* m, const nt_featherstone_params* p, const nt_collide_params* cp, nt_state* s0,
                                   nt_state* s1, const nt_control* ctrl, nt_contacts* c, float dt, int32_t substeps,
                                   void* stream) {
    if (!model_ok(m) || !p || !ctrl || !c || !fs_state_ok(s0) || !fs_state_ok(s1) || !s0->body_f || !s1->body_f || substeps <= 0)
        return NT_ERR_INVALID_ARG;
    if (m->nj <= 0 || m->na <= 0 || m->max_art_dofs <= 0 || !m->art_start) return NT_ERR_UNSUPPORTED;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s0;
    a.s_out = *s1;
    a.c = *ctrl;
    a.ct = *c;
    a.has_contacts = m->np > 0 ? 1 : 0;
    a.sp.friction_smoothing = p->friction_smoothing;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    a.substeps = substeps;
    return fs_launch(m, a, cp ? cp->envs_per_block : 0, true, (hipStream_t)stream);
}

int32_t nt_featherstone_lds_bytes_per_env(const nt_model* m) {
    if (!m) return -1;
    return make_fs_layout(*m, make_layout(*m)).rows * 4;
}

nt_status nt_eval_fk(const nt_model* m, const float* joint_q, const float* joint_qd, nt_state* out, void* stream) {
    if (!model_ok(m) || !joint_q || !joint_qd || !out || !out->body_q || !out->body_qd) return NT_ERR_INVALID_ARG;
    if (m->nj <= 0) return NT_ERR_UNSUPPORTED;
    KArgs a = {};
    a.m = *m;
    a.s_out = *out;
    const FsLayout F = make_fs_layout(*m, make_layout(*m));
    const size_t shared_ints = (size_t)topo_ints(*m) + fs_topo_ints(*m);
    int epb = 0;
    const int cands[3] = {16, 8, 4};
    for (int i = 0; i < 3 && !epb; ++i)
        if ((size_t)F.rows * 4 * cands[i] + shared_ints * 4 <= LDS_BYTES_PER_CU) epb = cands[i];
    if (!epb) return NT_ERR_UNSUPPORTED;
    int want = imax(m->nb, m->nj), cap = 256 / epb;
    a.nslot = want < cap ? want : cap;
    int threads = ((a.nslot * epb + 63) / 64) * 64;
    size_t lds_bytes = (size_t)F.rows * 4 * epb + shared_ints * 4;
    int blocks = (m->env_count + epb - 1) / epb;
    auto go = [&](auto kernel) -> nt_status {
        if (lds_bytes > 48 * 1024 &&
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NT_ERR_LAUNCH;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, a, joint_q, joint_qd);
        return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
    };
    if (epb == 16) return go(eval_fk_kernel<16>);
    if (epb == 8) return go(eval_fk_kernel<8>);
    return go(eval_fk_kernel<4>);
}

#ifdef NT_PHASE_TIMING
// debug build only: read and reset the phase cycle counters
int nt_debug_phase_clocks(unsigned long long* out) {
    unsigned long long zero[32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nt_phase_clock), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(nt_phase_clock), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

nt_status nt_state_reset(const nt_model* m, nt_state* dst, const nt_state* src, const uint8_t* world_mask, void* stream) {
    if (!model_ok(m) || !dst || !src || !world_mask) return NT_ERR_INVALID_ARG;
    auto go = [&](float* d, const float* s_, int rows) {
        if (!d || !s_ || rows <= 0) return;
        size_t n = (size_t)rows * m->env_stride;
        hipLaunchKernelGGL(masked_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, s_,
                           world_mask, rows, m->env_count, m->env_stride);
    };
    go(dst->body_q, src->body_q, 7 * m->nb);
    go(dst->body_qd, src->body_qd, 6 * m->nb);
    go(dst->body_f, src->body_f, 6 * m->nb);
    go(dst->joint_q, src->joint_q, m->nc);
    go(dst->joint_qd, src->joint_qd, m->nd);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_calibration_copy(const float* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(calibration_copy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, src, dst, (size_t)n);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

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
    hipLaunchKernelGGL(contacts_count_kernel, dim3((m->env_count + 63) / 64), dim3(64), 0, (hipStream_t)stream, *m, *c, scan_tmp);
    hipLaunchKernelGGL(contacts_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, m->env_count, scan_tmp, out_count);
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

nt_status nt_contacts_export_force(const nt_model* m, const nt_contacts* c, const float* contact_impulse, float dt, int32_t cap,
                                   float* out_force, int32_t* scan_tmp, void* stream) {
    if (!model_ok(m) || !c || !contact_impulse || !out_force || !scan_tmp || cap < 0 || !(dt > 0.0f)) return NT_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int E = m->env_count;
    if (hipMemsetAsync(out_force, 0, sizeof(float) * 6 * (size_t)cap, st) != hipSuccess) return NT_ERR_LAUNCH;
    hipLaunchKernelGGL(contacts_count_kernel, dim3((E + 63) / 64), dim3(64), 0, st, *m, *c, scan_tmp);
    // the total goes into the spare last entry of the cntA section
    hipLaunchKernelGGL(contacts_scan_kernel, dim3(1), dim3(1024), 0, st, E, scan_tmp, scan_tmp + 2 * (E + 1) + E);
    hipLaunchKernelGGL(contacts_export_force_kernel, dim3((E + 63) / 64), dim3(64), 0, st, *m, *c, contact_impulse, 1.0f / dt, cap,
                       scan_tmp, out_force);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
