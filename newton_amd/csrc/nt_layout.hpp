// nt_layout.hpp -- row constants, LDS layout, kernel arguments, block-shared topology tables, the per-lane context (Ctx) and the
// HBM <-> LDS staging helpers of the fused gfx950 kernels.
// Included by nt_kernels.hip inside its anonymous namespace, in this order: nt_layout.hpp, nt_collide.hpp, nt_xpbd.hpp,
// nt_semi_implicit.hpp, nt_featherstone.hpp (one translation unit; the split is for reading, not for separate compilation).
#pragma once

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
// per-contact wrench record of SolverSemiImplicit / SolverFeatherstone and of XPBD's restitution pass:
// lin_a, ang_a, lin_b, ang_b, has_a, has_b, shape0_is_pair_a
constexpr int CW_FLOATS = 15;
// per-contact correction record of the XPBD position solve: lin_a (lin_b is its exact negation: -n l_n - t l_f vs n l_n + t l_f),
// ang_a, ang_b, flags (bit 0 has_a, bit 1 has_b, bit 2 shape0_is_pair_a, stored as a small float)
constexpr int CWX_FLOATS = 10;
constexpr int CWX_ANG_A = 3, CWX_ANG_B = 6, CWX_FLAGS = 9;

__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }

// LDS layout, in float rows per environment (each row is EPB floats wide).
// A field is SLOT-MAJOR with a compile-time, ODD stride: element (comp, s) of Fld<NC> sits in row off + s * NC + comp.
//  * one address per (field, item): the components of an item are DS immediate offsets (comp * EPB * 4 bytes) instead of one
//    integer multiply-add each (comp * n + s with a run-time n), and neighbouring components pair into ds_read2_b32;
//  * conflict-free: a wavefront holds 64 / EPB consecutive slots of the same EPB environments, bank = (s * NC * EPB + e) mod 64,
//    and s * NC mod (64 / EPB) is a permutation of the slots because NC is odd (even component counts are padded by one row).
template <int NC>
struct Fld {
    int off;
    static constexpr int nc = NC;
};
constexpr int odd_up(int n) { return n | 1; }
constexpr int NC_JP = odd_up(NT_JOINT_PARAM_FLOATS), NC_DP = odd_up(NT_DOF_PARAM_FLOATS), NC_SP = odd_up(NT_SHAPE_PARAM_FLOATS),
              NC_BP = odd_up(NT_BODY_PARAM_FLOATS), NC_CWX = odd_up(CWX_FLOATS), NC_CW = odd_up(CW_FLOATS);
struct LdsLayout {
    // persistent
    Fld<7> bq, bqd;    // body_q [nb][7], body_qd [nb][6 (+1)]  (adjacent: the restitution snapshot copies both at once)
    Fld<NC_BP> bp;     // body params [nb][23] (inverse mass / inertia already "effective": zero for kinematic bodies)
    Fld<NC_JP> jp;     // joint params [nj][14 (+1)]
    Fld<NC_DP> dp;     // dof params [nd][11]
    Fld<NC_SP> sp;     // shape params [ns][20 (+1)]
    Fld<1> cf, ctq, ctqd;  // control: joint_f [nd], joint_target_q [ntq], joint_target_qd [nd]
    Fld<3> grav;       // gravity [1][3]
    Fld<9> bd;         // body-derived [nb][9]: world COM (3) + world-frame inverse inertia R I^-1 R^T (xx xy xz yy yz zz)
    Fld<1> pm;         // live contacts per pair [np] (written by the collide phase, read by the fused solver phases)
    Fld<1> px;         // exclusive prefix of pm [np + 1]: live contact i of the env is (pair p, sub-contact i - px[p])
    // scratch union
    int u;
    Fld<7> sx, sa;     // collide: shape world xform [ns][7], aabb [ns][6 (+1)]
    Fld<1> pc;         // collide: per-pair contact count [np]
    int poly;          // collide: manifold polygon scratch, 20 rows per convex pair (or per lane: pair-heavy tile)
    Fld<19> st;        // collide: admitted candidates of the analytic pairs [np][19] (normal, 4 x (center, dist)); staged tiles
    Fld<1> hl, hc;     // collide: the environment's compacted broad-phase hits [np] + their count [1] (staged tiles, int bits)
    Fld<7> bf;         // forces: body_f_tmp [nb][6 (+1)]
    Fld<13> jf;        // forces: joint wrenches [nj][12 (+1)]
    Fld<13> jl;        // joints: linear-part corrections [nj][12 (+1)]
    Fld<9> ja;         // joints: angular-part child terms [nj][9]
    Fld<NC_CWX> cw;    // contacts: XPBD per-contact corrections [np*cpp][10 (+1)]
    Fld<NC_CW> cwr;    // the same rows as 15-float records (XPBD restitution pass)
    Fld<7> si_bf; Fld<13> si_jf; Fld<NC_CW> si_cw;  // semi-implicit: body_f_tmp + joint wrenches + contact wrenches, all live together
    Fld<7> xiq, xiqd;  // XPBD restitution: pre-step body_q / body_qd, behind the XPBD scratch (solver_xpbd.py:414-416)
    int rows_per_env;  // collide / XPBD kernels
    int uni_floats;    // uniform-parameter tiles: bp / jp / dp / sp index ONE block-shared copy of this many floats instead of rows
    int rows_semi;     // SolverSemiImplicit kernel (its wrench records share the scratch with body_f_tmp + joint wrenches)
};

constexpr int NT_BIG_SCENE_LANES = 256;  // workgroup size of the one-environment-per-workgroup tile
constexpr int NT_MIN_SCRATCH_ROWS = 8;   // the live-contact prefix parks up to 8 partial sums in the scratch union

// collide scratch at row `base`: shape transforms / AABBs, pair counts, manifold polygon scratch (+ staged candidates);
// returns its size in rows.  SolverFeatherstone's fused rollout places the same block inside its own union
__host__ __device__ inline int place_collide_scratch(LdsLayout& L, const nt_model& m, const int base, const bool big) {
    L.sx.off = base; L.sa.off = L.sx.off + 7 * m.ns; L.pc.off = L.sa.off + 7 * m.ns;
    L.poly = L.pc.off + m.np;
    // manifold polygon scratch: 20 rows per convex pair, or (pair-heavy scenes, one environment per workgroup) per lane
    int coll = 14 * m.ns + m.np + 20 * (big ? NT_BIG_SCENE_LANES : (m.np - m.np_analytic));
    L.st.off = base + coll;
    if (!big) coll += 19 * m.np;
    L.hl.off = base + coll; L.hc.off = L.hl.off + m.np;
    if (!big) coll += m.np + 1;
    return coll;
}

// big: pair-heavy scenes (nt_model.contact_scratch_in_hbm).  Device code passes a compile-time constant so that the
// default kernels carry no trace of the second mode.
// restitution: SolverXPBD(enable_restitution=True) keeps the pre-step state and 15-float velocity records per contact slot
// uni: nt_model.params_uniform models on a uniform-parameter tile -- the body / joint / dof / shape parameters are identical
// in every environment, so the workgroup keeps ONE copy (block-shared, broadcast reads) and an environment's LDS footprint
// drops by 936 rows on the headline quadruped: 32 environments fit a CU instead of 16
__host__ __device__ inline LdsLayout make_layout(const nt_model& m, const bool big, const bool restitution = false,
                                                 const bool uni = false) {
    LdsLayout L;
    int o = 0, ou = 0;
    L.bq.off = o; o += 7 * m.nb;
    L.bqd.off = o; o += 7 * m.nb;
    int& po = uni ? ou : o;
    L.bp.off = po; po += NC_BP * m.nb;
    L.jp.off = po; po += NC_JP * m.nj;
    L.dp.off = po; po += NC_DP * m.nd;
    L.sp.off = po; po += NC_SP * m.ns;
    L.uni_floats = ou;
    L.cf.off = o; o += m.nd;
    L.ctq.off = o; o += m.ntq;
    L.ctqd.off = o; o += m.nd;
    L.grav.off = o; o += 3;
    L.bd.off = o; o += 9 * m.nb;
    L.pm.off = o; o += m.np;
    L.px.off = o; o += m.np + 1;
    L.u = o;
    const int coll = place_collide_scratch(L, m, L.u, big);
    // staged tiles: the force scratch sits BEHIND the collide scratch, so that the fused rollout can run the shape phase
    // and the joint-force phase in the same barrier interval (different waves); the pair-heavy tile keeps the overlap
    L.bf.off = big ? L.u : L.u + coll; L.jf.off = L.bf.off + 7 * m.nb;
    int forces = (big ? 0 : coll) + 7 * m.nb + 13 * m.nj;
    L.jl.off = L.u; L.ja.off = L.jl.off + 13 * m.nj;
    int joints = 22 * m.nj;
    L.cw.off = L.u; L.cwr.off = L.u;
    // big: the records live in nt_contacts.cw (HBM)
    int contacts = big ? 0 : (restitution ? NC_CW : NC_CWX) * m.np * m.cpp;
    L.si_bf.off = L.u; L.si_jf.off = L.si_bf.off + 7 * m.nb; L.si_cw.off = L.si_jf.off + 13 * m.nj;
    int semi = 7 * m.nb + 13 * m.nj + NC_CW * m.np * m.cpp;
    int xpbd = imax(imax(imax(coll, forces), imax(joints, contacts)), NT_MIN_SCRATCH_ROWS);
    L.xiq.off = L.u + xpbd; L.xiqd.off = L.xiq.off + 7 * m.nb;
    L.rows_per_env = L.u + xpbd + (restitution ? 14 * m.nb : 0);
    L.rows_semi = L.u + semi;
    return L;
}
inline LdsLayout make_layout_host(const nt_model& m, bool restitution = false, bool uni = false) {
    return make_layout(m, m.contact_scratch_in_hbm != 0, restitution, uni);
}

// the pre-step state snapshot (and the wide contact records) exist for restitution and for velocities from position deltas
__host__ __device__ inline bool xpbd_keeps_prestep_state(const nt_xpbd_params& p) {
    return p.enable_restitution != 0 || p.compute_body_velocity_from_position_delta != 0;
}

// Phase ablation exists only in throw-away measurement builds (-DNT_ABLATION, tools/xpbd_ablation.sh, tools/fs_ablation.sh): the
// product library carries no work-skipping switch.
#ifdef NT_ABLATION
#define NT_SKIP_DECL(args) const int nt_skip_mask = (args).debug_skip
#define NT_SKIP(bit) ((nt_skip_mask & (bit)) != 0)
#else
#define NT_SKIP_DECL(args) ((void)0)
#define NT_SKIP(bit) false
#endif

struct KArgs {
    nt_model m;
    nt_state s_in, s_out;
    nt_control c;
    nt_contacts ct;
    nt_xpbd_params p;
    nt_xpbd_report rep;  // optional reporting outputs of nt_xpbd_step (all NULL on the hot path)
    nt_semi_implicit_params sp;
    nt_featherstone_params fp;  // mass-matrix update cadence of SolverFeatherstone (zero: rebuild every step)
    float angular_damping;  // integrate_bodies damping of the active solver
    float dt;
    int substeps;
    int has_contacts;
    int nslot;       // slot-threads per environment
#ifdef NT_ABLATION
    int debug_skip;  // measurement builds only (tools/xpbd_ablation.sh): 1 collide, 2 forces+integrate, 4 contacts, 8 joints, 16 apply
#endif
};

// env-uniform topology, staged once per workgroup into LDS (block-shared ints behind the per-env rows)
struct Topo {
    const int *body_flags, *joint_type, *joint_enabled, *joint_parent, *joint_child, *joint_q_start, *joint_qd_start,
        *joint_tq_start, *joint_lin_count, *joint_ang_count, *shape_body, *shape_type, *shape_flags, *shape_group, *pair_a,
        *pair_b, *body_joint_start, *body_joint_list, *body_pair_start, *body_pair_list, *shape_mesh_start, *shape_mesh_count, *gshape_id;
    const float* gshape;  // [ng][NT_SHAPE_PARAM_FLOATS] parameters of the global (world -1) shapes, block-shared copy
    int *hit_count, *hit_list;  // pair-heavy tile only: the environment's compacted candidate list (1 + np ints)
};
__host__ __device__ inline int topo_ints(const nt_model& m) {
    return m.nb + 9 * m.nj + 6 * (m.ns + m.ng) + 2 * m.np + 2 * (m.nb + 1) + 2 * m.nj + 2 * m.np + m.ng +
           NT_SHAPE_PARAM_FLOATS * m.ng + (m.contact_scratch_in_hbm ? 1 + m.np : 0);
}

// The tile code EPB of every kernel template carries the environments per workgroup in its low byte and the
// uniform-parameter flag NT_UNI in bit 8: Ctx<EPB>::N is the count, Ctx<EPB>::UNI the flag.
constexpr int NT_UNI = 256;
template <int EPB>
struct Ctx {
    static constexpr int N = EPB & 255;
    static constexpr bool UNI = (EPB & NT_UNI) != 0;
    const KArgs& a;
    Topo T;
    float* lds;
    float* up;  // UNI: the block-shared parameter copy [L.uni_floats], behind the topology
    LdsLayout L;
    int e, slot, env, nslot;
    int tslot;  // start of the item loop of phases with fewer items than slot-threads.  Identity: dealing consecutive items to
                // DIFFERENT waves (13 bodies x 16 envs on all 8 waves instead of 4) was measured and lost 13 % on the headline
                // (87.4 vs 100.8 M env-steps/s) -- twice the wave-instructions cost more than the second wave per SIMD hides
    bool big;  // contact records in HBM, manifold polygon scratch per lane (compile-time constant at every construction site)
    int ES;
    bool valid;

    // rows: float rows per env in front of the block-shared topology ints (-1: the XPBD / collide layout)
    NT_DI Ctx(const KArgs& a_, float* lds_, int rows = -1, const bool big_ = false) : a(a_), lds(lds_), big(big_) {
        L = make_layout(a.m, big_, xpbd_keeps_prestep_state(a.p), UNI);
        if (rows < 0) rows = L.rows_per_env;
        e = threadIdx.x % N;
        slot = threadIdx.x / N;
        nslot = a.nslot;
        env = blockIdx.x * N + e;
        ES = a.m.env_stride;
        valid = env < a.m.env_count && slot < nslot;
        tslot = slot;
        const nt_model& m = a.m;
        int* ti = reinterpret_cast<int*>(lds + (size_t)rows * N);
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
        T.hit_count = ti + o;
        T.hit_list = ti + o + 1;
        up = reinterpret_cast<float*>(ti + topo_ints(m));
    }
    // LDS element (comp, s) of a slot-major field.  `n` (the slot count of the [comp][n] HBM twin) is not needed here; the
    // argument stays so that every access reads like its global-memory counterpart g(comp, n, s)
    template <int NC>
    NT_DI float& l(Fld<NC> f, int comp, int /*n*/, int s) const { return lds[(f.off + s * NC + comp) * N + e]; }
    // parameter element (fields bp / jp / dp / sp): per-environment row, or the block-shared copy of a uniform tile
    template <int NC>
    NT_DI float pl(Fld<NC> f, int comp, int /*n*/, int s) const {
        if constexpr (UNI) return up[f.off + s * NC + comp];
        else return lds[(f.off + s * NC + comp) * N + e];
    }
    template <int NC>
    NT_DI vec3 plv3(Fld<NC> f, int comp0, int n, int s) const {
        return vec3(pl(f, comp0, n, s), pl(f, comp0 + 1, n, s), pl(f, comp0 + 2, n, s));
    }
    template <int NC>
    NT_DI xform plxf(Fld<NC> f, int comp0, int n, int s) const {
        return xform(plv3(f, comp0, n, s),
                     quat(pl(f, comp0 + 3, n, s), pl(f, comp0 + 4, n, s), pl(f, comp0 + 5, n, s), pl(f, comp0 + 6, n, s)));
    }
    template <int NC>
    NT_DI mat33 plm33(Fld<NC> f, int comp0, int n, int s) const {
        return mat33(pl(f, comp0, n, s), pl(f, comp0 + 1, n, s), pl(f, comp0 + 2, n, s), pl(f, comp0 + 3, n, s),
                     pl(f, comp0 + 4, n, s), pl(f, comp0 + 5, n, s), pl(f, comp0 + 6, n, s), pl(f, comp0 + 7, n, s),
                     pl(f, comp0 + 8, n, s));
    }
    NT_DI size_t g(int comp, int n, int s) const { return (size_t)(comp * n + s) * ES + env; }

    template <int NC>
    NT_DI vec3 lv3(Fld<NC> f, int comp0, int n, int s) const {
        return vec3(l(f, comp0, n, s), l(f, comp0 + 1, n, s), l(f, comp0 + 2, n, s));
    }
    template <int NC>
    NT_DI void st_lv3(Fld<NC> f, int comp0, int n, int s, vec3 v) const {
        l(f, comp0, n, s) = v.x; l(f, comp0 + 1, n, s) = v.y; l(f, comp0 + 2, n, s) = v.z;
    }
    template <int NC>
    NT_DI xform lxf(Fld<NC> f, int comp0, int n, int s) const {
        return xform(lv3(f, comp0, n, s),
                     quat(l(f, comp0 + 3, n, s), l(f, comp0 + 4, n, s), l(f, comp0 + 5, n, s), l(f, comp0 + 6, n, s)));
    }
    template <int NC>
    NT_DI void st_lxf(Fld<NC> f, int n, int s, const xform& t) const {
        l(f, 0, n, s) = t.p.x; l(f, 1, n, s) = t.p.y; l(f, 2, n, s) = t.p.z;
        l(f, 3, n, s) = t.q.x; l(f, 4, n, s) = t.q.y; l(f, 5, n, s) = t.q.z; l(f, 6, n, s) = t.q.w;
    }
    template <int NC>
    NT_DI mat33 lm33(Fld<NC> f, int comp0, int n, int s) const {
        return mat33(l(f, comp0, n, s), l(f, comp0 + 1, n, s), l(f, comp0 + 2, n, s), l(f, comp0 + 3, n, s),
                     l(f, comp0 + 4, n, s), l(f, comp0 + 5, n, s), l(f, comp0 + 6, n, s), l(f, comp0 + 7, n, s),
                     l(f, comp0 + 8, n, s));
    }
    NT_DI vec3 gravity() const { return lv3(L.grav, 0, 1, 0); }
    // component-major [comp][n] arrays at a plain row offset (a solver's own scratch, e.g. SolverFeatherstone's)
    NT_DI float& l(int off, int comp, int n, int s) const { return lds[(off + comp * n + s) * N + e]; }
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
    NT_DI float inv_mass(int b) const { return pl(L.bp, BP_INV_MASS, a.m.nb, b); }
    NT_DI mat33 inv_inertia(int b) const { return plm33(L.bp, BP_INV_INERTIA, a.m.nb, b); }
    NT_DI mat33 inertia(int b) const { return plm33(L.bp, BP_INERTIA, a.m.nb, b); }
    NT_DI vec3 com(int b) const { return plv3(L.bp, BP_COM, a.m.nb, b); }
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
    NT_DI float dof(int row, int d) const { return pl(L.dp, row, a.m.nd, d); }
    NT_DI vec3 dof_axis(int d) const { return plv3(L.dp, DP_AXIS, a.m.nd, d); }

    // shape accessors: s < ns local (per-env params in LDS), otherwise the env-uniform global table
    NT_DI float shape_f(int s, int comp) const {
        if (s < a.m.ns) return pl(L.sp, comp, a.m.ns, s);
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
// field [ncomp][n][ES] in HBM <-> slot-major rows in LDS
template <int EPB, int NC>
NT_DI void stage_rows(const Ctx<EPB>& c, Fld<NC> f, const float* src, int ncomp, int n) {
    for (int comp = 0; comp < ncomp; ++comp)
        for (int s = c.slot; s < n; s += c.nslot) c.l(f, comp, n, s) = src[c.g(comp, n, s)];
}
template <int EPB, int NC>
NT_DI void unstage_rows(const Ctx<EPB>& c, Fld<NC> f, float* dst, int ncomp, int n) {
    for (int comp = 0; comp < ncomp; ++comp)
        for (int s = c.slot; s < n; s += c.nslot) dst[c.g(comp, n, s)] = c.l(f, comp, n, s);
}
// plain rows (a solver's own [row] arrays)
template <int EPB>
NT_DI void stage_rows(const Ctx<EPB>& c, int lds_off, const float* src, int rows) {
    for (int r = c.slot; r < rows; r += c.nslot) c.lds[(lds_off + r) * Ctx<EPB>::N + c.e] = src[(size_t)r * c.ES + c.env];
}
template <int EPB>
NT_DI void unstage_rows(const Ctx<EPB>& c, int lds_off, float* dst, int rows) {
    for (int r = c.slot; r < rows; r += c.nslot) dst[(size_t)r * c.ES + c.env] = c.lds[(lds_off + r) * Ctx<EPB>::N + c.e];
}

template <int EPB>
NT_DI void load_state(const Ctx<EPB>& c, const nt_state& s) {
    if (!c.valid) return;
    stage_rows(c, c.L.bq, s.body_q, 7, c.a.m.nb);
    stage_rows(c, c.L.bqd, s.body_qd, 6, c.a.m.nb);
}
template <int EPB>
NT_DI void store_state(const Ctx<EPB>& c, const nt_state& s) {
    if (!c.valid) return;
    unstage_rows(c, c.L.bq, s.body_q, 7, c.a.m.nb);
    unstage_rows(c, c.L.bqd, s.body_qd, 6, c.a.m.nb);
}
// parameters and controls: read once per kernel
template <int EPB>
NT_DI void load_params(const Ctx<EPB>& c, bool with_control) {
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    // body params carry the effective (kinematic => 0) inverse mass / inertia (solver.py:173-187)
    auto body_value = [&](int comp, int b, size_t col) {
        float v = m.body_param[(size_t)(comp * nb + b) * c.ES + col];
        bool inv_row = comp == BP_INV_MASS || (comp >= BP_INV_INERTIA && comp < BP_INV_INERTIA + 9);
        if (inv_row && (m.body_flags[b] & BODY_KINEMATIC)) v = 0.0f;  // (global copy: the LDS topology is not published yet)
        return v;
    };
    if constexpr (Ctx<EPB>::UNI) {
        // one copy per workgroup, read from the tile's first environment (the host vouches that all columns are equal)
        const size_t col = (size_t)blockIdx.x * Ctx<EPB>::N;
        for (int r = threadIdx.x; r < NT_BODY_PARAM_FLOATS * nb; r += blockDim.x) {
            int comp = r / nb, b = r - comp * nb;
            c.up[c.L.bp.off + b * NC_BP + comp] = body_value(comp, b, col);
        }
        for (int r = threadIdx.x; r < NT_JOINT_PARAM_FLOATS * m.nj; r += blockDim.x) {
            int comp = r / m.nj, j = r - comp * m.nj;
            c.up[c.L.jp.off + j * NC_JP + comp] = m.joint_param[(size_t)r * c.ES + col];
        }
        for (int r = threadIdx.x; r < NT_DOF_PARAM_FLOATS * m.nd; r += blockDim.x) {
            int comp = r / m.nd, d = r - comp * m.nd;
            c.up[c.L.dp.off + d * NC_DP + comp] = m.dof_param[(size_t)r * c.ES + col];
        }
        for (int r = threadIdx.x; r < NT_SHAPE_PARAM_FLOATS * m.ns; r += blockDim.x) {
            int comp = r / m.ns, sh = r - comp * m.ns;
            c.up[c.L.sp.off + sh * NC_SP + comp] = m.shape_param[(size_t)r * c.ES + col];
        }
    }
    if (!c.valid) return;
    if constexpr (!Ctx<EPB>::UNI) {
        for (int comp = 0; comp < NT_BODY_PARAM_FLOATS; ++comp)
            for (int b = c.slot; b < nb; b += c.nslot) c.lds[(c.L.bp.off + b * NC_BP + comp) * Ctx<EPB>::N + c.e] = body_value(comp, b, c.env);
        stage_rows(c, c.L.jp, m.joint_param, NT_JOINT_PARAM_FLOATS, m.nj);
        stage_rows(c, c.L.dp, m.dof_param, NT_DOF_PARAM_FLOATS, m.nd);
        stage_rows(c, c.L.sp, m.shape_param, NT_SHAPE_PARAM_FLOATS, m.ns);
    }
    stage_rows(c, c.L.grav, m.gravity, 3, 1);
    if (with_control) {
        stage_rows(c, c.L.cf, c.a.c.joint_f, 1, m.nd);
        stage_rows(c, c.L.ctq, c.a.c.joint_target_q, 1, m.ntq);
        stage_rows(c, c.L.ctqd, c.a.c.joint_target_qd, 1, m.nd);
    }
}
