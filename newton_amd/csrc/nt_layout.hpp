// nt_layout.hpp -- row constants, LDS layout, kernel arguments and the block-shared topology tables of the fused gfx950 kernels:
// everything that carries no vector-math type.  The per-lane context (Ctx) and the HBM <-> LDS staging helpers are nt_ctx.hpp.
// Included ONCE by nt_kernels.hip at the top of its anonymous namespace; nt_ctx.hpp / nt_xpbd.hpp are then included once per
// arithmetic namespace (`ieee`: nt:: helpers, no contraction; `fused`: ntf:: helpers, a * b + c may contract -- see nt_kernels.hip).
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
    Fld<1> lt;         // the environment's live contacts [np * cpp], compacted: entry i = pair << 4 | sub-contact (int bits), written
                       // with the prefix; the fused contact phases read it instead of searching px.  Only where has_lt: the XPBD /
                       // collide layout of analytic-only models on the staged tiles (np * cpp rows per environment cost the convex
                       // models a tile size -- C2 fell from 16 to 8 environments per workgroup, 45 -> 20 M env-steps/s -- and the
                       // other solvers never read it)
    int has_lt;
    Fld<1> lc;         // [1] number of live contacts appended to lt so far (int bits; LDS-record tiles: the pair lanes append with an atomic)
    Fld<17> cr;        // contact records [np * cpp][17] in LDS (NT_TILE_LDS_RECORDS; behind the snapshot rows)
    // scratch union
    int u;
    Fld<7> sx, sa;     // collide: shape world xform [ns][7], aabb [ns][6 (+1)]
    Fld<1> pc;         // collide: per-pair contact count [np]
    int poly;          // collide: manifold polygon scratch, 20 rows per convex pair (or per lane: pair-heavy tile)
    Fld<19> st;        // collide: admitted candidates of the analytic pairs [np][19] (normal, 4 x (center, dist)); staged tiles
    Fld<1> hl, hc;     // collide: the environment's compacted broad-phase hits [np] + their count [1] (staged tiles, int bits)
    Fld<7> bf;         // forces: body_f_tmp [nb][6 (+1)]
    Fld<13> jf;        // forces: joint wrenches [nj][12 (+1)]
    Fld<9> ji;         // joints: corrections in body-incidence order [2 nj][9] -- entry i of Topo::body_joint_list (a (joint, side)):
                       // lin (3), ang of the linear rows (3) from the joint's linear lane, summed angular-row terms (3, sign applied)
                       // from its angular lane; a body's entries are contiguous: its lane sums them from ONE base address
    Fld<NC_CWX> cw;    // contacts: XPBD per-contact corrections [np*cpp][10 (+1)]
    Fld<NC_CW> cwr;    // the same rows as 15-float records (XPBD restitution pass)
    Fld<7> si_bf; Fld<13> si_jf; Fld<NC_CW> si_cw;  // semi-implicit: body_f_tmp + joint wrenches + contact wrenches, all live together
    Fld<7> xiq, xiqd;  // XPBD restitution: pre-step body_q / body_qd, behind the XPBD scratch (solver_xpbd.py:414-416)
    int rows_per_env;  // collide / XPBD kernels
    int uni_floats;    // uniform-parameter tiles: bp / jp / dp / sp index ONE block-shared copy of this many floats instead of rows
    int rows_semi;     // SolverSemiImplicit kernel (its wrench records share the scratch with body_f_tmp + joint wrenches)
};

constexpr int NT_CR_STRIDE = 32;  // floats per (environment, slot) record of nt_contacts.cr: 17 used, one 128-byte line
constexpr int NT_BIG_SCENE_LANES = 256;  // workgroup size of the pair-heavy one-environment-per-workgroup tile ...
constexpr int NT_BIG_SCENE_LANES_WIDE = 384;  // ... and its wide form, taken when 20 more rows of manifold polygon scratch per extra lane
                                              // still fit the CU: two of the four SIMDs then interleave two waves (the pair phase is a
                                              // chain of dependent MPR / GJK iterations: config C5's geometry 65.2 -> 53.4 ms per frame
                                              // same-box, profiles/r05M_ab.txt).  512 lanes do not fit C5 (168 KB)
constexpr int NT_MIN_SCRATCH_ROWS = 8;   // the live-contact prefix parks up to 8 partial sums in the scratch union

// collide scratch at row `base`: shape transforms / AABBs, pair counts, manifold polygon scratch (+ staged candidates);
// returns its size in rows.  SolverFeatherstone's fused rollout places the same block inside its own union
// big_lanes: lanes of the pair-heavy tile's workgroup (each owns a polygon scratch block)
__host__ __device__ inline int place_collide_scratch(LdsLayout& L, const nt_model& m, const int base, const bool big,
                                                     const int big_lanes = NT_BIG_SCENE_LANES) {
    L.sx.off = base; L.sa.off = L.sx.off + 7 * m.ns; L.pc.off = L.sa.off + 7 * m.ns;
    L.poly = L.pc.off + m.np;
    // manifold polygon scratch: 20 rows per convex pair, or (pair-heavy scenes, one environment per workgroup) per lane
    int coll = 14 * m.ns + m.np + 20 * (big ? big_lanes : (m.np - m.np_analytic));
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
// opts: NT_TILE_* bits the launch code granted (KArgs::tile_opts; only the fused XPBD rollout asks for any)
constexpr int NT_TILE_POSE_SNAPSHOT = 1;  // keep the substep's incoming body poses (L.xiq) so that integrate_bodies can run beside the pair phase
constexpr int NT_TILE_LDS_RECORDS = 2;    // the contact records of a fused rollout live in LDS (L.cr); Contacts in HBM get the last substep's only
__host__ __device__ inline LdsLayout make_layout(const nt_model& m, const bool big, const bool restitution = false,
                                                 const bool uni = false, const bool live_list = true, const int opts = 0,
                                                 const int big_lanes = NT_BIG_SCENE_LANES) {
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
    L.pm.off = o; o += m.np;
    L.px.off = o; o += m.np + 1;
    L.has_lt = live_list && !big && m.np_analytic == m.np;
    L.lt.off = o; o += L.has_lt ? m.np * m.cpp : 0;
    L.lc.off = o; o += L.has_lt ? 1 : 0;
    L.bd.off = o; o += 9 * m.nb;  // (last of the persistent block: SolverFeatherstone's layout, which never reads it, starts here)
    L.u = o;
    const int coll = place_collide_scratch(L, m, L.u, big, big_lanes);
    // staged tiles: the force scratch sits BEHIND the collide scratch, so that the fused rollout can run the shape phase
    // and the joint-force phase in the same barrier interval (different waves); the pair-heavy tile keeps the overlap
    L.bf.off = big ? L.u : L.u + coll; L.jf.off = L.bf.off + 7 * m.nb;
    int forces = (big ? 0 : coll) + 7 * m.nb + 13 * m.nj;
    L.ji.off = L.u;
    int joints = 18 * m.nj;
    L.cw.off = L.u; L.cwr.off = L.u;
    // big: the records live in nt_contacts.cw (HBM)
    int contacts = big ? 0 : (restitution ? NC_CW : NC_CWX) * m.np * m.cpp;
    L.si_bf.off = L.u; L.si_jf.off = L.si_bf.off + 7 * m.nb; L.si_cw.off = L.si_jf.off + 13 * m.nj;
    int semi = 7 * m.nb + 13 * m.nj + NC_CW * m.np * m.cpp;
    int xpbd = imax(imax(imax(coll, forces), imax(joints, contacts)), NT_MIN_SCRATCH_ROWS);
    L.xiq.off = L.u + xpbd; L.xiqd.off = L.xiq.off + 7 * m.nb;
    const bool snapshot = restitution || (opts & NT_TILE_POSE_SNAPSHOT) != 0;  // poses (7 rows per body), + velocities with restitution
    L.rows_per_env = L.u + xpbd + (snapshot ? 7 * m.nb : 0) + (restitution ? 7 * m.nb : 0);
    L.cr.off = L.rows_per_env;
    if ((opts & NT_TILE_LDS_RECORDS) && L.has_lt && !restitution) L.rows_per_env += 17 * m.np * m.cpp;
    L.rows_semi = L.u + semi;
    return L;
}
inline LdsLayout make_layout_host(const nt_model& m, bool restitution = false, bool uni = false, int opts = 0, int big_lanes = NT_BIG_SCENE_LANES) {
    return make_layout(m, m.contact_scratch_in_hbm != 0, restitution, uni, true, opts, big_lanes);
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
    int tile_opts;   // NT_TILE_* bits of the LDS layout (set by the launch code when the tile still fits the CU)
#ifdef NT_ABLATION
    int debug_skip;  // measurement builds only (tools/xpbd_ablation.sh): 1 collide, 2 forces+integrate, 4 contacts, 8 joints, 16 apply
#endif
};

// env-uniform topology, staged once per workgroup into LDS (block-shared ints behind the per-env rows)
struct Topo {
    const int *body_flags, *joint_type, *joint_enabled, *joint_parent, *joint_child, *joint_q_start, *joint_qd_start,
        *joint_tq_start, *joint_lin_count, *joint_ang_count, *shape_body, *shape_type, *shape_flags, *shape_group, *pair_a,
        *pair_b, *body_joint_start, *body_joint_list, *body_pair_start, *body_pair_list, *shape_mesh_start, *shape_mesh_count, *gshape_id;
    const int* joint_inc;  // [2 nj] inverse of body_joint_list: position of (joint j, side) = code j << 1 | side in the list, -1 (world side)
    // [np][4] (not in the pair-heavy tile): a pair as the contact phases need it -- shape0, shape1 in the narrow phase's type-sorted
    // order (narrow_phase.py:525-528), their bodies (-1 static), bit 30 of the last word: shape0 is the pair's second shape
    const int* pair_desc;
    const float* gshape;  // [ng][NT_SHAPE_PARAM_FLOATS] parameters of the global (world -1) shapes, block-shared copy
    float* gworld;        // [ng][13] world transform + gap-widened AABB of the global shapes (static: staged once per launch)
    int *hit_count, *hit_list;  // pair-heavy tile only: the environment's compacted candidate list (1 + np ints)
};
__host__ __device__ inline int topo_ints(const nt_model& m) {
    return m.nb + 9 * m.nj + 6 * (m.ns + m.ng) + 2 * m.np + 2 * (m.nb + 1) + 2 * m.nj + 2 * m.np + m.ng +
           NT_SHAPE_PARAM_FLOATS * m.ng + 13 * m.ng + 2 * m.nj + (m.contact_scratch_in_hbm ? 1 + m.np : 4 * m.np);
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
#define NT_TICK_START()                                                                \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) nt_phase_clock[31] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define NT_TICK(slot) do { } while (0)
#define NT_TICK_START() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// SolverFeatherstone's LDS layout (rows behind the persistent block of the XPBD layout) and the sizes of its block-shared tables;
// no vector-math type, shared by both arithmetic namespaces and the host launch code (the phases are nt_featherstone.hpp)
// ------------------------------------------------------------------------------------------------
struct FsLayout {
    int jq, qdi, qdo, jfi, tau, qdd;  // joint_q [nc], internal qd in / out [nd], joint_f internal, tau, qdd [nd]
    int qdp;                          // public joint_qd [nd] (stays in LDS across the substeps of a rollout)
    int qcom, org;                    // body COM world position [3][nb], solve origin [3][nb]
    int S;                            // motion subspace columns [6][nd]
    int Is;                           // spatial inertia in the solve frame [36][nb]
    int vs, as, fs, ft;               // v_s, a_s, (f_b - f_g), total subtree wrench per joint [6][nb] each
    int bfx;                          // external wrench buffer body_f_ext [6][nb]
    int cw;                           // contact wrenches [CW_FLOATS][np*cpp]        (union with P/H)
    int P, H;                         // P[b][jl] = I_b S_j [6][nb][W];  H / L [nd][W]
    int Ic, Pd;                       // tree-structured mass matrix (in the P region): composite inertias [36][nb], I^c S_d [6][nd]
    int rows;
};
// tree: the tree-structured mass matrix is in use (nt_featherstone_params.dense_mass_matrix == 0 on a model fs_tree_ok accepts): the
// solve then needs the composite inertias + I^c S (36 nb + 6 nd rows) and the nnz packed entries of H instead of the dense P
// (6 nb W) and H (nd W) -- 948 rows less on the quadruped, which together with the uniform-parameter tile lets 16 environments share
// a CU (round 6).  Host launch code and kernels must pass the same flag.
__host__ __device__ inline int fs_tree_nnz_bound(const nt_model& m) { return m.nd * (m.nd + 1) / 2; }
__host__ __device__ inline FsLayout make_fs_layout(const nt_model& m, const LdsLayout& L, const bool tree = false, const int tree_nnz = -1) {
    FsLayout F;
    int o = L.bd.off;  // (the body-derived tile of the XPBD layout is not part of this solver's working set)
    F.jq = o; o += m.nc;
    F.qdi = o; o += m.nd;
    F.qdo = o; o += m.nd;
    F.jfi = o; o += m.nd;
    F.tau = o; o += m.nd;
    F.qdd = o; o += m.nd;
    F.qdp = o; o += m.nd;
    F.qcom = o; o += 3 * m.nb;
    F.org = o; o += 3 * m.nb;
    F.S = o; o += 6 * m.nd;
    F.Is = o; o += 36 * m.nb;
    F.vs = o; o += 6 * m.nb;
    F.as = o; o += 6 * m.nb;
    F.fs = o; o += 6 * m.nb;
    F.ft = o; o += 6 * m.nb;
    F.bfx = o; o += 6 * m.nb;
    F.cw = o;
    F.P = o;
    F.Ic = F.P;
    F.Pd = F.P + 36 * m.nb;
    const int pregion = tree ? 36 * m.nb + 6 * m.nd : imax(6 * m.nb * m.max_art_dofs, 36 * m.nb + 6 * m.nd);
    F.H = F.P + pregion;
    const int hregion = tree ? (tree_nnz >= 0 ? tree_nnz : fs_tree_nnz_bound(m)) : m.nd * m.max_art_dofs;
    int solve = pregion + hregion;
    int contacts = NC_CW * m.np * m.cpp;
    // the fused rollout runs the collide phases on this union too (shape transforms / AABBs, pair counts, manifold polygon
    // scratch, staged candidates)
    LdsLayout tmp = L;
    int coll = place_collide_scratch(tmp, m, F.cw, false);
    o += imax(imax(solve, contacts), coll);
    F.rows = o;
    return F;
}
// the layout flag of a launch: the tree-structured mass matrix is requested and the model qualifies (KArgs is declared above)
__host__ __device__ inline bool fs_tree_ok(const nt_model& m);
__host__ __device__ inline bool fs_tree_mode(const KArgs& a) { return a.fp.dense_mass_matrix == 0 && fs_tree_ok(a.m); }
// block-shared ints behind the staged topology: joint ancestor, joint depth, articulation of joint (3 * nj), joint of
// each dof (nd), and per joint a bit mask of the joints on its root path, itself included (nj * ceil(nj / 32))
__host__ __device__ inline int fs_mask_words(const nt_model& m) { return (m.nj + 31) / 32; }
// then per joint whether the end-of-step refresh of descendant FREE / DISTANCE joints reaches it (nj) and whether any does (1)
__host__ __device__ inline int fs_topo_base_ints(const nt_model& m) { return 3 * m.nj + m.nd + 2 * m.nj * fs_mask_words(m) + m.nj + 1; }
// then the dof tree of the tree-structured factorisation (models with at most 64 dofs and 64 joints; larger ones take the dense
// path): 64-bit masks as (lo, hi) pairs -- per dof its strict descendants and strict ancestors, per level its dofs, per joint the
// joints of its subtree --, per dof its depth and the offset of its H row, {entry count, deepest level}, and the non-zero entries
// (i, j) of H's lower triangle packed as i | j << 8
__host__ __device__ inline bool fs_tree_ok(const nt_model& m) { return m.nd >= 1 && m.nd <= 64 && m.nj <= 64; }
__host__ __device__ inline int fs_topo_ints(const nt_model& m) {
    const int tree = fs_tree_ok(m) ? 4 * m.nd + 2 * (m.nd + 1) + 2 * m.nj + 2 * m.nd + 2 + m.nd * (m.nd + 1) / 2 : 0;
    return fs_topo_base_ints(m) + tree;
}

