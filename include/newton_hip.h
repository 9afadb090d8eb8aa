/* newton_hip.h -- C ABI of libnewton_hip.so: the MI355X (gfx950) batched rigid-body stepper that
 * drops in behind Newton's Model/State/Control/Contacts + CollisionPipeline.collide() + Solver.step().
 *
 * Each entry point names the reference interface it replaces (paths relative to /root/reference):
 *   nt_collide            <- CollisionPipeline.collide(state, contacts)          newton/_src/sim/collide.py:1765-2207
 *   nt_xpbd_step          <- SolverXPBD.step(state_in, state_out, control, contacts, dt)
 *                                                                                newton/_src/solvers/xpbd/solver_xpbd.py:329-862
 *   nt_semi_implicit_step <- SolverSemiImplicit.step(...)                        newton/_src/solvers/semi_implicit/solver_semi_implicit.py:123-217
 *   nt_xpbd_rollout       <- the CUDA-graph-captured simulate() loop             newton/examples/basic/example_basic_urdf.py:117-141
 *                            (clear_forces + collide + step + swap, N substeps, one launch)
 *   nt_clear_forces       <- State.clear_forces()                                newton/_src/sim/state.py:189-200
 *   nt_eval_fk            <- newton.eval_fk(model, joint_q, joint_qd, state)     newton/_src/sim/articulation.py:500-573
 *   nt_pack_aos / nt_unpack_aos <- the implicit AoS wp.array views of Model/State/Contacts
 *                                                                                newton/_src/sim/state.py:113-171, contacts.py:227-277
 *   nt_contacts_export    <- the flat, atomically-appended Contacts arrays       newton/_src/sim/collide.py:166-254
 *
 * Conventions (identical to Newton's, docs/concepts/conventions.rst:105-146): transforms are
 * (px,py,pz,qx,qy,qz,qw); spatial vectors are (linear, angular); body_qd linear part is the COM
 * velocity in world frame; contact normals point shape0 -> shape1; contact points are body-frame.
 *
 * Memory: every pointer below is a DEVICE pointer owned by the caller (the Python host allocates
 * them as torch tensors; any other host can use hipMalloc).  The library keeps no global state and
 * never allocates.  All entry points enqueue work on `stream` and return without synchronising.
 *
 * Layout: env-major SoA.  A quantity with C components for slot s (body / joint / dof / shape /
 * contact slot) of environment e lives at  base[(c * NSLOT + s) * env_stride + e]  -- environment
 * index fastest, so a wave64 touching one (component, slot) reads 256 contiguous bytes.
 * All environments share one topology (same bodies / joints / shapes / candidate pairs -- what
 * ModelBuilder.replicate() produces); parameters may differ per environment.
 *
 * Return value: 0 = ok, <0 = error (see nt_error_string).
 */
#ifndef NEWTON_HIP_H
#define NEWTON_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t nt_status;
#define NT_OK 0
#define NT_ERR_INVALID_ARG (-1)
#define NT_ERR_LAUNCH (-2)
#define NT_ERR_UNSUPPORTED (-3)

#define NT_MAX_CONTACTS_PER_PAIR 5   /* <=4 analytic (narrow_phase.py:871) or <=5 manifold (multicontact.py:852-956) */
#define NT_CONTACT_FLOATS 17     /* point0[3] point1[3] offset0[3] offset1[3] normal[3] margin0 margin1 */
#define NT_BODY_PARAM_FLOATS 23  /* com[3] inv_mass inertia[9] inv_inertia[9] mass */
#define NT_JOINT_PARAM_FLOATS 14 /* X_p[7] X_c[7] */
#define NT_DOF_PARAM_FLOATS 11   /* axis[3] limit_lower limit_upper target_ke target_kd limit_ke limit_kd armature damping */
#define NT_SHAPE_PARAM_FLOATS 20 /* xform[7] scale[3] margin gap mu mu_torsional mu_rolling ke kd kf ka restitution */

/* Model: env-uniform topology + per-env parameters (Newton: newton/_src/sim/model.py:808-1364) */
typedef struct {
    int32_t env_count;   /* E  = Model.world_count */
    int32_t env_stride;  /* ES >= E, multiple of 64 */
    int32_t nb;          /* bodies per env */
    int32_t nj;          /* joints per env */
    int32_t nd;          /* dofs per env   (joint_qd) */
    int32_t nc;          /* coords per env (joint_q) */
    int32_t ntq;         /* joint_target_q entries per env (coords or dofs, model.py:1584-1591) */
    int32_t ns;          /* env-local shapes per env */
    int32_t ng;          /* global (world -1, static) shapes, shared by every env */
    int32_t np;          /* candidate shape pairs per env (Model.shape_contact_pairs, one env's slice) */
    int32_t cpp;         /* contact slots per pair: 4 (all pairs analytic) or 5 (some pair uses the convex manifold) */
    int32_t np_analytic; /* pairs [0, np_analytic) have an analytic primitive path, pairs [np_analytic, np) go through
                            MPR/GJK + manifold (narrow_phase.py:642-655,1004-1014): the reference appends all analytic
                            contacts before all convex ones, and the pair table is stored in that order */
    int32_t na;          /* articulations per env (SolverFeatherstone only; 0 otherwise) */
    int32_t max_art_dofs;/* widest articulation, in dofs (row width of the LDS-resident joint-space inertia H) */
    int32_t shape_local0;/* Newton shape id of env 0's first local shape: env-local shapes occupy ids
                            [shape_local0, shape_local0 + env_count*ns), global shapes sit before and / or after (gshape_id) */
    int32_t contact_scratch_in_hbm; /* 0: the per-contact correction records of the solvers live in LDS (default).
                            1: pair-heavy scenes whose records do not fit the CU's LDS keep them in nt_contacts.cw (HBM, L2
                            resident); the stepping kernels then run one environment per workgroup.  XPBD / collide only. */
    int32_t params_uniform; /* 1: body_param / joint_param / dof_param / shape_param hold the same values in every environment
                            (replicated worlds, newton.ModelBuilder.replicate without per-world randomisation).  The XPBD rollout
                            then keeps ONE block-shared copy per workgroup in LDS; 0 is always valid */
    /* topology, int32, env-uniform */
    const int32_t* body_flags;          /* [nb]   BodyFlags */
    const int32_t* joint_type;          /* [nj]   JointType */
    const int32_t* joint_enabled;       /* [nj] */
    const int32_t* joint_parent;        /* [nj]   env-local body index or -1 */
    const int32_t* joint_child;         /* [nj] */
    const int32_t* joint_q_start;       /* [nj]   env-local */
    const int32_t* joint_qd_start;      /* [nj] */
    const int32_t* joint_tq_start;      /* [nj]   joint_target_q_start */
    const int32_t* joint_lin_count;     /* [nj]   joint_dof_dim[:,0] */
    const int32_t* joint_ang_count;     /* [nj]   joint_dof_dim[:,1] */
    const int32_t* shape_body;          /* [ns+ng] env-local body or -1 */
    const int32_t* shape_type;          /* [ns+ng] GeoType */
    const int32_t* shape_flags;         /* [ns+ng] ShapeFlags */
    const int32_t* shape_group;         /* [ns+ng] collision group */
    const int32_t* pair_a;              /* [np] shape index (0..ns-1 local, ns..ns+ng-1 global), pair_a < pair_b in Newton ids */
    const int32_t* pair_b;              /* [np] */
    /* ordered incidence lists (CSR) used for deterministic, tid-ordered reductions */
    const int32_t* body_joint_start;    /* [nb+1] */
    const int32_t* body_joint_list;     /* [2*nj] (padded) joint*2 + (1 if body is the child else 0), ascending joint; parent entry first */
    const int32_t* body_pair_start;     /* [nb+1] */
    const int32_t* body_pair_list;      /* [2*np] (padded) pair*2 + (1 if the body owns pair_b's shape else 0), ascending pair */
    const int32_t* art_start;           /* [na+1] first env-local joint of each articulation (Model.articulation_start / _end) */
    /* convex-hull shapes (GeoType.CONVEX_MESH, support_function.py:152-171): env-uniform vertex table */
    const int32_t* shape_mesh_start;    /* [ns+ng] first vertex of the shape's hull in mesh_points, -1 for other types */
    const int32_t* shape_mesh_count;    /* [ns+ng] */
    const int32_t* gshape_id;           /* [ng] Newton shape ids of the global shapes, ascending */
    const float* mesh_points;           /* [V][3] unscaled vertices (AoS); the per-env shape scale is applied on the fly */
    const float* shape_mesh_bounds;     /* [ns+ng][6] unscaled min xyz, max xyz of the hull (local AABB = bounds * scale) */
    /* per-env parameters, float SoA */
    const float* body_param;            /* [NT_BODY_PARAM_FLOATS][nb][ES] */
    const float* gravity;               /* [3][ES] */
    const float* joint_param;           /* [NT_JOINT_PARAM_FLOATS][nj][ES] */
    const float* dof_param;             /* [NT_DOF_PARAM_FLOATS][nd][ES] */
    const float* shape_param;           /* [NT_SHAPE_PARAM_FLOATS][ns][ES] */
    const float* gshape_param;          /* [ng][NT_SHAPE_PARAM_FLOATS] (AoS, tiny, env-uniform) */
} nt_model;

/* State (newton/_src/sim/state.py:113-171) */
typedef struct {
    float* body_q;   /* [7][nb][ES] */
    float* body_qd;  /* [6][nb][ES] */
    float* body_f;   /* [6][nb][ES] */
    float* joint_q;  /* [nc][ES]   (may be NULL for maximal-coordinate solvers) */
    float* joint_qd; /* [nd][ES] */
    float* body_parent_f; /* [6][nb][ES] or NULL: extended attribute State.body_parent_f (state.py:77,156-163), the incoming
                           * joint wrench at the body COM, world frame; written into the output state by nt_xpbd_step
                           * (xpbd/kernels.py:2497-2544) and nt_featherstone_step / _rollout (featherstone/kernels.py:2371-2416) */
} nt_state;

/* Control (newton/_src/sim/control.py:31-68) */
typedef struct {
    const float* joint_f;         /* [nd][ES] */
    const float* joint_target_q;  /* [ntq][ES] */
    const float* joint_target_qd; /* [nd][ES] */
} nt_control;

/* Contact rows that do not live in the fixed slots: what the mesh-SDF and hydroelastic legs of CollisionPipeline.collide append
 * (collide.py:1999 -> narrow_phase.py:2838-3167 -> sdf_contact.py:1534-1990 / sdf_hydroelastic.py:905).  Newton's flat AoS
 * Contacts layout (contacts.py:227-277), grouped per world: world w owns rows [row_start[w], row_start[w + 1]), inside a world
 * the rows of one shape pair are consecutive and the pairs ascend in (shape0, shape1).  An inert row carries shape0 = shape1 = -1.
 * The solvers sum a body's rows in ascending row order through the per-world block lists (no float atomics):
 * body b of world w touches the row blocks body_blk_list[body_blk_start[w * (nb + 1) + b] .. body_blk_start[w * (nb + 1) + b + 1]),
 * entry i = { (first_row << 1) | side, row_count }, side 0: b is the body of shape0, 1: of shape1.  All pointers NULL = no such rows. */
typedef struct {
    const int32_t* row_start;       /* [env_count + 1] */
    const int32_t* shape0;          /* [cap] Newton global shape ids */
    const int32_t* shape1;
    const float* point0;            /* [cap][3] body frame */
    const float* point1;
    const float* offset0;           /* [cap][3] body frame */
    const float* offset1;
    const float* normal;            /* [cap][3] world, shape0 -> shape1 */
    const float* margin0;           /* [cap] */
    const float* margin1;
    const float* stiffness;         /* [cap] or NULL: Contacts.rigid_contact_stiffness / _damping / _friction of the rows */
    const float* damping;
    const float* friction_scale;
    const int32_t* body_blk_start;  /* [env_count * (nb + 1)] */
    const int32_t* body_blk_list;   /* [..][2] */
    float* cw;                      /* [cap][10] solver scratch: XPBD correction record of every row */
    float* impulse;                 /* [cap][6] or NULL: with nt_xpbd_report.contact_impulse, nt_xpbd_step accumulates the rows' weighted
                                     * impulses here (accumulate_weighted_contact_impulse, xpbd/kernels.py:2403-2461) -- the rows follow the
                                     * slot contacts in Contacts.force: force[n_slot_contacts + k] = impulse[k-th live row] / dt
                                     * (convert_contact_impulse_to_force, :2464-2494) */
    float* restitution;             /* [cap][14] or NULL: scratch of the restitution pass over the rows (required with
                                     * nt_xpbd_params.enable_restitution when rows exist; without it the pass covers the slots only) */
} nt_flat_rows;

/* Contacts (newton/_src/sim/contacts.py:227-277), fixed slots: slot = pair * cpp + k.
 * Unused slots carry shape0 = shape1 = -1, which every reference consumer skips
 * (xpbd/kernels.py:2201, semi_implicit/kernels_contact.py:425). */
typedef struct {
    int32_t* shape0;      /* [np*cpp][ES] Newton global shape id or -1 */
    int32_t* shape1;      /* [np*cpp][ES] */
    float* data;          /* [NT_CONTACT_FLOATS][np*cpp][ES] */
    int32_t* env_count;   /* [ES] contacts emitted per env (== per-env slice of rigid_contact_count) */
    uint8_t* pair_hit;    /* [np][ES] 1 if the pair passed the broad phase (candidate pair set, per env) */
    float* cw;            /* 15 * np*cpp * ES floats of solver scratch, only when nt_model.contact_scratch_in_hbm (else NULL); layout
                           * internal to the kernels: one contiguous record per (environment, contact slot) */
    float* cr;            /* env_count * np*cpp * 32 floats or NULL: scratch of the pair-heavy fused rollout (nt_model.contact_scratch_in_hbm):
                           * the contact records of a substep as one 128-byte line per (environment, slot) -- that tile runs one
                           * environment per workgroup, where the env-major `data` costs a cache line per float (measured: 150 GB of
                           * HBM traffic per launch on config C5's geometry, 30 x the algorithmic bytes).  With it the substeps before the
                           * last write no Contacts buffer at all; shape0 / shape1 / data receive the last substep's contacts as always */
    const float* prop;    /* [3][np*cpp][ES] or NULL: per-contact stiffness, damping, friction scale
                           * (Contacts.rigid_contact_stiffness / _damping / _friction, contacts.py:227-277); a value > 0
                           * overrides the shape-material ke / kd and scales mu in eval_body_contact
                           * (semi_implicit/kernels_contact.py:452-459) -- SolverSemiImplicit and SolverFeatherstone; SolverXPBD
                           * ignores them like the reference (solver_xpbd.py:619-651) */
    /* optional outputs of nt_collide, indexed by Newton's global shape id: the world transform and the gap-widened world AABB of
     * every shape as compute_shape_aabbs leaves them (collide.py:283-472: geom_xform, aabb_lower / aabb_upper) -- what the
     * stages outside the tiles (candidate pairs of the SDF legs, their narrow phases) read.  NULL = not written */
    float* world_xform;      /* [shape_count][7] */
    float* world_aabb_lower; /* [shape_count][3] */
    float* world_aabb_upper; /* [shape_count][3] */
    nt_flat_rows flat;       /* rows of the SDF legs (all NULL when the model has none) */
} nt_contacts;

typedef struct {
    int32_t iterations;
    float joint_linear_relaxation, joint_angular_relaxation;
    float joint_linear_compliance, joint_angular_compliance;
    float rigid_contact_relaxation;
    int32_t rigid_contact_con_weighting;
    float angular_damping;
    int32_t enable_restitution; /* apply_rigid_restitution after the iterations (xpbd/kernels.py:2583-2728) */
    int32_t compute_body_velocity_from_position_delta; /* update_body_velocities after the iterations (xpbd/kernels.py:2547-2579;
                                  the SolverXPBD attribute of the same name, off by default) */
} nt_xpbd_params;

/* Optional reporting buffers of nt_xpbd_step (solver_xpbd.py:368-386): NULL members are skipped. */
typedef struct {
    float* contact_impulse; /* [6][np*cpp][ES] out: per contact slot, the 1/N-weighted impulse on shape0's body accumulated
                             * over the iterations (xpbd/kernels.py:2398-2461); nt_contacts_export_force turns it into
                             * Contacts.force */
    float* joint_impulse;   /* [6][nj][ES] scratch: per-joint child-side impulse; required when s_out->body_parent_f is set
                             * and the model has joints */
} nt_xpbd_report;

typedef struct {
    float angular_damping;
    float friction_smoothing;
    float joint_attach_ke, joint_attach_kd;
} nt_semi_implicit_params;

/* SolverFeatherstone(model, angular_damping=0.05, friction_smoothing=1.0, ...) solver_featherstone.py:136-146.
 * update_mass_matrix_interval is fixed at 1; use_tile_gemm / fuse_cholesky have no meaning here (H never leaves LDS). */
typedef struct {
    float angular_damping; /* accepted like the reference constructor; the reference step does not use it either */
    float friction_smoothing;
    /* SolverFeatherstone(update_mass_matrix_interval = k) (solver_featherstone.py:141,767): substep s of a call rebuilds
     * P / H and refactorises when mass_matrix_cache is NULL, force_update != 0 (first substep only: the reference's
     * _mass_matrix_dirty), or (step_index + s) % k == 0; the Cholesky factor ([nd * max_art_dofs][ES], env-major) is stored
     * to / reloaded from mass_matrix_cache in between.  Zero-initialised fields = rebuild every step (k = 1). */
    int32_t update_mass_matrix_interval;
    int32_t step_index;
    int32_t force_update;
    float* mass_matrix_cache;
    /* 0 (default): tree-structured mass matrix -- composite rigid body inertias (H_ij = S_j^T I^c_i S_i for dof j on dof i's root
     * path, structural zeros elsewhere) and the sparse L^T D L factorisation that follows the dof tree (no fill-in), worked level by
     * level by the whole workgroup; within the 1e-5 single-step contract of the reference's dense path, not bit-identical to it.
     * 1: the reference's own operation order -- H = J^T (M J) over the dense lower triangle, dense_cholesky, dense_subs
     * (kernels.py:1466-1501,1690-1797), the same role use_tile_gemm / fuse_cholesky play in the reference constructor. */
    int32_t dense_mass_matrix;
} nt_featherstone_params;

typedef struct {
    int32_t broad_phase;  /* 0 explicit pairs (default), 1 nxn, 2 sap -- all emit the same per-env pair set */
    int32_t envs_per_block; /* 0 = auto (widest tile that fits LDS and still fills the CUs; 1 for scenes over 20 KB of LDS per env); otherwise 1, 8, 16, 32 or 64 */
} nt_collide_params;

/* -------- hot path -------- */
nt_status nt_clear_forces(const nt_model* m, nt_state* s, void* stream);
nt_status nt_collide(const nt_model* m, const nt_state* s, nt_contacts* c, const nt_collide_params* p, void* stream);
nt_status nt_xpbd_step(const nt_model* m, const nt_xpbd_params* p, nt_state* s_in, nt_state* s_out,
                       const nt_control* ctrl, const nt_contacts* c /*nullable*/, float dt, int32_t envs_per_block,
                       const nt_xpbd_report* report /*nullable*/, void* stream);
nt_status nt_semi_implicit_step(const nt_model* m, const nt_semi_implicit_params* p, nt_state* s_in, nt_state* s_out,
                                const nt_control* ctrl, const nt_contacts* c /*nullable*/, float dt,
                                int32_t envs_per_block, void* stream);
/* SolverFeatherstone.step (newton/_src/solvers/featherstone/solver_featherstone.py:462-1066): advances joint_q / joint_qd and
 * rebuilds body_q / body_qd of s_out; like the reference it also refreshes s_in->body_q from s_in->joint_q (FK).
 * Scope: PRISMATIC, REVOLUTE, BALL, FIXED, root FREE, D6 with <= 1 angular axis; child body of env-local joint j is body j. */
nt_status nt_featherstone_step(const nt_model* m, const nt_featherstone_params* p, nt_state* s_in, nt_state* s_out,
                               const nt_control* ctrl, const nt_contacts* c /*nullable*/, float dt,
                               int32_t envs_per_block, void* stream);
/* substeps x {clear_forces; collide; featherstone step; swap} in one launch (result in s0 for even substeps, s1 for odd) */
nt_status nt_featherstone_rollout(const nt_model* m, const nt_featherstone_params* p, const nt_collide_params* cp,
                                  nt_state* s0, nt_state* s1, const nt_control* ctrl, nt_contacts* c, float dt,
                                  int32_t substeps, void* stream);
/* substeps x {clear_forces; collide; xpbd step; swap}: the result is in s0 when substeps is even, s1 when odd,
 * exactly like the reference loop's pointer swap. */
nt_status nt_xpbd_rollout(const nt_model* m, const nt_xpbd_params* p, const nt_collide_params* cp, nt_state* s0,
                          nt_state* s1, const nt_control* ctrl, nt_contacts* c, float dt, int32_t substeps, void* stream);
/* The launch shape nt_xpbd_rollout would take for this model: out = {environments per workgroup, workgroup size, minimum
 * waves per SIMD, uniform-parameter tile (nt_model.params_uniform), bit 0 convex code present | bit 1 pair-heavy tile};
 * the kernel is xpbd_rollout_kernel<out[0] + 256 * out[3], convex, pair-heavy, out[1], out[2]> (profilers print that name) */
nt_status nt_xpbd_rollout_shape(const nt_model* m, const nt_xpbd_params* p, const nt_collide_params* cp, int32_t out[5]);

/* -------- boundary helpers -------- */
nt_status nt_eval_fk(const nt_model* m, const float* joint_q /*[nc][ES]*/, const float* joint_qd /*[nd][ES]*/,
                     nt_state* out, void* stream);
/* RL-style reset (newton/_src/solvers/solver.py:344-375, core/reset.py:13-60): for every env whose world_mask byte is
 * non-zero, overwrite all of dst's state arrays with src's (e.g. a default state).  world_mask: [env_count] uint8. */
nt_status nt_state_reset(const nt_model* m, nt_state* dst, const nt_state* src, const uint8_t* world_mask, void* stream);
/* AoS [E*nslot][ncomp] (Newton flat array) <-> SoA [ncomp][nslot][ES] */
nt_status nt_pack_aos(const float* aos, float* soa, int32_t ncomp, int32_t nslot, int32_t env_count, int32_t env_stride,
                      void* stream);
nt_status nt_unpack_aos(const float* soa, float* aos, int32_t ncomp, int32_t nslot, int32_t env_count, int32_t env_stride,
                        void* stream);
/* Compact the fixed-slot contacts into Newton's flat append order (env, pair, sub-contact).
 * out_* are AoS arrays with capacity `cap`; out_count[0] receives the total (it keeps counting past cap,
 * like the reference's atomic counter, collide.py:176-177). Order: every env's analytic contacts (env, pair, k), then every env's convex contacts, like the
 * reference's two narrow-phase launches. Uses `scan_tmp` ([4*(env_count+1)] int32) as scratch. */
nt_status nt_contacts_export(const nt_model* m, const nt_contacts* c, int32_t cap, int32_t* out_count, int32_t* out_shape0,
                             int32_t* out_shape1, float* out_point0, float* out_point1, float* out_offset0,
                             float* out_offset1, float* out_normal, float* out_margin0, float* out_margin1,
                             int32_t* scan_tmp, void* stream);
/* SolverXPBD.update_contacts (solver_xpbd.py:864-921, xpbd/kernels.py:2464-2494): Contacts.force[i] = impulse / dt for the
 * i-th exported contact (same order as nt_contacts_export), zero beyond the count.  out_force: AoS [cap][6]. */
nt_status nt_contacts_export_force(const nt_model* m, const nt_contacts* c, const float* contact_impulse, float dt,
                                   int32_t cap, float* out_force, int32_t* scan_tmp, void* stream);

/* -------- frame-to-frame contact matching (newton/_src/geometry/contact_match.py:266-391 match + resolve, :442-480 save) --------
 * History of the previous frame in the fixed-slot layout: world-space midpoint 0.5 (world(point0) + world(point1)) and normal of
 * every live slot.  nt_contacts_match writes, per slot of the CURRENT contacts: the matched previous slot index (same pair,
 * closest midpoint within pos_threshold whose normal passes dot >= normal_dot_threshold; one previous contact is claimed by at
 * most one new contact: smallest distance, ties by the smaller sub-contact index), -1 = the pair had no contact last frame
 * (MATCH_NOT_FOUND), -2 = no candidate within the thresholds or the race was lost (MATCH_BROKEN).  reset_world_mask [env_count]
 * (nullable): contacts of those worlds report -1.  Call order per frame: collide -> nt_contacts_match -> ... ->
 * nt_contacts_save_history (with the state the contacts were generated on). */
typedef struct {
    float* prev_pos_world;  /* [3][np*cpp][ES] */
    float* prev_normal;     /* [3][np*cpp][ES] */
    uint8_t* prev_live;     /* [np*cpp][ES] */
    float* prev_body_frame; /* [12][np*cpp][ES] or NULL: sticky mode only -- point0, point1, offset0, offset1 of the record used */
} nt_contact_history;
nt_status nt_contacts_match(const nt_model* m, const nt_state* s, const nt_contacts* c, const nt_contact_history* h,
                            float pos_threshold /*0.0005*/, float normal_dot_threshold /*0.995*/, const uint8_t* reset_world_mask,
                            int32_t* match_index /*[np*cpp][ES]*/, void* stream);
/* ContactMatcher.replay_matched (contact_match.py:530-562,933-996; CollisionPipeline(contact_matching="sticky")): matched
 * contacts that still touch get last frame's body-frame points / offsets and normal back (call between match and save_history) */
nt_status nt_contacts_replay_matched(const nt_model* m, const nt_state* s, nt_contacts* c, const nt_contact_history* h,
                                     const int32_t* match_index /*[np*cpp][ES], slot space*/, void* stream);
nt_status nt_contacts_save_history(const nt_model* m, const nt_state* s, const nt_contacts* c, nt_contact_history* h, void* stream);

/* -------- standalone broad phases (newton.geometry.BroadPhaseAllPairs / BroadPhaseSAP / BroadPhaseExplicit) --------
 * World-aware candidate-pair search on arbitrary AABB arrays (newton/_src/geometry/broad_phase_nxn.py:29-535,
 * broad_phase_sap.py:44-848, broad_phase_common.py:20-388).  All arrays are device pointers in Newton's flat AoS layout.
 * Output: canonical pairs (min, max) appended in unspecified order (the reference appends atomically too);
 * count[0] is ADDED to (zero it first) and keeps counting past `cap` (broad_phase_common.py:204-218). */
typedef struct {
    const float* lower;           /* [n][3] AABB lower bounds */
    const float* upper;           /* [n][3] */
    const float* gap;             /* [n] per-shape cutoff added to both sides, or NULL when the AABBs are pre-expanded */
    const int32_t* group;         /* [n] collision groups (0 off, >0 exclusive, <0 collides with other groups) */
    const int32_t* world;         /* [n] world ids, -1 = shared by every world */
    const int32_t* filter_pairs;  /* [num_filter_pairs][2] excluded pairs, canonical and lexicographically sorted; may be NULL */
    int32_t num_filter_pairs;
    int32_t include_static_kinematic_pairs; /* 0: drop pairs whose two bodies are static / kinematic (needs shape_body) */
    const int32_t* shape_body;    /* [n] or NULL (no immovable filtering) */
    const int32_t* body_flags;    /* [bodies] or NULL (static-static filtering only) */
} nt_broadphase_in;

/* index_map / slice_ends: precompute_world_map (broad_phase_common.py:271-388): per world its shapes followed by the shared
 * ones, plus a trailing segment holding only the shared shapes; num_regular_worlds = segments - 1. */
nt_status nt_broadphase_nxn(const nt_broadphase_in* in, const int32_t* index_map, const int32_t* slice_ends, int32_t segments,
                            int32_t num_regular_worlds, int32_t map_len, int32_t* pairs /*[cap][2]*/, int32_t* count,
                            int32_t cap, void* stream);
/* sorted_map: index_map with every segment ordered by ascending (lower.x - gap); same pair set as nt_broadphase_nxn */
nt_status nt_broadphase_sap(const nt_broadphase_in* in, const int32_t* sorted_map, const int32_t* slice_ends, int32_t segments,
                            int32_t num_regular_worlds, int32_t map_len, int32_t* pairs, int32_t* count, int32_t cap,
                            void* stream);
/* BroadPhaseSAP.launch entirely on the device (broad_phase_sap.py:395-848): projects the gap-widened AABBs on the reference's
 * fixed axis normalize(0.5935, 0.7790, 0.1235), sorts every world segment in LDS (one workgroup per segment, bitonic, stable on
 * the map position) and sweeps.  index_map / slice_ends as for nt_broadphase_nxn; max_segment = longest segment (<= 4096, else
 * NT_ERR_UNSUPPORTED); sorted_map [map_len] and projections [2][map_len] are caller-provided scratch / outputs. */
nt_status nt_broadphase_sap_device(const nt_broadphase_in* in, const int32_t* index_map, const int32_t* slice_ends,
                                   int32_t segments, int32_t num_regular_worlds, int32_t map_len, int32_t max_segment,
                                   int32_t* sorted_map, float* projections, int32_t* pairs, int32_t* count, int32_t cap,
                                   void* stream);
/* pair_list: [n_pairs][2] precomputed shape pairs (Model.shape_contact_pairs); only the AABB (and immovable) test remains */
nt_status nt_broadphase_explicit(const nt_broadphase_in* in, const int32_t* pair_list, int32_t n_pairs, int32_t* pairs,
                                 int32_t* count, int32_t cap, void* stream);

/* -------- sparse "texture" SDFs (newton/_src/geometry/sdf_texture.py) --------
 * TextureSDFData (:126-160) with the CUDA textures replaced by the plain arrays they hold: a coarse float grid sampled at the
 * subgrid corners, packed (subgrid_size+1)^3 blocks for the narrow band (float32 / uint16 / uint8, :487-690) and the
 * indirection slots (10-bit block coordinates, 0xFFFFFFFF = empty, 0xFFFFFFFE = "use the coarse grid", :44-45).
 * Built on the host by newton_amd.sdf (create_texture_sdf_from_mesh / _primitive). All pointers are device pointers. */
typedef struct {
    const float* coarse;       /* [cz+1][cy+1][cx+1] */
    const void* subgrid;       /* [tex_size]^3, z-major; element type by `quantization` */
    const uint32_t* slots;     /* [cx][cy][cz] */
    int32_t cx, cy, cz;        /* coarse cells per axis (= subgrids per axis) */
    int32_t tex_size;
    int32_t subgrid_size;      /* fine cells per subgrid edge (8) */
    int32_t quantization;      /* 4 float32, 2 uint16, 1 uint8 (QuantizationMode) */
    int32_t scale_baked;       /* the shape scale is already inside the SDF values */
    float box_lower[3], box_upper[3], inv_dx[3], voxel_size[3];
    float voxel_radius, min_value, value_range;
} nt_sdf;
/* texture_sample_sdf (:1129-1135) and the centred-difference gradient of the narrow phase (:1619-1697) at local points
 * [n][3]; either output may be NULL. */
nt_status nt_sdf_sample(const nt_sdf* sdf, const float* points, int32_t n, float* out_dist /*[n]*/, float* out_grad /*[n][3]*/,
                        void* stream);
/* texture_sample_sdf_hw (:1533-1538): the one-fetch sampler of the mesh-SDF narrow phase (value only) */
nt_status nt_sdf_sample_hw(const nt_sdf* sdf, const float* points, int32_t n, float* out_dist /*[n]*/, void* stream);
/* texture_sample_sdf_at_voxel (:1220-1228): exact value at integer fine-grid vertices ijk [n][3] (hydroelastic corner values) */
nt_status nt_sdf_sample_voxels(const nt_sdf* sdf, const int32_t* ijk, int32_t n, float* out_dist /*[n]*/, void* stream);

/* mesh_sdf_collision_kernel (sdf_contact.py:1098-1515, reduce_contacts=False): every edge of one shape of a pair against the
 * SDF of the other, both ways.  Flat Newton arrays; contacts are appended (atomic counter, which keeps counting past
 * `capacity`) as ContactData rows: out_pair = index into `pairs`, out_key = (edge << 2) | (mode << 1) (contact_data.py:60-90
 * sub-key), out_data = centre[3] (world), normal[3] (shape0 -> shape1), distance, margin0, margin1. */
typedef struct {
    const int32_t* pairs;            /* [pair_count][2] shape ids */
    int32_t pair_count;
    const float* shape_transform;    /* [S][7] world transform of every shape (body_q * shape_transform) */
    const float* shape_data;         /* [S][4] scale xyz, margin */
    const float* shape_gap;          /* [S] */
    const int32_t* shape_sdf_index;  /* [S] index into sdf_table or -1 */
    const nt_sdf* sdf_table;         /* [sdf_count] (device memory) */
    int32_t sdf_count;
    const int32_t* shape_edge_range; /* [S][2] (start, count) into the edge tables (Model.shape_edge_range) */
    const float* edge_centers;       /* [E][4] scaled local centre, radius      (Model.mesh_edge_centers) */
    const float* edge_halves;        /* [E][4] scaled local half vector, corner ownership code (Model.mesh_edge_halves) */
    int32_t* out_count;              /* [1], zero it first */
    int32_t* out_pair;               /* [capacity] */
    int32_t* out_key;                /* [capacity] */
    float* out_data;                 /* [capacity][9] */
    int32_t capacity;
    const int32_t* pair_count_device; /* optional: the live pair count in device memory (e.g. a broad phase's candidate
                                         counter, read by the kernel: no host round trip); pair_count then is the capacity of
                                         `pairs` and bounds it */
    /* world-region pairs (the collide pipeline, nt_sdf_candidate_pairs): `pairs` holds pairs_per_world entries per world of
     * which the first (pair_world_prefix[w + 1] - pair_world_prefix[w]) are live; the flat index f of the prefix walks them and
     * out_pair carries w * pairs_per_world + k.  out_blk [worlds * pairs_per_world][2] receives (raw row offset, row count) of
     * every live pair -- nt_mesh_sdf_collide_reduced only, whose rows of one pair form one block.  All NULL / 0 otherwise. */
    const int32_t* pair_world_prefix; /* [worlds + 1] */
    int32_t worlds, pairs_per_world;
    int32_t* out_blk;
    const uint8_t* pair_kind;         /* [worlds * pairs_per_world] or NULL: only pairs of kind 0 are processed (1 = hydroelastic pair,
                                         nt_hydro_pairs) */
    /* staged variant of nt_mesh_sdf_collide_reduced (same rows, bit for bit): the edge-independent part of every live pair is
     * evaluated one lane per pair, the edges that survive culling are compacted over ALL pairs into one list, searched one lane
     * per survivor, then reduced per pair that has any -- four dense launches instead of one workgroup per pair, and no counter
     * that every pair hits.  Scratch of the call.  The survivor list is cut into hit_stripe_count equal stripes, each filled by
     * the waves that map to it; hit_count[0] > 0 afterwards = that many survivors found their stripe full and were dropped (size
     * hit_capacity up).  With out_blk (world-region pairs) a pair's rows are written from the START OF ITS SURVIVOR BLOCK in the
     * out_* arrays -- `capacity` must be >= hit_capacity, out_count is not touched, out_blk carries (offset, count); without
     * out_blk the rows are appended through out_count as in the single kernel.  All NULL / 0: the single kernel.
     * `pairs` below counts worlds * pairs_per_world positions for world-region pairs, pair_count otherwise. */
    int32_t* hit_count;               /* [4] dropped survivors, runnable pairs, stripes in use (<= hit_stripe_count: a small call uses
                                         fewer, each hit_capacity / that long), spare */
    int32_t* hit_stripes;             /* [hit_stripe_count * 16] fill of every stripe (one counter per 64 bytes; zeroed by the call) */
    int32_t hit_stripe_count;
    int32_t hit_capacity;
    int32_t* hit_pair;                /* [hit_capacity] position in `pairs` */
    int32_t* hit_fp;                  /* [hit_capacity] (edge << 2) | (mode << 1), -1 once the search rejected it */
    float* hit_rec;                   /* [hit_capacity][8] midpoint value, then world point, distance, normal */
    int32_t* hit_blk;                 /* [pairs][2 modes][2] (offset, count) of each (pair, mode)'s block in the list */
    float* unit_ctx;                  /* [pairs][2][24] per runnable pair and mode: transform into the SDF's space, thresholds, edges */
} nt_mesh_sdf_args;
nt_status nt_mesh_sdf_collide(const nt_mesh_sdf_args* args, void* stream);

/* mesh_sdf_collision_global_reduce_kernel + GlobalContactReducer + export_reduced_contacts_kernel (sdf_contact.py:1534-1990,
 * contact_reduction_global.py:1519-1752,2098-2290, deterministic packing): the same edge-vs-SDF contacts, reduced per shape
 * pair to the winners of 20 normal bins x (6 spatial extremes + deepest) and 100 voxel-depth slots, roundoff twins dropped,
 * every survivor once.  One contiguous block of rows per pair, ascending fingerprint ((edge << 2) | (mode << 1)) inside the
 * block -- the order Newton's deterministic contact sort produces; out_data[3..5] is the normal after the reducer's
 * octahedral round trip (what the reference exports).  Needs the per-shape tables the reference's Model carries for the voxel
 * slots. */
typedef struct {
    const float* shape_aabb_lower;   /* [S][3] Model.shape_collision_aabb_lower (shape-local) */
    const float* shape_aabb_upper;   /* [S][3] Model.shape_collision_aabb_upper */
    const int32_t* shape_voxel_res;  /* [S][3] Model._shape_voxel_resolution (builder.py:11544-11570) */
    int32_t threads;                 /* workgroup size per pair: 64, 128 or 256 (0 = 256); pick ~ the edge count of a mesh */
    const float* shape_edge_radius_max; /* [S] or NULL: largest mesh_edge_centers radius (w component) among the shape's collision
                                           edges.  Lets the kernel skip a (pair, mode) whose edge carrier's local AABB, seen from the
                                           SDF shape, lies beyond the cull threshold of its longest edge -- no edge could pass
                                           edge culling (sdf_contact.py:1318-1340), so the result is unchanged */
    int32_t keep_all;                /* 1 = no reduction (CollisionPipeline(reduce_contacts=False), narrow_phase.py:3044,3097-3130 launching
                                        mesh_sdf_collision_kernel): every contact the edge search admits -- nt_mesh_sdf_collide's set --
                                        as one block per pair, ascending fingerprint, the search's own normal.  Staged variant with
                                        out_blk only (NT_ERR_UNSUPPORTED otherwise) */
} nt_contact_reduce_shapes;
nt_status nt_mesh_sdf_collide_reduced(const nt_mesh_sdf_args* args, const nt_contact_reduce_shapes* shapes, void* stream);

/* The reduction stage alone on an unreduced contact list (what export_and_reduce_contact_centered_two_spatial_depths is
 * called with, contact_reduction_global.py:1519-1535), grouped by shape pair: contacts [segment_start[k], segment_start[k+1])
 * belong to one pair; fingerprints are unique inside a pair and below 2^22.  Appends, per segment, the indices of the surviving
 * contacts in ascending fingerprint order and their exported normals. */
typedef struct {
    const int32_t* segment_start;    /* [segments + 1] */
    int32_t segments;
    const float* pos;                /* [n][3] world point */
    const float* normal;             /* [n][3] a -> b */
    const float* depth;              /* [n] */
    const int32_t* fp;               /* [n] fingerprint */
    const float* centered;           /* [n][3] point relative to the pair's midpoint */
    const float* inner;              /* [n] inner spatial depth */
    const float* outer;              /* [n] outer spatial depth */
    const float* local;              /* [n][3] point in the edge shape's frame */
    const float* aabb_lo;            /* [n][3] that shape's local AABB */
    const float* aabb_hi;            /* [n][3] */
    const int32_t* res;              /* [n][3] and voxel resolution */
    int32_t* out_count;              /* [1], zero it first */
    int32_t* out_index;              /* [capacity] index into the list */
    float* out_normal;               /* [capacity][3] */
    int32_t capacity;
} nt_contact_reduce_list;
nt_status nt_contacts_reduce_list(const nt_contact_reduce_list* args, void* stream);

/* write_contact (newton/_src/sim/collide.py:203-254) for ContactData rows that live outside the environment tiles -- the rows
 * nt_mesh_sdf_collide(_reduced) emits (out_pair, out_data): body-frame points and offsets, normal, margins in Newton's flat
 * Contacts layout, row i -> contact i (the input order is kept; a row beyond the pair's gap becomes the inert contact
 * shape0 = shape1 = -1 that eval_body_contact skips, where the reference would not have appended it). */
typedef struct {
    int32_t row_count;               /* rows to write (the capacity when row_count_device is given: rows past the live count
                                        are left untouched, like Newton's arrays past rigid_contact_count) */
    const int32_t* row_count_device; /* optional: live row count in device memory (nt_mesh_sdf_args.out_count) */
    const int32_t* row_pair;         /* [n] index into `pairs` (nt_mesh_sdf_args.out_pair) */
    const int32_t* pairs;            /* [P][2] shape ids */
    const float* row_data;           /* [n][9] centre, normal a -> b, distance, margin a, margin b */
    const float* body_q;             /* [B][7] State.body_q */
    const int32_t* shape_body;       /* [S] */
    const float* shape_gap;          /* [S] */
    int32_t* out_shape0;             /* [n] Contacts.rigid_contact_shape0 ... */
    int32_t* out_shape1;
    float* out_point0;               /* [n][3] */
    float* out_point1;
    float* out_offset0;
    float* out_offset1;
    float* out_normal;
    float* out_margin0;              /* [n] */
    float* out_margin1;
} nt_contact_rows;
nt_status nt_contact_rows_write(const nt_contact_rows* args, void* stream);

/* eval_body_contact (newton/_src/solvers/semi_implicit/kernels_contact.py:381-556, force_in_world_frame=False) on Newton's
 * flat arrays, argument for argument: penalty force of every contact row added to body_f (float atomics, like the reference).
 * Lets SolverSemiImplicit / SolverFeatherstone consume contacts that are not in the fixed-slot tiles: evaluate into State.body_f
 * after clear_forces, then step. */
typedef struct {
    const float* body_q;             /* [B][7] */
    const float* body_qd;            /* [B][6] linear, angular */
    const float* body_com;           /* [B][3] */
    const float* shape_ke;           /* [S] Model.shape_material_ke ... */
    const float* shape_kd;
    const float* shape_kf;
    const float* shape_ka;
    const float* shape_mu;
    const int32_t* shape_body;       /* [S] */
    const int32_t* contact_count;    /* optional [1] device counter; contact_max bounds it */
    int32_t contact_max;
    const float* point0;             /* [n][3] */
    const float* point1;
    const float* normal;
    const int32_t* shape0;           /* [n] */
    const int32_t* shape1;
    const float* margin0;            /* [n] */
    const float* margin1;
    const float* contact_stiffness;  /* optional [n] (Contacts.rigid_contact_stiffness; 0 = use the shape materials) */
    const float* contact_damping;
    const float* contact_friction_scale;
    float friction_smoothing;
    float* body_f;                   /* [B][6] accumulated */
} nt_flat_contact_forces;
nt_status nt_eval_body_contact_flat(const nt_flat_contact_forces* args, void* stream);

/* HydroelasticSDF contact generation, unreduced (sdf_hydroelastic.py:905-1296 launch, :1982-2140 generate, :1823-1928 decode):
 * marching cubes on the iso-pressure surface p_a == p_b (p = -kh * signed depth) of every SDF pair; one contact per face with
 * the per-contact stiffness area * pressure / |separation| (penetrating) or margin_contact_area * k_a k_b / (k_a + k_b)
 * (inside the gap band) that Contacts.rigid_contact_stiffness carries to eval_body_contact
 * (semi_implicit/kernels_contact.py:453-455).  tri_range / flat_edge_verts: marching-cubes case tables (newton_amd.mc_tables).
 * Rows are appended: out_pair, out_key = voxel * 5 + face (the reference's face fingerprint), out_shapes = (shape_a, shape_b)
 * after the finer-SDF-is-B normalisation, out_data = centre[3] (world), normal[3] (a -> b), separation, stiffness, area, pressure. */
typedef struct {
    const int32_t* pairs;            /* [pair_count][2] */
    int32_t pair_count;
    const float* shape_transform;    /* [S][7] world */
    const float* shape_data;         /* [S][4] scale xyz, margin */
    const float* shape_gap;          /* [S] */
    const float* shape_kh;           /* [S] Model.shape_material_kh */
    const int32_t* shape_sdf_index;  /* [S] */
    const nt_sdf* sdf_table;
    int32_t sdf_count;
    const int32_t* tri_range;        /* [257] */
    const uint8_t* flat_edge_verts;  /* [n][2] */
    float margin_contact_area;       /* HydroelasticSDF.Config.margin_contact_area (1e-2) */
    float edge_clamp_min;            /* Config.mc_edge_clamp_min (0.02) */
    int32_t* out_count;
    int32_t* out_pair;
    int32_t* out_key;
    int32_t* out_shapes;             /* [capacity][2] */
    float* out_data;                 /* [capacity][10] */
    int32_t capacity;
    /* ---- nt_hydro_pairs only (the collide pipeline: world-region pairs as in nt_mesh_sdf_args) ---- */
    const int32_t* pair_world_prefix; /* [worlds + 1] */
    int32_t worlds, pairs_per_world;
    const uint8_t* pair_kind;        /* [worlds * pairs_per_world]: 1 = hydroelastic pair (both shapes HYDROELASTIC), others skipped */
    int32_t* out_pairs_normalized;   /* [worlds * pairs_per_world][2] or NULL: (shape_a, shape_b) after the finer-SDF-is-B swap */
    int32_t* out_blk;                /* [worlds * pairs_per_world][2]: (0, rows of the pair) */
    int32_t* out_rank;               /* [capacity] rank of the row inside its pair (rows of a pair are not contiguous) */
    float* out_stiffness;            /* [capacity] Contacts.rigid_contact_stiffness of the row; out_data is then [capacity][9]:
                                        centre, normal a -> b, margin-relative separation, 0, 0 (the pipeline's raw row) */
    /* ---- reduce_contacts = True (HydroelasticSDF.Config defaults; contact_reduction_hydroelastic.py): every pair's faces are
     * buffered, aggregated per normal bin (force, centre of pressure, depth-volume), optionally pruned voxel by voxel
     * (two strongest penetrating faces + the closest non-penetrating one), ranked in the pair's table (20 normal bins x
     * (6 spatial extremes + deepest) + 100 voxel slots, + 100 speculative voxel slots), and the winners leave with the bin's
     * aggregate stiffness |agg force| / sum of their depths and, with normal matching, normals rotated so that their weighted sum is
     * the aggregate force direction.  The non-deterministic variant of the reference evaluated in thread order (contact ids follow the
     * face order, sums the contact order).  The pair's rows are contiguous,
     * out_blk = (0, rows), out_rank = position in the pair's export order.  All 0 / NULL: unreduced. */
    int32_t reduce;                  /* bit 0: reduce, bit 1: pre_prune_contacts, bit 2: normal_matching, bit 3: anchor_contact (an
                                        extra row per normal bin at its centre of pressure, key 0x400000 | bin), bit 4: moment_matching
                                        (friction scales that preserve the bin's friction moment; implies anchors) */
    const float* shape_aabb_lower;   /* [S][3] Model.shape_collision_aabb_lower (shape-local) */
    const float* shape_aabb_upper;   /* [S][3] */
    const int32_t* shape_voxel_res;  /* [S][3] Model._shape_voxel_resolution */
    int32_t* face_count;             /* [2] scratch: faces buffered, pairs whose chunk list overflowed (zero both first) */
    float* face_rec;                 /* [face_capacity][12] scratch */
    int32_t face_capacity;
    float* out_friction;             /* [capacity] or NULL: Contacts.rigid_contact_friction of the row (1 for reduced rows) */
    /* ---- nt_hydro_pairs with reduce: the same pipeline as dense device stages (optional; stage_count NULL: one workgroup per pair).
     * A wave per candidate pair queues the surviving 8^3 blocks of the finer SDF, a wave per (pair, block) runs the octree levels
     * 4 / 2 / 1 and marching cubes, a workgroup per pair with faces reduces them; faces, ids, order and rows are unchanged.
     * Scratch only, nothing to initialise (the call zeroes stage_count); a full queue / chunk pool drops whole pairs / blocks and
     * counts them in stage_count[2] (report it like face_count[1]). */
    int32_t* stage_count;            /* [8]: queue items, chunk records, dropped pairs / blocks, -, pairs with blocks, - */
    int32_t* stage_queue;            /* [stage_queue_capacity][2] (pair = world * pairs_per_world + k, block of shape B) */
    int32_t stage_queue_capacity;
    int32_t stage_chunk_capacity;
    int32_t* stage_pair;             /* [worlds * pairs_per_world][2] (first queue item, items) of every hydroelastic pair */
    int32_t* stage_item;             /* [stage_queue_capacity][2] (first chunk record, records) of every queue item */
    int32_t* stage_chunk;            /* [stage_chunk_capacity][4] (first face, faces (negative: not stored), buffered contacts, voxels)
                                        per 64 iso voxels of a block */
    int32_t* stage_active;           /* [worlds * pairs_per_world] the pairs that queued blocks (arrival order; each writes its own rows) */
} nt_hydro_args;
nt_status nt_hydro_collide(const nt_hydro_args* args, void* stream);
/* HydroelasticSDF.launch (sdf_hydroelastic.py:905-1296, reduce_contacts=False) inside the collide pipeline: SAT of the SDF boxes,
 * the octree over the finer SDF's blocks (8 / 4 / 2 / 1 voxels, pressure-interval pruning :1444-1700) in LDS, marching cubes
 * on the surviving voxels, one contact row per face with its rank inside the pair (nt_sdf_rows_finalize places the rows). */
nt_status nt_hydro_pairs(const nt_hydro_args* args, void* stream);

/* -------- the mesh-SDF leg of CollisionPipeline.collide as device stages (csrc/nt_sdf_pipeline.hip) --------
 * collide.py:1999 -> narrow_phase.py:2838-3167: pairs whose two shapes carry a texture SDF and collision edges (not box-box,
 * narrow_phase.py:620-655) go through the mesh-mesh SDF kernel with the global contact reduction, their contacts are appended
 * after the primitive / GJK-MPR ones.  Stage order per collide():
 *   nt_collide (exports nt_contacts.world_xform / world_aabb_*)  ->  nt_sdf_candidate_pairs  ->  nt_mesh_sdf_collide_reduced
 *   (world-region pairs, block records)  ->  nt_sdf_rows_finalize  ->  rows in nt_contacts.flat, consumed by nt_xpbd_step and,
 *   through nt_flat_rows_forces, by nt_semi_implicit_step / nt_featherstone_step.
 * Nothing returns to the host and nothing depends on atomic arrival order: two runs give bit-identical rows. */
typedef struct {
    int32_t env_count, env_stride;   /* worlds, env-major stride of the state arrays */
    int32_t nb, ns;                  /* bodies / env-local shapes per world (nt_model) */
    int32_t shape_local0;            /* Newton shape id of world 0's first env-local shape */
    int32_t template_pairs;          /* SDF shape pairs one world can have ... */
    const int32_t* template_pair;    /* ... [template_pairs][2] template shape ids (< ns: env-local, else ns + rank of a global
                                        shape), ascending in the Newton ids they map to: group / filter / same-body rules and the
                                        SDF-pair routing applied on the host once (they are env-uniform) */
    const int32_t* gshape_id;        /* [ng] Newton ids of the global (world -1) shapes */
    const int32_t* shape_body;       /* [ns] env-local body of every env-local shape (-1 static) */
    const float* shape_gap;          /* [shape_count] Model.shape_gap, Newton ids */
    int32_t pairs_per_world;         /* capacity of a world's candidate list */
    const uint8_t* template_kind;    /* [template_pairs] or NULL (all 0): 0 = mesh-SDF edge contacts, 1 = hydroelastic (both shapes carry
                                        ShapeFlags.HYDROELASTIC: narrow_phase.py:531-538) */
    uint8_t* world_pair_kind;        /* [env_count * pairs_per_world] out of nt_sdf_candidate_pairs when template_kind is given */
} nt_sdf_scene;
/* candidate pairs of every world: world_pairs[(w * pairs_per_world + k)][2] = Newton shape ids (shape0 < shape1), ascending;
 * pair_count[w] keeps counting past pairs_per_world (overflow check), pair_prefix[env_count + 1] = exclusive scan of the
 * clamped counts = the flat pair index the narrow phase walks. */
nt_status nt_sdf_candidate_pairs(const nt_sdf_scene* sc, const float* aabb_lower, const float* aabb_upper, int32_t* world_pairs,
                                 int32_t* pair_count, int32_t* pair_prefix, void* stream);
typedef struct {
    const int32_t* pair_count;   /* [env_count] */
    const int32_t* world_pairs;  /* [env_count * pairs_per_world][2] */
    const int32_t* blk;          /* [env_count * pairs_per_world][2] (raw row offset, row count) per pair (nt_mesh_sdf_args.out_blk) */
    int32_t* pair_row;           /* [env_count * pairs_per_world] out: row offset of the pair inside its world */
    int32_t* row_start;          /* [env_count + 1] out: nt_flat_rows.row_start */
    const int32_t* raw_count;    /* [1] raw rows appended by the narrow phase */
    const int32_t* raw_pair;     /* [raw_capacity] world * pairs_per_world + k */
    const int32_t* raw_key;      /* [raw_capacity] fingerprint (edge << 2 | mode << 1) */
    const float* raw_data;       /* [raw_capacity][9] centre, normal a -> b, distance, margin a, margin b */
    int32_t raw_capacity, row_capacity;
    int32_t* shape0;             /* [row_capacity] out: the nt_flat_rows arrays */
    int32_t* shape1;
    float* point0;
    float* point1;
    float* offset0;
    float* offset1;
    float* normal;
    float* margin0;
    float* margin1;
    int32_t* key;                /* [row_capacity] or NULL: the row's fingerprint (tests, deterministic sort key) */
    const int32_t* raw_rank;     /* [raw_capacity] or NULL: rank of the raw row inside its pair, for rows whose pair kind is 1 (their
                                    blocks are not contiguous: nt_hydro_pairs); kind-0 rows use raw index - block offset */
    const float* raw_stiffness;  /* [raw_capacity] or NULL: per-contact stiffness of kind-1 rows */
    const float* raw_friction;   /* [raw_capacity] or NULL: per-contact friction scale of kind-1 rows (reduced hydroelastic rows: 1) */
    float* stiffness;            /* [row_capacity] or NULL: out, nt_flat_rows.stiffness / damping / friction_scale: the hydroelastic */
    float* damping;              /*   rows carry their stiffness and zero damping / friction scale (ContactData defaults), */
    float* friction_scale;       /*   mesh-SDF rows zeros (collide.py:196-199) */
    int32_t raw_base;            /* raw rows [0, raw_base) are the blocks the staged narrow phase placed at the start of each pair's
                                    survivor block (nt_mesh_sdf_args.hit_capacity; gaps in between), rows appended through raw_count
                                    start at raw_base (the counter is initialised to it); 0: every raw row was appended */
    const float* raw_radius;     /* [raw_capacity][2] or NULL: effective radii (a, b) the export hands to write_contact for the rows of
                                    pair kind 3 (nt_mesh_triangle_args.out_radius: a sphere / capsule partner); every other row: 0 */
} nt_sdf_rows_io;
/* final row ranges (world-major, pairs ascending, rows in fingerprint order), write_contact (collide.py:166-254) of every raw
 * row at its final position, and the per-body row-block lists.  body_q: State.body_q, env-major [7][nb][ES].
 * world_rows [env_count] scratch; body_blk_start [env_count * (nb + 1)], body_blk_list [env_count * 2 * pairs_per_world][2]. */
nt_status nt_sdf_rows_finalize(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, int32_t* world_rows,
                               int32_t* body_blk_start, int32_t* body_blk_list, void* stream);
/* eval_body_contact (semi_implicit/kernels_contact.py:381-556) over the flat rows, every body's wrenches summed in ascending
 * row order (the reference: float atomics) and ADDED to body_f: call between clear_forces and nt_semi_implicit_step /
 * nt_featherstone_step on a force buffer the step then reads as State.body_f. */
typedef struct {
    const float* body_q;          /* [7][nb][ES] */
    const float* body_qd;         /* [6][nb][ES] */
    const float* body_com;        /* nt_model.body_param (rows 0..2 = COM), [..][nb][ES] */
    const float* shape_material;  /* [shape_count][5] ke, kd, kf, ka, mu by Newton shape id */
    float friction_smoothing;
    float* body_f;                /* [6][nb][ES] accumulated */
} nt_flat_force_params;
nt_status nt_flat_rows_forces(const nt_sdf_scene* sc, const nt_flat_rows* rows, const nt_flat_force_params* p, void* stream);

/* Frame-to-frame matching of the SDF legs' rows (newton/_src/geometry/contact_match.py:266-391 match + resolve, :442-480 save,
 * :530-562 sticky replay; called from CollisionPipeline.collide, collide.py:2033-2135, like nt_contacts_match for the slots).
 * The reference binary-searches the previous frame's key-sorted contacts for the (shape0, shape1) range; the rows are already
 * grouped per (world, candidate pair) in ascending pair order, so the range is the previous frame's row block of the same pair,
 * found by a binary search in that world's previous candidate list.  Claims race through a packed 64-bit atomic min exactly as
 * in the reference (distance, then the low 32 key bits = shape1's low 9 bits << 23 | fingerprint's low 23 bits).
 * All arrays are device memory owned by the caller and zero-initialised before the first call. */
typedef struct {
    int32_t* prev_row_start;           /* [env_count + 1] */
    int32_t* prev_pair_count;          /* [env_count] (clamped to pairs_per_world); 0 = the world has no history (reset) */
    int32_t* prev_world_pairs;         /* [env_count * pairs_per_world][2] */
    int32_t* prev_pair_row;            /* [env_count * pairs_per_world] */
    int32_t* prev_pair_rows;           /* [env_count * pairs_per_world] rows of the pair (clamped to the row arrays) */
    uint8_t* prev_live;                /* [row_capacity] the row was a contact (shape0 != shape1) */
    float* prev_pos_world;             /* [row_capacity][3] world-space midpoint of the two contact points */
    float* prev_normal;                /* [row_capacity][3] */
    float* prev_body_frame;            /* [row_capacity][12] point0, point1, offset0, offset1 of the record used last frame, or NULL
                                          (sticky mode only) */
    unsigned long long* prev_claim;    /* [row_capacity] claim words, reset by nt_flat_rows_save_history */
} nt_flat_history;
/* match_index [row_capacity]: for every live row of this frame the ROW index of the matched previous row, -1 (MATCH_NOT_FOUND:
 * the pair had no contact last frame) or -2 (MATCH_BROKEN: nothing within the thresholds, or lost the race); inert rows -1.
 * io: the struct nt_sdf_rows_finalize was called with (pair tables + the row arrays).  body_q: env-major [7][nb][ES]. */
nt_status nt_flat_rows_match(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, const nt_flat_history* h,
                             float pos_threshold, float normal_dot_threshold, int32_t* match_index, void* stream);
/* sticky mode: matched rows that still touch (fresh gap <= 0) take last frame's body-frame points, offsets and normal */
nt_status nt_flat_rows_replay_matched(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, const nt_flat_history* h,
                                      const int32_t* match_index, void* stream);
/* persist this frame's rows (after the replay) as the next frame's history */
nt_status nt_flat_rows_save_history(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, const nt_flat_history* h,
                                    void* stream);

/* -------- introspection -------- */
/* ---- building the descriptor from Newton's own arrays ------------------------------------------------------------------
 * nt_model above is the device-side view (env-major SoA, env-uniform topology).  A binding that holds a finalized
 * newton.Model does not have to derive it: nt_model_create takes the flat arrays newton.Model already owns
 * (newton/_src/sim/model.py:808-1364, rigid subset; HOST pointers, e.g. wp.array.numpy()) and returns a handle whose
 * descriptor points at tables in HIP device memory (on_device = 1) or host memory (0: inspection / tests).
 * Requirements are those of the kernels: homogeneous worlds as ModelBuilder.replicate() produces them, bodies / joints /
 * env-local shapes world-major, static shapes in the global world (-1); anything else returns NT_ERR_UNSUPPORTED and
 * nt_model_last_error() says which array differs.  Counts are totals over all worlds. */
typedef struct {
    int32_t world_count;               /* Model.world_count (0: one implicit world) */
    int32_t body_count, joint_count, shape_count;
    int32_t joint_dof_count, joint_coord_count, joint_target_q_count;
    int32_t articulation_count, shape_contact_pair_count, mesh_point_count, gravity_count;
    /* bodies */
    const int32_t* body_world;         /* [body_count] */
    const int32_t* body_flags;
    const float* body_com;             /* [body_count][3] */
    const float* body_mass;
    const float* body_inv_mass;
    const float* body_inertia;         /* [body_count][9] */
    const float* body_inv_inertia;
    /* joints */
    const int32_t* joint_world;        /* [joint_count] */
    const int32_t* joint_type;
    const int32_t* joint_enabled;
    const int32_t* joint_parent;
    const int32_t* joint_child;
    const int32_t* joint_q_start;
    const int32_t* joint_qd_start;
    const int32_t* joint_target_q_start;
    const int32_t* joint_dof_dim;      /* [joint_count][2] linear, angular */
    const float* joint_X_p;            /* [joint_count][7] */
    const float* joint_X_c;
    const float* joint_axis;           /* [joint_dof_count][3] */
    const float* joint_limit_lower;    /* [joint_dof_count] ... */
    const float* joint_limit_upper;
    const float* joint_target_ke;
    const float* joint_target_kd;
    const float* joint_limit_ke;
    const float* joint_limit_kd;
    const float* joint_armature;
    const float* joint_damping;
    const int32_t* articulation_start; /* [articulation_count] first joint */
    const int32_t* articulation_end;   /* [articulation_count] one past the last joint */
    /* shapes */
    const int32_t* shape_world;        /* [shape_count] */
    const int32_t* shape_body;
    const int32_t* shape_type;
    const int32_t* shape_flags;
    const int32_t* shape_collision_group;
    const float* shape_transform;      /* [shape_count][7] */
    const float* shape_scale;          /* [shape_count][3] */
    const float* shape_margin;
    const float* shape_gap;
    const float* shape_material_mu;
    const float* shape_material_mu_torsional;
    const float* shape_material_mu_rolling;
    const float* shape_material_ke;
    const float* shape_material_kd;
    const float* shape_material_kf;
    const float* shape_material_ka;
    const float* shape_material_restitution;
    const int32_t* shape_contact_pairs; /* [shape_contact_pair_count][2] Model.shape_contact_pairs */
    const int32_t* shape_mesh_start;   /* [shape_count] or NULL: first hull vertex of a CONVEX_MESH shape in mesh_points, else -1 */
    const int32_t* shape_mesh_count;   /* [shape_count] or NULL */
    const float* mesh_points;          /* [mesh_point_count][3] unscaled hull vertices of the shared Mesh assets */
    const float* gravity;              /* [gravity_count][3]: one row per world, the last one for the global world (model.py:1300-1304) */
    /* pair routing out of the tiles (narrow_phase.py:531-538,618-640); NULL = no shape has a texture SDF / collision edges */
    const int32_t* shape_sdf_index;    /* [shape_count] Model._shape_sdf_index: index into the texture-SDF table or -1 */
    const int32_t* shape_edge_range;   /* [shape_count][2] Model.shape_edge_range: (first, count) of the shape's collision edges */
} nt_newton_model;
typedef struct nt_model_handle nt_model_handle;
nt_status nt_model_create(const nt_newton_model* src, int32_t on_device, nt_model_handle** out);
const nt_model* nt_model_get(const nt_model_handle* h);
/* position of every device (tile) pair in one world's slice of Model.shape_contact_pairs ([np] int64; analytic pairs come first) */
nt_status nt_model_pair_order(const nt_model_handle* h, int64_t* out);
/* The pairs of one world that leave the tiles, for the pipeline's SDF leg (nt_sdf_scene.template_pair / template_kind): *count of
 * them, and -- where the pointers are not NULL -- pairs [count][2] as template shape ids (env-local shape < ns, else ns + rank in
 * nt_model.gshape_id; ordered by ascending Newton (shape0, shape1), shape0 first), kind [count] (0 = both shapes carry a texture SDF
 * and collision edges, not box-box: mesh-SDF edge contacts; 1 = both shapes hydroelastic with SDFs: the SDF-SDF leg when the
 * pipeline enables it; 2 = (triangle mesh, infinite plane): vertex leg) and has_edges [count] (both shapes carry SDF + edges: what
 * a kind-1 pair falls back to without a hydroelastic configuration).  A MESH shape appears to the tiles as a CONVEX_MESH over its
 * vertex bounds (AABB only); MESH pairs that are none of the above return NT_ERR_UNSUPPORTED from nt_model_create. */
nt_status nt_model_sdf_pairs(const nt_model_handle* h, int32_t* count, int32_t* pairs, uint8_t* kind, uint8_t* has_edges);
/* Model.notify_model_changed(): re-pack the parameter tables (same topology) and refresh params_uniform */
nt_status nt_model_refresh_params(nt_model_handle* h, const nt_newton_model* src);
void nt_model_destroy(nt_model_handle* h);
const char* nt_model_last_error(void);           /* detail of the last NT_ERR_* of the nt_model_* calls on this thread */

/* -------- hipGraph capture of a frame (csrc/nt_graph.hip) --------
 * The reference's examples wrap simulate() -- clear_forces / collide / step per substep -- in wp.ScopedCapture and replay it with
 * wp.capture_launch (newton/examples/basic/example_basic_urdf.py:112-141).  Every entry point above launches on the caller's stream,
 * owns no memory and never reads back, so the same calls record into one hipGraph:
 *     nt_graph_capture_begin(stream);  <the frame's nt_* calls on `stream`>;  nt_graph_capture_end(stream, &g);
 *     nt_graph_launch(g, stream);  ...  nt_graph_destroy(g);
 * `stream` must be a created stream (not the legacy default stream).  Issue the frame once before capturing: the first launch of a
 * kernel whose tile needs more than 48 KB of LDS sets a function attribute, which is not a stream operation.  Replay repeats the
 * recorded launches on the recorded buffers: the frame must leave the caller's state pointers as it found them (an even number of
 * state swaps). */
typedef struct nt_graph nt_graph;
nt_status nt_graph_capture_begin(void* stream);
nt_status nt_graph_capture_end(void* stream, nt_graph** out);
nt_status nt_graph_launch(nt_graph* graph, void* stream);
void nt_graph_destroy(nt_graph* graph);

const char* nt_error_string(nt_status s);
const char* nt_build_info(void);                 /* "gfx950 ..." */
/* environments per workgroup the stepping kernels would use for this model (requested: 0 = auto), 0 if the working set
 * does not fit the CU's LDS in the model's current contact_scratch_in_hbm mode */
int32_t nt_pick_envs_per_block(const nt_model* m, int32_t requested);
int32_t nt_featherstone_lds_bytes_per_env(const nt_model* m);
int32_t nt_lds_bytes_per_env(const nt_model* m); /* LDS footprint of one env in the step kernels */
/* dst[i] = src[i], 4 B/lane coalesced: known-byte-count kernel used to calibrate the HBM PMC counters */
nt_status nt_calibration_copy(const float* src, float* dst, int64_t n, void* stream);
/* dst[i] = src[i] in 16 B per lane (src / dst 16-byte aligned, n a multiple of 4): the streaming copy bench.py times to print the
 * HBM bandwidth this box reaches (2 n 4 bytes per call) beside the 8 TB/s vendor figure the roofline fraction is quoted against
 * (BASELINE.md section 5).  Measurement aid, no reference counterpart. */
nt_status nt_bandwidth_probe(const float* src, float* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
