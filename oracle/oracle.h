/* TEST INFRASTRUCTURE ONLY -- CPU oracle (a restatement of Newton's Warp kernels).
 * Nothing under newton_amd/ may include, link or call this. See oracle/README.md.
 *
 * Data layout = Newton's own flat AoS arrays (newton/_src/sim/model.py:808-1364,
 * state.py:113-171, control.py:31-68, contacts.py:227-277), all host pointers.
 * Kernels are executed in ascending-tid order, so wp.atomic_add accumulation order is
 * the serial order a Warp-CPU launch would produce.
 *
 * Pinning: the three solvers reproduce, bit for bit in positions and linear velocities, vectors recorded from the REFERENCE's
 * own solver source executed on a pure-Python stand-in for Warp (tests/golden/make_xpbd_reference_vectors.py,
 * tests/test_reference_vectors.py).  Still restated (PARITY UNPINNED at bit level): the fp32 operation order inside Warp's
 * builtins (wp_builtins.h; warp-lang is not present in /root/reference nor installable here).  The collision pipeline is
 * pinned the same way (make_collide_reference_vectors.py: AABBs and contact arrays bit-identical to the reference kernels).
 */
#ifndef NEWTON_ORACLE_H
#define NEWTON_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int body_count, joint_count, shape_count, dof_count, coord_count, world_count, pair_count;
    /* bodies */
    const float* body_com;         /* [B][3] */
    const float* body_mass;        /* [B] */
    const float* body_inertia;     /* [B][9] row-major */
    const float* body_inv_mass;    /* [B]  (effective: zero for kinematic bodies) */
    const float* body_inv_inertia; /* [B][9] (effective) */
    const int32_t* body_flags;     /* [B] */
    const int32_t* body_world;     /* [B] */
    const float* gravity;          /* [W+1][3], last = global world -1 */
    /* joints */
    const int32_t* joint_type;     /* [J] */
    const int32_t* joint_enabled;  /* [J] 0/1 */
    const int32_t* joint_parent;   /* [J] */
    const int32_t* joint_child;    /* [J] */
    const float* joint_X_p;        /* [J][7] */
    const float* joint_X_c;        /* [J][7] */
    const int32_t* joint_q_start;  /* [J] */
    const int32_t* joint_qd_start; /* [J] */
    const int32_t* joint_target_q_start; /* [J] */
    const int32_t* joint_dof_dim;  /* [J][2] */
    const int32_t* joint_articulation; /* [J] */
    const float* joint_axis;       /* [D][3] */
    const float* joint_limit_lower;/* [D] */
    const float* joint_limit_upper;/* [D] */
    const float* joint_limit_ke;   /* [D] */
    const float* joint_limit_kd;   /* [D] */
    const float* joint_target_ke;  /* [D] */
    const float* joint_target_kd;  /* [D] */
    const float* joint_armature;   /* [D] */
    const float* joint_damping;    /* [D] */
    /* articulations */
    int articulation_count;
    const int32_t* articulation_start; /* [A] */
    const int32_t* articulation_end;   /* [A] */
    /* shapes */
    const float* shape_transform;  /* [S][7] */
    const int32_t* shape_body;     /* [S] */
    const int32_t* shape_type;     /* [S] */
    const float* shape_scale;      /* [S][3] */
    const float* shape_margin;     /* [S] */
    const float* shape_gap;        /* [S] */
    const int32_t* shape_flags;    /* [S] */
    const int32_t* shape_world;    /* [S] */
    const int32_t* shape_collision_group; /* [S] */
    const float* shape_collision_radius;  /* [S] */
    const float* shape_material_ke, *shape_material_kd, *shape_material_kf, *shape_material_ka;
    const float* shape_material_mu, *shape_material_mu_torsional, *shape_material_mu_rolling;
    const float* shape_material_restitution;
    const int32_t* shape_contact_pairs; /* [P][2] */
    const int32_t* joint_ancestor;      /* [J] joint whose child is this joint's parent body, or -1 (builder.py:12341-12348) */
    /* convex-hull shapes (GeoType.CONVEX_MESH): shared vertex table + per-shape slice, local AABB with the scale baked in */
    const float* mesh_points;           /* [V][3] */
    const int32_t* shape_mesh_start;    /* [S] first vertex or -1 */
    const int32_t* shape_mesh_count;    /* [S] */
    const float* shape_collision_aabb_lower; /* [S][3] (builder.py:11575-11612) */
    const float* shape_collision_aabb_upper; /* [S][3] */
    /* explicitly excluded shape pairs, canonical (min, max), lexicographically sorted (broad_phase_common.py:132-162) */
    int filter_pair_count;
    const int32_t* shape_collision_filter_pairs; /* [F][2] */
} o_model;

typedef struct {
    float* body_q;   /* [B][7] */
    float* body_qd;  /* [B][6] */
    float* body_f;   /* [B][6] */
    float* joint_q;  /* [coords] */
    float* joint_qd; /* [D] */
    float* body_parent_f; /* [B][6] nullable: extended state attribute (state.py:77,156-163), written by XPBD / Featherstone */
} o_state;

typedef struct {
    const float* joint_f;         /* [D] */
    const float* joint_target_q;  /* [coords or D] indexed via joint_target_q_start */
    const float* joint_target_qd; /* [D] */
} o_control;

typedef struct {
    int rigid_contact_max;
    int32_t* rigid_contact_count;   /* [1] */
    int32_t* shape0;                /* [Cmax] */
    int32_t* shape1;
    float* point0;                  /* [Cmax][3] */
    float* point1;
    float* offset0;
    float* offset1;
    float* normal;
    float* margin0;                 /* [Cmax] */
    float* margin1;
    int32_t* tids;
    /* optional per-contact overrides (contacts.py:227-277; kernels_contact.py:452-459): NULL = use the shape materials */
    const float* stiffness;         /* [Cmax] > 0 replaces ke */
    const float* damping;           /* [Cmax] > 0 replaces kd */
    const float* friction_scale;    /* [Cmax] > 0 scales mu */
} o_contacts;

typedef struct {
    int iterations;
    float joint_linear_relaxation, joint_angular_relaxation;
    float joint_linear_compliance, joint_angular_compliance;
    float rigid_contact_relaxation;
    int rigid_contact_con_weighting;
    float angular_damping;
    int enable_restitution;
    int compute_body_velocity_from_position_delta; /* SolverXPBD attribute, solver_xpbd.py:767-783 (update_body_velocities) */
} o_xpbd_params;

typedef struct {
    float angular_damping;
    float friction_smoothing;
    float joint_attach_ke, joint_attach_kd;
    int enable_tri_contact;
} o_semi_implicit_params;

typedef struct {
    float angular_damping; /* accepted for API parity; unused by the reference step */
    float friction_smoothing;
    /* SolverFeatherstone(update_mass_matrix_interval) (solver_featherstone.py:141,767): when mass_matrix_cache is non-NULL the
     * Cholesky factors L of all articulations (concatenated n x n blocks, articulation order) are stored there on steps that
     * rebuild J / M / H and reused on steps with update_mass_matrix == 0 */
    int update_mass_matrix;
    float* mass_matrix_cache;
} o_featherstone_params;

/* broad phase kinds */
enum { O_BP_EXPLICIT = 0, O_BP_NXN = 1, O_BP_SAP = 2 };

/* ---- kernels (each mirrors one reference launch) ---- */
void o_integrate_bodies(const o_model* m, const float* body_q, const float* body_qd, const float* body_f,
                        float angular_damping, float dt, float* body_q_new, float* body_qd_new);
void o_xpbd_step(const o_model* m, const o_xpbd_params* p, o_state* s_in, o_state* s_out,
                 const o_control* c, const o_contacts* contacts /*nullable*/, float dt);
void o_xpbd_step_report(const o_model* m, const o_xpbd_params* p, o_state* s_in, o_state* s_out, const o_control* c,
                        const o_contacts* contacts /*nullable*/, float dt, float* contact_force_out /*nullable [Cmax][6]*/);
void o_xpbd_rollout(const o_model* m, const o_xpbd_params* p, o_state* s0, o_state* s1, const o_control* c,
                    o_contacts* contacts, float dt, int substeps);
void o_semi_implicit_step(const o_model* m, const o_semi_implicit_params* p, o_state* s_in, o_state* s_out,
                          const o_control* c, const o_contacts* contacts /*nullable*/, float dt);
void o_featherstone_step(const o_model* m, const o_featherstone_params* p, o_state* s_in, o_state* s_out,
                         const o_control* c, const o_contacts* contacts /*nullable*/, float dt);
void o_featherstone_probe_H(float* out, int cap); /* test probe: H of articulation 0 from the next step */
/* collide: returns number of candidate pairs; candidate pairs written to out_pairs (cap pairs) if non-null */
int o_collide(const o_model* m, const float* body_q, int broad_phase, o_contacts* contacts,
              int32_t* out_pairs, int out_pairs_cap, float* out_aabb_lower, float* out_aabb_upper);
void o_eval_fk(const o_model* m, const float* joint_q, const float* joint_qd, float* body_q, float* body_qd);

/* single-function probes used by the known-answer tests (tests/test_oracle_known_answers.py) */
int o_probe_primitive(int type_a, int type_b, const float* xf_a, const float* xf_b, const float* scale_a,
                      const float* scale_b, float margin, float* out_dist4, float* out_pos12, float* out_normal3);

#ifdef __cplusplus
}
#endif
#endif
