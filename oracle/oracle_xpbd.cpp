// TEST INFRASTRUCTURE ONLY -- CPU oracle: SolverXPBD rigid path + shared integrator.
// Literal restatement (ascending-tid serial execution) of
//   integrate_rigid_body / integrate_bodies      newton/_src/solvers/solver.py:63-170
//   apply_body_deltas                            newton/_src/solvers/xpbd/kernels.py:864-933
//   apply_joint_forces                           newton/_src/solvers/xpbd/kernels.py:945-1075
//   update_joint_axis_limits / _weighted_target  newton/_src/solvers/xpbd/kernels.py:1078-1103
//   solve_body_joints                            newton/_src/solvers/xpbd/kernels.py:1513-2044
//   compute_contact_constraint_delta / compute_positional_correction / compute_angular_correction
//                                                newton/_src/solvers/xpbd/kernels.py:2047-2161
//   solve_body_contact_positions                 newton/_src/solvers/xpbd/kernels.py:2164-2399
//   copy_kinematic_body_state_kernel             newton/_src/solvers/xpbd/kernels.py:19-32
//   SolverXPBD.step control flow (ping-pong)     newton/_src/solvers/xpbd/solver_xpbd.py:329-862
#include <cstring>
#include <vector>

#include "oracle_common.h"

using namespace orc;

// ---------------------------------------------------------------- solver.py:63-107
static void integrate_rigid_body(const transform& q, const spatial& qd, const spatial& f, vec3 com, const mat33& inertia,
                                 float inv_mass, const mat33& inv_inertia, vec3 gravity, float angular_damping, float dt,
                                 transform& q_new, spatial& qd_new) {
    vec3 x0 = q.p;
    quat r0 = q.q;
    vec3 w0 = qd.bottom;
    vec3 v0 = qd.top;
    vec3 t0 = f.bottom;
    vec3 f0 = f.top;

    vec3 x_com = x0 + quat_rotate(r0, com);

    // linear part
    vec3 v1 = v0 + (f0 * inv_mass + gravity * nonzero(inv_mass)) * dt;
    vec3 x1 = x_com + v1 * dt;

    // angular part (body frame)
    vec3 wb = quat_rotate_inv(r0, w0);
    vec3 tb = quat_rotate_inv(r0, t0) - cross(wb, inertia * wb);

    vec3 w1 = quat_rotate(r0, wb + inv_inertia * tb * dt);
    quat r1 = normalize(r0 + quat(w1, 0.0f) * r0 * 0.5f * dt);

    w1 *= 1.0f - angular_damping * dt;

    q_new = transform(x1 - quat_rotate(r1, com), r1);
    qd_new = spatial(v1, w1);
}

extern "C" void o_integrate_bodies(const o_model* m, const float* body_q, const float* body_qd, const float* body_f,
                                   float angular_damping, float dt, float* body_q_new, float* body_qd_new) {
    for (int tid = 0; tid < m->body_count; ++tid) {
        if ((m->body_flags[tid] & BODY_KINEMATIC) != 0) {
            stx(body_q_new, tid, ldx(body_q, tid));
            sts(body_qd_new, tid, lds(body_qd, tid));
            continue;
        }
        int world_idx = m->body_world[tid];
        if (world_idx < 0) world_idx += m->world_count + 1;  // warp negative indexing: gravity[-1] = global
        vec3 g = ld3(m->gravity, world_idx);
        transform qn;
        spatial qdn;
        // NOTE: integrate_bodies receives model.body_inv_mass / body_inv_inertia (not the
        // kinematic-effective copies), solver.py:283-291.  The o_model passes raw arrays in
        // body_inv_mass_raw semantics: for dynamic bodies raw == effective.
        integrate_rigid_body(ldx(body_q, tid), lds(body_qd, tid), lds(body_f, tid), ld3(m->body_com, tid),
                             ldm(m->body_inertia, tid), m->body_inv_mass[tid], ldm(m->body_inv_inertia, tid), g,
                             angular_damping, dt, qn, qdn);
        stx(body_q_new, tid, qn);
        sts(body_qd_new, tid, qdn);
    }
}

// ---------------------------------------------------------------- xpbd/kernels.py:864-933
static void apply_body_deltas(const o_model* m, const float* q_in, const float* qd_in, const float* deltas,
                              const float* constraint_inv_weights /*nullable*/, float dt, float* q_out, float* qd_out) {
    for (int tid = 0; tid < m->body_count; ++tid) {
        float inv_m = m->body_inv_mass[tid];
        if (inv_m == 0.0f) {
            stx(q_out, tid, ldx(q_in, tid));
            sts(qd_out, tid, lds(qd_in, tid));
            continue;
        }
        mat33 inv_I = ldm(m->body_inv_inertia, tid);
        mat33 body_I = ldm(m->body_inertia, tid);

        transform tf = ldx(q_in, tid);
        spatial delta = lds(deltas, tid);
        spatial qd = lds(qd_in, tid);

        vec3 v0 = qd.top;
        vec3 w0 = qd.bottom;
        vec3 p0 = tf.p;
        quat q0 = tf.q;

        float weight = 1.0f;
        if (constraint_inv_weights) {
            float inv_weight = constraint_inv_weights[tid];
            if (inv_weight > 0.0f) weight = 1.0f / inv_weight;
        }

        vec3 dp = delta.top * (inv_m * weight);
        vec3 dq = delta.bottom * weight;

        vec3 wb = quat_rotate_inv(q0, w0);
        vec3 dwb = inv_I * quat_rotate_inv(q0, dq);
        // coriolis forces delta from dwb = (wb + dwb) I (wb + dwb) - wb I wb
        vec3 tb = cross(dwb, body_I * (wb + dwb)) + cross(wb, body_I * dwb);
        vec3 dw1 = quat_rotate(q0, dwb - (dt * inv_I) * tb);

        // update orientation
        quat q1 = q0 + 0.5f * quat(dw1 * dt, 0.0f) * q0;
        q1 = normalize(q1);

        // update position
        vec3 com = ld3(m->body_com, tid);
        vec3 x_com = p0 + quat_rotate(q0, com);
        vec3 p1 = x_com + dp * dt;
        p1 -= quat_rotate(q1, com);

        stx(q_out, tid, transform(p1, q1));

        vec3 v1 = v0 + dp;
        vec3 w1 = w0 + dw1;

        if (length(v1) < 1e-4f) v1 = vec3(0.0f);
        if (length(w1) < 1e-4f) w1 = vec3(0.0f);

        sts(qd_out, tid, spatial(v1, w1));
    }
}

// ---------------------------------------------------------------- xpbd/kernels.py:945-1075
static void apply_joint_forces(const o_model* m, const float* body_q, const float* joint_f, float dt, float* body_f,
                               float* joint_impulse /*nullable, [J][6]*/) {
    for (int tid = 0; tid < m->joint_count; ++tid) {
        int type = m->joint_type[tid];
        if (!m->joint_enabled[tid]) continue;
        if (type == FIXED || type == ROD) continue;

        int id_c = m->joint_child[tid];
        int id_p = m->joint_parent[tid];

        transform X_pj = ldx(m->joint_X_p, tid);
        transform X_cj = ldx(m->joint_X_c, tid);

        transform X_wp = X_pj;
        transform pose_p = X_pj;
        vec3 com_p(0.0f);
        if (id_p >= 0) {
            pose_p = ldx(body_q, id_p);
            X_wp = pose_p * X_wp;
            com_p = ld3(m->body_com, id_p);
        }
        vec3 r_p = X_wp.p - transform_point(pose_p, com_p);

        transform pose_c = ldx(body_q, id_c);
        transform X_wc = pose_c * X_cj;
        vec3 com_c = ld3(m->body_com, id_c);
        vec3 r_c = X_wc.p - transform_point(pose_c, com_c);

        int qd_start = m->joint_qd_start[tid];
        int lin_axis_count = m->joint_dof_dim[2 * tid + 0];
        int ang_axis_count = m->joint_dof_dim[2 * tid + 1];

        vec3 t_total, f_total;

        if (type == FREE || type == DISTANCE) {
            f_total = vec3(joint_f[qd_start + 0], joint_f[qd_start + 1], joint_f[qd_start + 2]);
            t_total = vec3(joint_f[qd_start + 3], joint_f[qd_start + 4], joint_f[qd_start + 5]);
            adds(body_f, id_c, spatial(f_total, t_total));
            if (id_p >= 0) subs(body_f, id_p, spatial(f_total, t_total));
            if (joint_impulse) adds(joint_impulse, tid, spatial(f_total, t_total) * dt);  // kernels.py:1018-1019
            continue;
        } else if (type == BALL) {
            t_total = vec3(joint_f[qd_start + 0], joint_f[qd_start + 1], joint_f[qd_start + 2]);
        } else if (type == REVOLUTE || type == PRISMATIC || type == D6) {
            for (int k = 0; k < 3; ++k) {
                if (lin_axis_count > k) {
                    vec3 axis = ld3(m->joint_axis, qd_start + k);
                    float f = joint_f[qd_start + k];
                    vec3 a_p = transform_vector(X_wp, axis);
                    f_total += f * a_p;
                }
            }
            for (int k = 0; k < 3; ++k) {
                if (ang_axis_count > k) {
                    vec3 axis = ld3(m->joint_axis, qd_start + lin_axis_count + k);
                    float f = joint_f[qd_start + lin_axis_count + k];
                    vec3 a_p = transform_vector(X_wp, axis);
                    t_total += f * a_p;
                }
            }
        }

        spatial child_wrench_at_com(f_total, t_total + cross(r_c, f_total));
        if (id_p >= 0) subs(body_f, id_p, spatial(f_total, t_total + cross(r_p, f_total)));
        adds(body_f, id_c, child_wrench_at_com);
        if (joint_impulse) adds(joint_impulse, tid, child_wrench_at_com * dt);  // kernels.py:1074-1075
    }
}

// ---------------------------------------------------------------- xpbd/kernels.py:1078-1103
struct limits6 {
    vec3 lower, upper;
};
static limits6 update_joint_axis_limits(vec3 axis, float limit_lower, float limit_upper, limits6 in) {
    vec3 lo_temp = axis * limit_lower;
    vec3 up_temp = axis * limit_upper;
    vec3 lo = vmin(lo_temp, up_temp);
    vec3 up = vmax(lo_temp, up_temp);
    limits6 out;
    out.lower = vmin(in.lower, lo);
    out.upper = vmax(in.upper, up);
    return out;
}
struct tw6 {
    vec3 targets, weights;
};
static tw6 update_joint_axis_weighted_target(vec3 axis, float target, float weight, tw6 in) {
    vec3 weighted_axis = axis * weight;
    tw6 out;
    out.targets = in.targets + weighted_axis * target;
    out.weights = in.weights + vabs(weighted_axis);
    return out;
}

// ---------------------------------------------------------------- xpbd/kernels.py:2047-2161
static float compute_contact_constraint_delta(float err, const transform& tf_a, const transform& tf_b, float m_inv_a,
                                              float m_inv_b, const mat33& I_inv_a, const mat33& I_inv_b, vec3 linear_a,
                                              vec3 linear_b, vec3 angular_a, vec3 angular_b, float relaxation, float dt) {
    float denom = 0.0f;
    denom += length_sq(linear_a) * m_inv_a;
    denom += length_sq(linear_b) * m_inv_b;
    vec3 rot_angular_a = quat_rotate_inv(tf_a.q, angular_a);
    vec3 rot_angular_b = quat_rotate_inv(tf_b.q, angular_b);
    denom += dot(rot_angular_a, I_inv_a * rot_angular_a);
    denom += dot(rot_angular_b, I_inv_b * rot_angular_b);
    float delta_lambda = -err;
    if (denom > 0.0f) delta_lambda /= dt * denom;
    return delta_lambda * relaxation;
}

static float compute_positional_correction(float err, float derr, const transform& tf_a, const transform& tf_b,
                                           float m_inv_a, float m_inv_b, const mat33& I_inv_a, const mat33& I_inv_b,
                                           vec3 linear_a, vec3 linear_b, vec3 angular_a, vec3 angular_b, float lambda_in,
                                           float compliance, float damping, float dt) {
    float denom = 0.0f;
    denom += length_sq(linear_a) * m_inv_a;
    denom += length_sq(linear_b) * m_inv_b;
    vec3 rot_angular_a = quat_rotate_inv(tf_a.q, angular_a);
    vec3 rot_angular_b = quat_rotate_inv(tf_b.q, angular_b);
    denom += dot(rot_angular_a, I_inv_a * rot_angular_a);
    denom += dot(rot_angular_b, I_inv_b * rot_angular_b);
    float alpha = compliance;
    float gamma = compliance * damping;
    float delta_lambda = -(err + alpha * lambda_in + gamma * derr);
    if (denom + alpha > 0.0f) delta_lambda /= (dt + gamma) * denom + alpha / dt;
    return delta_lambda;
}

static float compute_angular_correction(float err, float derr, const transform& tf_a, const transform& tf_b,
                                        const mat33& I_inv_a, const mat33& I_inv_b, vec3 angular_a, vec3 angular_b,
                                        float lambda_in, float compliance, float damping, float dt) {
    float denom = 0.0f;
    vec3 rot_angular_a = quat_rotate_inv(tf_a.q, angular_a);
    vec3 rot_angular_b = quat_rotate_inv(tf_b.q, angular_b);
    denom += dot(rot_angular_a, I_inv_a * rot_angular_a);
    denom += dot(rot_angular_b, I_inv_b * rot_angular_b);
    float alpha = compliance;
    float gamma = compliance * damping;
    float delta_lambda = -(err + alpha * lambda_in + gamma * derr);
    if (denom + alpha > 0.0f) delta_lambda /= (dt + gamma) * denom + alpha / dt;
    return delta_lambda;
}

// gather per-axis limits/targets for up to 3 axes starting at (axis_idx0, target_idx0)
static void gather_axes(const o_model* m, const o_control* c, int count, int axis_idx0, int target_idx0, limits6& lim,
                        vec3& axis_target_pos, vec3& axis_stiffness, vec3& axis_target_vel, vec3& axis_damping) {
    lim.lower = vec3(0.0f);
    lim.upper = vec3(0.0f);
    tw6 pos_ke, vel_kd;
    for (int k = 0; k < 3; ++k) {
        if (count > k) {
            int axis_idx = axis_idx0 + k;
            int target_axis_idx = target_idx0 + k;
            vec3 axis = ld3(m->joint_axis, axis_idx);
            float lower = m->joint_limit_lower[axis_idx];
            float upper = m->joint_limit_upper[axis_idx];
            if (k == 0) {
                vec3 lo_temp = axis * lower;
                vec3 up_temp = axis * upper;
                lim.lower = vmin(lo_temp, up_temp);
                lim.upper = vmax(lo_temp, up_temp);
            } else {
                lim = update_joint_axis_limits(axis, lower, upper, lim);
            }
            float ke = m->joint_target_ke[axis_idx];
            float kd = m->joint_target_kd[axis_idx];
            float target_pos = c->joint_target_q[target_axis_idx];
            float target_vel = c->joint_target_qd[axis_idx];
            if (ke > 0.0f) pos_ke = update_joint_axis_weighted_target(axis, target_pos, ke, pos_ke);
            if (kd > 0.0f) vel_kd = update_joint_axis_weighted_target(axis, target_vel, kd, vel_kd);
        }
    }
    axis_target_pos = pos_ke.targets;
    axis_stiffness = pos_ke.weights;
    axis_target_vel = vel_kd.targets;
    axis_damping = vel_kd.weights;
    for (int i = 0; i < 3; ++i)
        if (axis_stiffness[i] > 0.0f) axis_target_pos[i] /= axis_stiffness[i];
    for (int i = 0; i < 3; ++i)
        if (axis_damping[i] > 0.0f) axis_target_vel[i] /= axis_damping[i];
}

// ---------------------------------------------------------------- xpbd/kernels.py:1513-2044
static void solve_body_joints(const o_model* m, const o_xpbd_params* prm, const o_control* c, const float* body_q,
                              const float* body_qd, float dt, float* deltas, float* joint_impulse /*nullable, [J][6]*/) {
    const float joint_linear_compliance = prm->joint_linear_compliance;
    const float joint_angular_compliance = prm->joint_angular_compliance;
    const float angular_relaxation = prm->joint_angular_relaxation;
    const float linear_relaxation = prm->joint_linear_relaxation;

    for (int tid = 0; tid < m->joint_count; ++tid) {
        int type = m->joint_type[tid];
        if (!m->joint_enabled[tid]) continue;
        if (type == FREE) continue;

        int id_c = m->joint_child[tid];
        int id_p = m->joint_parent[tid];

        transform X_pj = ldx(m->joint_X_p, tid);
        transform X_cj = ldx(m->joint_X_c, tid);

        transform X_wp = X_pj;
        float m_inv_p = 0.0f;
        mat33 I_inv_p;
        transform pose_p = X_pj;
        vec3 com_p(0.0f), vel_p(0.0f), omega_p(0.0f);
        if (id_p >= 0) {
            pose_p = ldx(body_q, id_p);
            X_wp = pose_p * X_wp;
            com_p = ld3(m->body_com, id_p);
            m_inv_p = m->body_inv_mass[id_p];
            I_inv_p = ldm(m->body_inv_inertia, id_p);
            spatial qd = lds(body_qd, id_p);
            vel_p = qd.top;
            omega_p = qd.bottom;
        }

        transform pose_c = ldx(body_q, id_c);
        transform X_wc = pose_c * X_cj;
        vec3 com_c = ld3(m->body_com, id_c);
        float m_inv_c = m->body_inv_mass[id_c];
        mat33 I_inv_c = ldm(m->body_inv_inertia, id_c);
        spatial qdc = lds(body_qd, id_c);
        vec3 vel_c = qdc.top;
        vec3 omega_c = qdc.bottom;

        if (m_inv_p == 0.0f && m_inv_c == 0.0f) continue;

        vec3 lin_delta_p(0.0f), ang_delta_p(0.0f), lin_delta_c(0.0f), ang_delta_c(0.0f);

        transform rel_pose = transform_inverse(X_wp) * X_wc;
        vec3 rel_p = rel_pose.p;

        vec3 x_p = X_wp.p;
        vec3 x_c = X_wc.p;

        float linear_compliance = joint_linear_compliance;
        float angular_compliance = joint_angular_compliance;

        int axis_start = m->joint_qd_start[tid];
        int target_axis_start = m->joint_target_q_start[tid];
        int lin_axis_count = m->joint_dof_dim[2 * tid + 0];
        int ang_axis_count = m->joint_dof_dim[2 * tid + 1];

        vec3 world_com_p = transform_point(pose_p, com_p);
        vec3 world_com_c = transform_point(pose_c, com_c);

        if (type == DISTANCE) {
            vec3 r_p = x_p - world_com_p;
            vec3 r_c = x_c - world_com_c;
            float lower = m->joint_limit_lower[axis_start];
            float upper = m->joint_limit_upper[axis_start];
            if (lower < 0.0f && upper < 0.0f) continue;  // no limits
            vec3 anchor_delta = x_c - x_p;
            float d = length(anchor_delta);
            float err = 0.0f;
            if (lower >= 0.0f && d < lower)
                err = d - lower;
            else if (upper >= 0.0f && d > upper)
                err = d - upper;

            if (std::fabs(err) > 1e-9f) {
                vec3 linear_c;
                if (d > 1e-9f) {
                    linear_c = anchor_delta / d;
                } else {
                    vec3 com_delta = world_com_c - world_com_p;
                    if (length_sq(com_delta) > 1e-18f)
                        linear_c = normalize(com_delta);
                    else
                        linear_c = transform_vector(X_wp, vec3(1.0f, 0.0f, 0.0f));
                }
                vec3 linear_p = -linear_c;
                vec3 angular_p = -cross(r_p, linear_c);
                vec3 angular_c = cross(r_c, linear_c);
                float derr = dot(linear_p, vel_p) + dot(linear_c, vel_c) + dot(angular_p, omega_p) + dot(angular_c, omega_c);
                float lambda_in = 0.0f;
                float compliance = linear_compliance;
                float ke = m->joint_target_ke[axis_start];
                if (ke > 0.0f) compliance = 1.0f / ke;
                float damping = m->joint_target_kd[axis_start];
                float d_lambda = compute_positional_correction(err, derr, pose_p, pose_c, m_inv_p, m_inv_c, I_inv_p, I_inv_c,
                                                               linear_p, linear_c, angular_p, angular_c, lambda_in,
                                                               compliance, damping, dt);
                lin_delta_p += linear_p * (d_lambda * linear_relaxation);
                ang_delta_p += angular_p * (d_lambda * angular_relaxation);
                lin_delta_c += linear_c * (d_lambda * linear_relaxation);
                ang_delta_c += angular_c * (d_lambda * angular_relaxation);
            }
        } else {
            limits6 lim;
            vec3 axis_target_pos, axis_stiffness, axis_target_vel, axis_damping;
            gather_axes(m, c, lin_axis_count, axis_start, target_axis_start, lim, axis_target_pos, axis_stiffness,
                        axis_target_vel, axis_damping);
            vec3 axis_limits_lower = lim.lower;
            vec3 axis_limits_upper = lim.upper;

            vec3 projected_rel_p = rel_p;
            for (int dim = 0; dim < 3; ++dim) {
                float lower = axis_limits_lower[dim];
                float upper = axis_limits_upper[dim];
                if (rel_p[dim] < lower)
                    projected_rel_p[dim] = lower;
                else if (rel_p[dim] > upper)
                    projected_rel_p[dim] = upper;
                else if (axis_stiffness[dim] > 0.0f)
                    projected_rel_p[dim] = clampf(axis_target_pos[dim], lower, upper);
            }

            mat33 frame_p = quat_to_matrix(X_wp.q);
            vec3 r_p = transform_point(X_wp, projected_rel_p) - world_com_p;
            vec3 r_c = x_c - world_com_c;

            for (int dim = 0; dim < 3; ++dim) {
                float e = rel_p[dim];
                vec3 linear_c(frame_p(0, dim), frame_p(1, dim), frame_p(2, dim));
                vec3 linear_p = -linear_c;
                vec3 angular_p = -cross(r_p, linear_c);
                vec3 angular_c = cross(r_c, linear_c);
                float derr = dot(linear_p, vel_p) + dot(linear_c, vel_c) + dot(angular_p, omega_p) + dot(angular_c, omega_c);

                float err = 0.0f;
                float compliance = linear_compliance;
                float damping = 0.0f;

                float target_vel = axis_target_vel[dim];
                float derr_rel = derr - target_vel;

                float lower = axis_limits_lower[dim];
                float upper = axis_limits_upper[dim];
                if (e < lower) {
                    err = e - lower;
                } else if (e > upper) {
                    err = e - upper;
                } else {
                    float target_pos = axis_target_pos[dim];
                    target_pos = clampf(target_pos, lower, upper);
                    if (axis_stiffness[dim] > 0.0f) {
                        err = e - target_pos;
                        compliance = 1.0f / axis_stiffness[dim];
                        damping = axis_damping[dim];
                    } else if (axis_damping[dim] > 0.0f) {
                        compliance = 1.0f / axis_damping[dim];
                        damping = axis_damping[dim];
                    }
                }

                if (std::fabs(err) > 1e-9f || std::fabs(derr_rel) > 1e-9f) {
                    float lambda_in = 0.0f;
                    float d_lambda = compute_positional_correction(err, derr_rel, pose_p, pose_c, m_inv_p, m_inv_c, I_inv_p,
                                                                   I_inv_c, linear_p, linear_c, angular_p, angular_c,
                                                                   lambda_in, compliance, damping, dt);
                    lin_delta_p += linear_p * (d_lambda * linear_relaxation);
                    ang_delta_p += angular_p * (d_lambda * angular_relaxation);
                    lin_delta_c += linear_c * (d_lambda * linear_relaxation);
                    ang_delta_c += angular_c * (d_lambda * angular_relaxation);
                }
            }
        }

        if (type == FIXED || type == PRISMATIC || type == REVOLUTE || type == D6) {
            quat q_p = X_wp.q;
            quat q_c = X_wc.q;

            // make quats lie in same hemisphere
            if (dot(q_p, q_c) < 0.0f) q_c = q_c * -1.0f;

            quat rel_q = quat_inverse(q_p) * q_c;

            quat qtwist = normalize(quat(rel_q.x, 0.0f, 0.0f, rel_q.w));
            quat qswing = rel_q * quat_inverse(qtwist);

            float s = std::sqrt(rel_q.x * rel_q.x + rel_q.w * rel_q.w);
            float invs = 1.0f / s;
            float invscube = invs * invs * invs;

            float err_0 = 2.0f * std::asin(clampf(qtwist.x, -1.0f, 1.0f));
            float err_1 = qswing.y;
            float err_2 = qswing.z;
            quat grad_0(invs - rel_q.x * rel_q.x * invscube, 0.0f, 0.0f, -(rel_q.w * rel_q.x) * invscube);
            quat grad_1(-rel_q.w * (rel_q.w * rel_q.z + rel_q.x * rel_q.y) * invscube, rel_q.w * invs, -rel_q.x * invs,
                        rel_q.x * (rel_q.w * rel_q.z + rel_q.x * rel_q.y) * invscube);
            quat grad_2(rel_q.w * (rel_q.w * rel_q.y - rel_q.x * rel_q.z) * invscube, rel_q.x * invs, rel_q.w * invs,
                        rel_q.x * (rel_q.z * rel_q.x - rel_q.w * rel_q.y) * invscube);
            grad_0 = grad_0 * (2.0f / std::fabs(qtwist.w));

            float swing_sq = qswing.w * qswing.w;
            const float angularEps = 1.0e-4f;
            if (swing_sq + angularEps < 1.0f) {
                float d = std::sqrt(1.0f - qswing.w * qswing.w);
                float theta = 2.0f * std::acos(clampf(qswing.w, -1.0f, 1.0f));
                float scale = theta / d;
                err_1 *= scale;
                err_2 *= scale;
                grad_1 = grad_1 * scale;
                grad_2 = grad_2 * scale;
            }

            vec3 errs(err_0, err_1, err_2);
            vec3 grad_x(grad_0.x, grad_1.x, grad_2.x);
            vec3 grad_y(grad_0.y, grad_1.y, grad_2.y);
            vec3 grad_z(grad_0.z, grad_1.z, grad_2.z);
            vec3 grad_w(grad_0.w, grad_1.w, grad_2.w);

            limits6 lim;
            vec3 axis_target_pos, axis_stiffness, axis_target_vel, axis_damping;
            gather_axes(m, c, ang_axis_count, axis_start + lin_axis_count, target_axis_start + lin_axis_count, lim,
                        axis_target_pos, axis_stiffness, axis_target_vel, axis_damping);
            vec3 axis_limits_lower = lim.lower;
            vec3 axis_limits_upper = lim.upper;

            for (int dim = 0; dim < 3; ++dim) {
                float e = errs[dim];
                quat grad(grad_x[dim], grad_y[dim], grad_z[dim], grad_w[dim]);
                quat quat_c = 0.5f * q_p * grad * quat_inverse(q_c);
                vec3 angular_c(quat_c.x, quat_c.y, quat_c.z);
                vec3 angular_p = -angular_c;
                float derr = dot(angular_p, omega_p) + dot(angular_c, omega_c);

                float err = 0.0f;
                float compliance = angular_compliance;
                float damping = 0.0f;

                float target_vel = axis_target_vel[dim];
                float angular_c_len = length(angular_c);
                float derr_rel = derr - target_vel * angular_c_len;

                float lower = axis_limits_lower[dim];
                float upper = axis_limits_upper[dim];
                if (e < lower) {
                    err = e - lower;
                } else if (e > upper) {
                    err = e - upper;
                } else {
                    float target_pos = axis_target_pos[dim];
                    target_pos = clampf(target_pos, lower, upper);
                    if (axis_stiffness[dim] > 0.0f) {
                        err = e - target_pos;
                        compliance = 1.0f / axis_stiffness[dim];
                        damping = axis_damping[dim];
                    } else if (axis_damping[dim] > 0.0f) {
                        damping = axis_damping[dim];
                        compliance = 1.0f / axis_damping[dim];
                    }
                }

                float d_lambda = compute_angular_correction(err, derr_rel, pose_p, pose_c, I_inv_p, I_inv_c, angular_p,
                                                            angular_c, 0.0f, compliance, damping, dt) *
                                 angular_relaxation;

                ang_delta_p += angular_p * d_lambda;
                ang_delta_c += angular_c * d_lambda;
            }
        }

        if (id_p >= 0) adds(deltas, id_p, spatial(lin_delta_p, ang_delta_p));
        if (id_c >= 0) adds(deltas, id_c, spatial(lin_delta_c, ang_delta_c));
        if (joint_impulse) adds(joint_impulse, tid, spatial(lin_delta_c, ang_delta_c));  // kernels.py:2043-2044
    }
}

// ---------------------------------------------------------------- xpbd/kernels.py:2164-2399
static void solve_body_contact_positions(const o_model* m, const o_contacts* ct, const float* body_q, const float* body_qd,
                                         float relaxation, float dt, float* deltas, float* contact_inv_weight /*nullable*/,
                                         float* contact_impulse /*nullable, [Cmax][6]*/) {
    int count = ct->rigid_contact_count[0];
    for (int tid = 0; tid < ct->rigid_contact_max; ++tid) {
        if (tid >= count) break;

        int shape_a = ct->shape0[tid];
        int shape_b = ct->shape1[tid];
        if (shape_a == shape_b) continue;
        int body_a = -1;
        if (shape_a >= 0) body_a = m->shape_body[shape_a];
        int body_b = -1;
        if (shape_b >= 0) body_b = m->shape_body[shape_b];
        if (body_a == body_b) continue;

        transform X_wb_a, X_wb_b;
        if (body_a >= 0) X_wb_a = ldx(body_q, body_a);
        if (body_b >= 0) X_wb_b = ldx(body_q, body_b);

        vec3 bx_a = transform_point(X_wb_a, ld3(ct->point0, tid));
        vec3 bx_b = transform_point(X_wb_b, ld3(ct->point1, tid));

        vec3 n = ld3(ct->normal, tid);
        // contact_surface_separation  newton/_src/sim/contacts.py:70-92
        float d = dot(n, bx_b - bx_a) - (ct->margin0[tid] + ct->margin1[tid]);

        if (d >= 0.0f) continue;

        float m_inv_a = 0.0f, m_inv_b = 0.0f;
        mat33 I_inv_a, I_inv_b;
        vec3 com_a(0.0f), com_b(0.0f);
        vec3 omega_a(0.0f), omega_b(0.0f);
        vec3 offset_a = ld3(ct->offset0, tid);
        vec3 offset_b = ld3(ct->offset1, tid);

        if (body_a >= 0) {
            com_a = ld3(m->body_com, body_a);
            m_inv_a = m->body_inv_mass[body_a];
            I_inv_a = ldm(m->body_inv_inertia, body_a);
            omega_a = lds(body_qd, body_a).bottom;
        }
        if (body_b >= 0) {
            com_b = ld3(m->body_com, body_b);
            m_inv_b = m->body_inv_mass[body_b];
            I_inv_b = ldm(m->body_inv_inertia, body_b);
            omega_b = lds(body_qd, body_b).bottom;
        }

        int mat_nonzero = 0;
        float mu = 0.0f, mu_torsional = 0.0f, mu_rolling = 0.0f;
        if (shape_a >= 0) {
            mat_nonzero += 1;
            mu += m->shape_material_mu[shape_a];
            mu_torsional += m->shape_material_mu_torsional[shape_a];
            mu_rolling += m->shape_material_mu_rolling[shape_a];
        }
        if (shape_b >= 0) {
            mat_nonzero += 1;
            mu += m->shape_material_mu[shape_b];
            mu_torsional += m->shape_material_mu_torsional[shape_b];
            mu_rolling += m->shape_material_mu_rolling[shape_b];
        }
        if (mat_nonzero > 0) {
            mu /= float(mat_nonzero);
            mu_torsional /= float(mat_nonzero);
            mu_rolling /= float(mat_nonzero);
        }

        vec3 r_a = bx_a - transform_point(X_wb_a, com_a);
        vec3 r_b = bx_b - transform_point(X_wb_b, com_b);

        vec3 angular_a = -cross(r_a, n);
        vec3 angular_b = cross(r_b, n);

        if (contact_inv_weight) {
            if (body_a >= 0) contact_inv_weight[body_a] += 1.0f;
            if (body_b >= 0) contact_inv_weight[body_b] += 1.0f;
        }

        float lambda_n = compute_contact_constraint_delta(d, X_wb_a, X_wb_b, m_inv_a, m_inv_b, I_inv_a, I_inv_b, -n, n,
                                                          angular_a, angular_b, relaxation, dt);

        vec3 lin_delta_a = -n * lambda_n;
        vec3 lin_delta_b = n * lambda_n;
        vec3 ang_delta_a = angular_a * lambda_n;
        vec3 ang_delta_b = angular_b * lambda_n;

        // linear friction
        if (mu > 0.0f) {
            // contact_surface_point  newton/_src/sim/contacts.py:95-115
            bx_a = transform_point(X_wb_a, ld3(ct->point0, tid) + offset_a);
            bx_b = transform_point(X_wb_b, ld3(ct->point1, tid) + offset_b);

            vec3 delta = bx_b - bx_a;
            vec3 friction_delta = delta - dot(n, delta) * n;

            r_a = bx_a - transform_point(X_wb_a, com_a);
            r_b = bx_b - transform_point(X_wb_b, com_b);

            vec3 rel_v_kin_t(0.0f);
            if (body_a >= 0 && (m->body_flags[body_a] & BODY_KINEMATIC) != 0) {
                vec3 v_a = velocity_at_point(lds(body_qd, body_a), r_a);
                rel_v_kin_t = rel_v_kin_t - (v_a - dot(n, v_a) * n);
            }
            if (body_b >= 0 && (m->body_flags[body_b] & BODY_KINEMATIC) != 0) {
                vec3 v_b = velocity_at_point(lds(body_qd, body_b), r_b);
                rel_v_kin_t = rel_v_kin_t + (v_b - dot(n, v_b) * n);
            }
            friction_delta += rel_v_kin_t * dt;

            vec3 perp = normalize(friction_delta);

            angular_a = -cross(r_a, perp);
            angular_b = cross(r_b, perp);

            float err = length(friction_delta);

            if (err > 0.0f) {
                float lambda_fr = compute_contact_constraint_delta(err, X_wb_a, X_wb_b, m_inv_a, m_inv_b, I_inv_a, I_inv_b,
                                                                   -perp, perp, angular_a, angular_b, relaxation, dt);
                lambda_fr = wmax(lambda_fr, -lambda_n * mu);

                lin_delta_a -= perp * lambda_fr;
                lin_delta_b += perp * lambda_fr;
                ang_delta_a += angular_a * lambda_fr;
                ang_delta_b += angular_b * lambda_fr;
            }
        }

        vec3 delta_omega = omega_b - omega_a;

        if (mu_torsional > 0.0f) {
            float err = dot(delta_omega, n) * dt;
            if (std::fabs(err) > 0.0f) {
                vec3 lin(0.0f);
                float lambda_torsion = compute_contact_constraint_delta(err, X_wb_a, X_wb_b, m_inv_a, m_inv_b, I_inv_a,
                                                                        I_inv_b, lin, lin, -n, n, relaxation, dt);
                lambda_torsion = clampf(lambda_torsion, -lambda_n * mu_torsional, lambda_n * mu_torsional);
                ang_delta_a -= n * lambda_torsion;
                ang_delta_b += n * lambda_torsion;
            }
        }

        if (mu_rolling > 0.0f) {
            delta_omega -= dot(n, delta_omega) * n;
            float err = length(delta_omega) * dt;
            if (err > 0.0f) {
                vec3 lin(0.0f);
                vec3 roll_n = normalize(delta_omega);
                float lambda_roll = compute_contact_constraint_delta(err, X_wb_a, X_wb_b, m_inv_a, m_inv_b, I_inv_a, I_inv_b,
                                                                     lin, lin, -roll_n, roll_n, relaxation, dt);
                lambda_roll = wmax(lambda_roll, -lambda_n * mu_rolling);
                ang_delta_a -= roll_n * lambda_roll;
                ang_delta_b += roll_n * lambda_roll;
            }
        }

        if (body_a >= 0) adds(deltas, body_a, spatial(lin_delta_a, ang_delta_a));
        if (body_b >= 0) adds(deltas, body_b, spatial(lin_delta_b, ang_delta_b));
        if (contact_impulse) adds(contact_impulse, tid, spatial(lin_delta_a, ang_delta_a));  // kernels.py:2398-2399
    }
}

// ---------------------------------------------------------------- xpbd/kernels.py:2402-2461
static void accumulate_weighted_contact_impulse(const o_model* m, const o_contacts* ct, const float* contact_impulse_iter,
                                                const float* constraint_inv_weight /*nullable*/, float* contact_impulse) {
    int count = ct->rigid_contact_count[0];
    for (int tid = 0; tid < ct->rigid_contact_max; ++tid) {
        if (tid >= count) break;
        spatial impulse = lds(contact_impulse_iter, tid);
        float weight = 1.0f;
        if (constraint_inv_weight) {
            float n_a = 0.0f, n_b = 0.0f;
            int shape_a = ct->shape0[tid];
            if (shape_a >= 0) {
                int body_a = m->shape_body[shape_a];
                if (body_a >= 0) n_a = constraint_inv_weight[body_a];
            }
            int shape_b = ct->shape1[tid];
            if (shape_b >= 0) {
                int body_b = m->shape_body[shape_b];
                if (body_b >= 0) n_b = constraint_inv_weight[body_b];
            }
            float n_sum = n_a + n_b;
            if (n_sum > 0.0f) {
                if (n_a == 0.0f) weight = 1.0f / n_b;
                else if (n_b == 0.0f) weight = 1.0f / n_a;
                else weight = 2.0f / n_sum;
            }
        }
        adds(contact_impulse, tid, spatial(impulse.top * weight, impulse.bottom * weight));
    }
}

// ---------------------------------------------------------------- solver_xpbd.py:329-862 (rigid-only model)
// apply_rigid_restitution (xpbd/kernels.py:2583-2728): velocity-level restitution impulses from the pre-step state
static void apply_rigid_restitution(const o_model* m, const o_contacts* ct, const float* body_q, const float* body_qd,
                                    const float* body_q_prev, const float* body_qd_prev, float dt, float* deltas) {
    (void)body_q;
    int count = ct->rigid_contact_count[0];
    for (int tid = 0; tid < ct->rigid_contact_max; ++tid) {
        if (tid >= count) break;
        int shape_a = ct->shape0[tid], shape_b = ct->shape1[tid];
        if (shape_a == shape_b) continue;
        int body_a = -1, body_b = -1, mat_nonzero = 0;
        float restitution = 0.0f;
        if (shape_a >= 0) {
            mat_nonzero += 1;
            restitution += m->shape_material_restitution[shape_a];
            body_a = m->shape_body[shape_a];
        }
        if (shape_b >= 0) {
            mat_nonzero += 1;
            restitution += m->shape_material_restitution[shape_b];
            body_b = m->shape_body[shape_b];
        }
        if (mat_nonzero > 0) restitution /= float(mat_nonzero);
        if (body_a == body_b) continue;
        float m_inv_a = 0.0f, m_inv_b = 0.0f;
        mat33 I_inv_a, I_inv_b;
        transform X_wb_a_prev = transform_identity(), X_wb_b_prev = transform_identity();
        vec3 com_a(0.0f), com_b(0.0f), v_a(0.0f), v_b(0.0f), v_a_new(0.0f), v_b_new(0.0f);
        float inv_mass = 0.0f;
        if (body_a >= 0) {
            X_wb_a_prev = ldx(body_q_prev, body_a);
            m_inv_a = m->body_inv_mass[body_a];
            I_inv_a = ldm(m->body_inv_inertia, body_a);
            com_a = ld3(m->body_com, body_a);
        }
        if (body_b >= 0) {
            X_wb_b_prev = ldx(body_q_prev, body_b);
            m_inv_b = m->body_inv_mass[body_b];
            I_inv_b = ldm(m->body_inv_inertia, body_b);
            com_b = ld3(m->body_com, body_b);
        }
        // contact_surface_point (sim/contacts.py:97-115)
        vec3 bx_a = transform_point(X_wb_a_prev, ld3(ct->point0, tid) + ld3(ct->offset0, tid));
        vec3 bx_b = transform_point(X_wb_b_prev, ld3(ct->point1, tid) + ld3(ct->offset1, tid));
        vec3 n = ld3(ct->normal, tid);
        float d = dot(n, bx_b - bx_a);
        if (d >= 0.0f) continue;
        vec3 r_a = bx_a - transform_point(X_wb_a_prev, com_a);
        vec3 r_b = bx_b - transform_point(X_wb_b_prev, com_b);
        vec3 rxn_a(0.0f), rxn_b(0.0f);
        if (body_a >= 0) {
            int w = m->body_world[body_a];
            if (w < 0) w += m->world_count + 1;
            v_a = velocity_at_point(lds(body_qd_prev, body_a), r_a) + ld3(m->gravity, w) * dt;
            v_a_new = velocity_at_point(lds(body_qd, body_a), r_a);
            rxn_a = quat_rotate_inv(X_wb_a_prev.q, cross(r_a, n));
            inv_mass += m_inv_a + dot(rxn_a, I_inv_a * rxn_a);
        }
        if (body_b >= 0) {
            int w = m->body_world[body_b];
            if (w < 0) w += m->world_count + 1;
            v_b = velocity_at_point(lds(body_qd_prev, body_b), r_b) + ld3(m->gravity, w) * dt;
            v_b_new = velocity_at_point(lds(body_qd, body_b), r_b);
            rxn_b = quat_rotate_inv(X_wb_b_prev.q, cross(r_b, n));
            inv_mass += m_inv_b + dot(rxn_b, I_inv_b * rxn_b);
        }
        if (inv_mass == 0.0f) continue;
        float rel_vel_old = dot(n, v_b - v_a);
        float rel_vel_new = dot(n, v_b_new - v_a_new);
        if (rel_vel_old >= 0.0f) continue;
        float dv = (-rel_vel_new - restitution * rel_vel_old) / inv_mass;
        if (body_a >= 0) {
            float dv_a = -dv;
            vec3 dq = quat_rotate(X_wb_a_prev.q, I_inv_a * rxn_a * dv_a);
            adds(deltas, body_a, spatial(n * m_inv_a * dv_a, dq));
        }
        if (body_b >= 0) {
            float dv_b = dv;
            vec3 dq = quat_rotate(X_wb_b_prev.q, I_inv_b * rxn_b * dv_b);
            adds(deltas, body_b, spatial(n * m_inv_b * dv_b, dq));
        }
    }
}

// contact_force_out (nullable, [Cmax][6]): what SolverXPBD.update_contacts would write into contacts.force after this step
// (solver_xpbd.py:864-921, kernels.py:2464-2494); s_out->body_parent_f (nullable): kernels.py:2497-2544.
extern "C" void o_xpbd_step_report(const o_model* m, const o_xpbd_params* p, o_state* s_in, o_state* s_out,
                                   const o_control* c, const o_contacts* contacts, float dt, float* contact_force_out) {
    const int B = m->body_count;
    if (B == 0) return;
    std::vector<float> contact_impulse, contact_impulse_iter, joint_impulse;
    if (contacts && contact_force_out) {
        contact_impulse.assign(6 * (size_t)contacts->rigid_contact_max, 0.0f);
        contact_impulse_iter.assign(6 * (size_t)contacts->rigid_contact_max, 0.0f);
    }
    if (s_out->body_parent_f && m->joint_count > 0) joint_impulse.assign(6 * (size_t)m->joint_count, 0.0f);
    float* ji = joint_impulse.empty() ? nullptr : joint_impulse.data();
    // body_q_init / body_qd_init (solver_xpbd.py:414-416)
    std::vector<float> body_q_init, body_qd_init;
    if (p->enable_restitution || p->compute_body_velocity_from_position_delta) {
        body_q_init.assign(s_in->body_q, s_in->body_q + 7 * B);
        body_qd_init.assign(s_in->body_qd, s_in->body_qd + 6 * B);
    }
    std::vector<float> body_deltas(6 * B, 0.0f);
    std::vector<float> inv_weight;
    if (contacts && p->rigid_contact_con_weighting) inv_weight.assign(B, 0.0f);

    // apply_joint_forces into a clone of state_in.body_f  (solver_xpbd.py:420-451)
    std::vector<float> body_f_tmp(s_in->body_f, s_in->body_f + 6 * B);
    if (m->joint_count) apply_joint_forces(m, s_in->body_q, c->joint_f, dt, body_f_tmp.data(), ji);

    // integrate_bodies state_in -> state_out  (solver_xpbd.py:453-459)
    o_integrate_bodies(m, s_in->body_q, s_in->body_qd, body_f_tmp.data(), p->angular_damping, dt, s_out->body_q,
                       s_out->body_qd);

    // kinematic copy needs the *input* state; XPBD clobbers state_in as ping-pong scratch, so the
    // reference's copy_kinematic_body_state at the end reads whatever is left in state_in.  Kinematic
    // bodies are passed through unchanged by every kernel, so the value is still the input value.
    float* body_q = s_out->body_q;
    float* body_qd = s_out->body_qd;
    int counter = 0;  // _body_delta_counter

    auto apply = [&](const float* weights) {
        float *q_src, *qd_src, *q_dst, *qd_dst;
        if (counter == 0) {
            q_src = s_out->body_q; qd_src = s_out->body_qd; q_dst = s_in->body_q; qd_dst = s_in->body_qd;
        } else {
            q_src = s_in->body_q; qd_src = s_in->body_qd; q_dst = s_out->body_q; qd_dst = s_out->body_qd;
        }
        counter = 1 - counter;
        apply_body_deltas(m, q_src, qd_src, body_deltas.data(), weights, dt, q_dst, qd_dst);
        body_q = q_dst;
        body_qd = qd_dst;
    };

    for (int it = 0; it < p->iterations; ++it) {
        std::fill(body_deltas.begin(), body_deltas.end(), 0.0f);
        if (contacts) {
            if (!inv_weight.empty()) std::fill(inv_weight.begin(), inv_weight.end(), 0.0f);
            if (!contact_impulse_iter.empty()) std::fill(contact_impulse_iter.begin(), contact_impulse_iter.end(), 0.0f);
            solve_body_contact_positions(m, contacts, body_q, body_qd, p->rigid_contact_relaxation, dt, body_deltas.data(),
                                         inv_weight.empty() ? nullptr : inv_weight.data(),
                                         contact_impulse_iter.empty() ? nullptr : contact_impulse_iter.data());
            if (!contact_impulse_iter.empty())
                accumulate_weighted_contact_impulse(m, contacts, contact_impulse_iter.data(),
                                                    inv_weight.empty() ? nullptr : inv_weight.data(), contact_impulse.data());
            apply(inv_weight.empty() ? nullptr : inv_weight.data());
        }
        if (m->joint_count) {
            std::fill(body_deltas.begin(), body_deltas.end(), 0.0f);
            solve_body_joints(m, p, c, body_q, body_qd, dt, body_deltas.data(), ji);
            apply(nullptr);
        }
    }

    if (body_q != s_out->body_q) {
        std::memcpy(s_out->body_q, body_q, sizeof(float) * 7 * B);
        std::memcpy(s_out->body_qd, body_qd, sizeof(float) * 6 * B);
    }

    // update_body_velocities (xpbd/kernels.py:2547-2579, solver_xpbd.py:767-783): velocities from the position change of the step
    if (p->compute_body_velocity_from_position_delta)
        for (int tid = 0; tid < B; ++tid) {
            transform pose = ldx(s_out->body_q, tid), pose_prev = ldx(body_q_init.data(), tid);
            vec3 com = ld3(m->body_com, tid);
            vec3 x_com = pose.p + quat_rotate(pose.q, com);
            vec3 x_com_prev = pose_prev.p + quat_rotate(pose_prev.q, com);
            vec3 v = (x_com - x_com_prev) / dt;
            quat dq = pose.q * quat_inverse(pose_prev.q);
            vec3 omega = (2.0f / dt) * vec3(dq.x, dq.y, dq.z);
            if (dq.w < 0.0f) omega = -omega;
            sts(s_out->body_qd, tid, spatial(v, omega));
        }

    // restitution (solver_xpbd.py:784-858): uses the effective (kinematic -> 0) inverse mass / inertia of the model
    if (p->enable_restitution && contacts) {
        std::fill(body_deltas.begin(), body_deltas.end(), 0.0f);
        apply_rigid_restitution(m, contacts, s_out->body_q, s_out->body_qd, body_q_init.data(), body_qd_init.data(), dt,
                                body_deltas.data());
        for (int tid = 0; tid < B; ++tid) adds(s_out->body_qd, tid, lds(body_deltas.data(), tid));  // apply_body_delta_velocities
    }

    // copy_kinematic_body_state (kernels.py:19-32)
    for (int tid = 0; tid < B; ++tid) {
        if ((m->body_flags[tid] & BODY_KINEMATIC) == 0) continue;
        stx(s_out->body_q, tid, ldx(s_in->body_q, tid));
        sts(s_out->body_qd, tid, lds(s_in->body_qd, tid));
    }

    // convert_joint_impulse_to_parent_f (kernels.py:2497-2544, solver_xpbd.py:736-754)
    if (s_out->body_parent_f) {
        std::fill(s_out->body_parent_f, s_out->body_parent_f + 6 * B, 0.0f);
        if (ji) {
            float inv_dt = 1.0f / dt;
            for (int tid = 0; tid < m->joint_count; ++tid) {
                if (!m->joint_enabled[tid] || m->joint_type[tid] == FREE) continue;
                int id_c = m->joint_child[tid];
                if (id_c < 0) continue;
                spatial impulse = lds(ji, tid);
                adds(s_out->body_parent_f, id_c, spatial(impulse.top * inv_dt, impulse.bottom * inv_dt));
            }
        }
    }
    // convert_contact_impulse_to_force (kernels.py:2464-2494), as SolverXPBD.update_contacts does right after the step
    if (contact_force_out && contacts) {
        int count = contacts->rigid_contact_count[0];
        float inv_dt = 1.0f / dt;
        for (int tid = 0; tid < contacts->rigid_contact_max; ++tid) {
            spatial f;
            if (tid < count) {
                spatial impulse = lds(contact_impulse.data(), tid);
                f = spatial(impulse.top * inv_dt, impulse.bottom * inv_dt);
            }
            sts(contact_force_out, tid, f);
        }
    }
}

extern "C" void o_xpbd_step(const o_model* m, const o_xpbd_params* p, o_state* s_in, o_state* s_out, const o_control* c,
                            const o_contacts* contacts, float dt) {
    o_xpbd_step_report(m, p, s_in, s_out, c, contacts, dt, nullptr);
}

// substeps x { clear_forces; collide; xpbd step; swap } entirely in C (bench.py's cpu_baseline leg: one foreign call per
// env shard, so host threads scale without Python in the loop).  The result is in s0 for even substeps, s1 for odd.
extern "C" void o_xpbd_rollout(const o_model* m, const o_xpbd_params* p, o_state* s0, o_state* s1, const o_control* c,
                               o_contacts* contacts, float dt, int substeps) {
    o_state* a = s0;
    o_state* b = s1;
    for (int s = 0; s < substeps; ++s) {
        std::fill(a->body_f, a->body_f + 6 * m->body_count, 0.0f);
        o_collide(m, a->body_q, O_BP_EXPLICIT, contacts, nullptr, 0, nullptr, nullptr);
        o_xpbd_step(m, p, a, b, c, contacts, dt);
        std::swap(a, b);
    }
}
