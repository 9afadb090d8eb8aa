// TEST INFRASTRUCTURE ONLY -- CPU oracle: SolverSemiImplicit rigid path.
// Literal restatement (ascending-tid serial execution) of
//   joint_force / eval_body_joints     newton/_src/solvers/semi_implicit/kernels_body.py:17-520
//   eval_body_contact                  newton/_src/solvers/semi_implicit/kernels_contact.py:381-556
//   SolverSemiImplicit.step            newton/_src/solvers/semi_implicit/solver_semi_implicit.py:123-217
//   integrate_bodies                   newton/_src/solvers/solver.py:63-170 (shared, oracle_xpbd.cpp)
// warp builtins restated here: wp.quat_twist_angle_signed (kernels_body.py:206), wp.norm_huber
// (kernels_contact.py:537), wp.step -- restated builtins (wp_builtins.h); the solver itself is pinned by execution of the reference source
// (tests/test_reference_vectors.py: semi/*).
// wp.acos clamps its argument to [-1, 1] (Warp builtin semantics), which keeps the FIXED / PRISMATIC / BALL angular
// error finite when a normalised quaternion's w drifts a few ulp above 1.
// D6 joints with 2 or 3 angular axes decompose the relative rotation with quat_decompose (wp_builtins.h).
#include <vector>

#include "oracle_common.h"

using namespace orc;

namespace orc {
float joint_force(float q, float qd, float joint_target_q, float joint_target_qd, float target_ke, float target_kd,
                         float limit_lower, float limit_upper, float limit_ke, float limit_kd, float damping) {
    float limit_f = 0.0f, damping_f = 0.0f, target_f = 0.0f;
    target_f = target_ke * (joint_target_q - q) + target_kd * (joint_target_qd - qd);
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
}  // namespace orc

// signed twist angle of q about `axis`, wrapped to [-pi, pi]
static float quat_twist_angle_signed(vec3 axis, quat q) {
    const float pi = 3.14159265358979323846f;
    float a = q.x * axis.x + q.y * axis.y + q.z * axis.z;
    float angle = 2.0f * std::atan2(a, q.w);
    if (angle > pi) angle -= 2.0f * pi;
    if (angle < -pi) angle += 2.0f * pi;
    return angle;
}

static float norm_huber(vec3 v, float delta) {
    float a = dot(v, v);
    if (a <= delta * delta) return 0.5f * a;
    return delta * (std::sqrt(a) - 0.5f * delta);
}

static float dof_force(const o_model* m, const o_control* c, int dof, int tq, float q, float qd) {
    return joint_force(q, qd, c->joint_target_q[tq], c->joint_target_qd[dof], m->joint_target_ke[dof], m->joint_target_kd[dof],
                       m->joint_limit_lower[dof], m->joint_limit_upper[dof], m->joint_limit_ke[dof], m->joint_limit_kd[dof],
                       m->joint_damping[dof]);
}

static void eval_body_joints(const o_model* m, const o_control* c, const float* body_q, const float* body_qd,
                             float joint_attach_ke, float joint_attach_kd, float* body_f) {
    for (int tid = 0; tid < m->joint_count; ++tid) {
        int type = m->joint_type[tid];
        int c_child = m->joint_child[tid];
        int c_parent = m->joint_parent[tid];
        if (!m->joint_enabled[tid]) continue;
        int qd_start = m->joint_qd_start[tid];
        int target_q_start = m->joint_target_q_start[tid];
        const float* joint_f = c->joint_f;
        if (type == FREE || type == DISTANCE) {
            spatial wrench(vec3(joint_f[qd_start], joint_f[qd_start + 1], joint_f[qd_start + 2]),
                           vec3(joint_f[qd_start + 3], joint_f[qd_start + 4], joint_f[qd_start + 5]));
            adds(body_f, c_child, wrench);
            continue;
        }
        transform X_pj = ldx(m->joint_X_p, tid), X_cj = ldx(m->joint_X_c, tid);
        transform X_wp = X_pj;
        vec3 r_p, w_p, v_p;
        if (c_parent >= 0) {
            transform bq = ldx(body_q, c_parent);
            X_wp = bq * X_wp;
            r_p = X_wp.p - transform_point(bq, ld3(m->body_com, c_parent));
            spatial twist_p = lds(body_qd, c_parent);
            w_p = twist_p.bottom;
            v_p = twist_p.top + cross(w_p, r_p);
        }
        transform bqc = ldx(body_q, c_child);
        transform X_wc = bqc * X_cj;
        vec3 r_c = X_wc.p - transform_point(bqc, ld3(m->body_com, c_child));
        spatial twist_c = lds(body_qd, c_child);
        vec3 w_c = twist_c.bottom;
        vec3 v_c = twist_c.top + cross(w_c, r_c);

        int lin_axis_count = m->joint_dof_dim[2 * tid], ang_axis_count = m->joint_dof_dim[2 * tid + 1];
        vec3 x_p = X_wp.p, x_c = X_wc.p;
        quat q_p = X_wp.q, q_c = X_wc.q;
        vec3 x_err = x_c - x_p;
        quat r_err = quat_inverse(q_p) * q_c;
        vec3 v_err = v_c - v_p;
        vec3 w_err = w_c - w_p;
        vec3 t_total, f_total;
        const float angular_damping_scale = 0.01f;

        if (type == FIXED) {
            vec3 ang_err = normalize(vec3(r_err.x, r_err.y, r_err.z)) * std::acos(clampf(r_err.w, -1.0f, 1.0f)) * 2.0f;
            f_total += x_err * joint_attach_ke + v_err * joint_attach_kd;
            t_total += transform_vector(X_wp, ang_err) * joint_attach_ke + w_err * joint_attach_kd * angular_damping_scale;
        }
        if (type == PRISMATIC) {
            vec3 axis = ld3(m->joint_axis, qd_start);
            vec3 axis_p = transform_vector(X_wp, axis);
            float q = dot(x_err, axis_p);
            float qd = dot(v_err, axis_p);
            f_total = axis_p * (-joint_f[qd_start] - dof_force(m, c, qd_start, target_q_start, q, qd));
            vec3 ang_err = normalize(vec3(r_err.x, r_err.y, r_err.z)) * std::acos(clampf(r_err.w, -1.0f, 1.0f)) * 2.0f;
            f_total += (x_err - q * axis_p) * joint_attach_ke + (v_err - qd * axis_p) * joint_attach_kd;
            t_total += transform_vector(X_wp, ang_err) * joint_attach_ke + w_err * joint_attach_kd * angular_damping_scale;
        }
        if (type == REVOLUTE) {
            vec3 axis = ld3(m->joint_axis, qd_start);
            vec3 axis_p = transform_vector(X_wp, axis);
            vec3 axis_c = transform_vector(X_wc, axis);
            float q = quat_twist_angle_signed(axis, r_err);
            float qd = dot(w_err, axis_p);
            t_total = axis_p * (-joint_f[qd_start] - dof_force(m, c, qd_start, target_q_start, q, qd));
            vec3 swing_err = cross(axis_p, axis_c);
            f_total += x_err * joint_attach_ke + v_err * joint_attach_kd;
            t_total += swing_err * joint_attach_ke + (w_err - qd * axis_p) * joint_attach_kd * angular_damping_scale;
        }
        if (type == BALL) {
            f_total += x_err * joint_attach_ke + v_err * joint_attach_kd;
            for (int k = 0; k < 3; ++k) {
                vec3 axis_k = transform_vector(X_wp, ld3(m->joint_axis, qd_start + k));
                t_total += axis_k * (-joint_f[qd_start + k] + m->joint_damping[qd_start + k] * dot(axis_k, w_err));
            }
        }
        if (type == D6) {
            vec3 pos(0.0f), vel(0.0f);
            for (int k = 0; k < 3; ++k) {
                bool take = (k == 0 && lin_axis_count >= 1) || (k == 1 && lin_axis_count >= 2) || (k == 2 && lin_axis_count == 3);
                if (!take) continue;
                vec3 axis_k = transform_vector(X_wp, ld3(m->joint_axis, qd_start + k));
                float qk = dot(x_err, axis_k);
                float qdk = dot(v_err, axis_k);
                f_total += axis_k * (-joint_f[qd_start + k] - dof_force(m, c, qd_start + k, target_q_start + k, qk, qdk));
                pos += qk * axis_k;
                vel += qdk * axis_k;
            }
            f_total += (x_err - pos) * joint_attach_ke + (v_err - vel) * joint_attach_kd;
            if (ang_axis_count == 0) {
                vec3 ang_err = normalize(vec3(r_err.x, r_err.y, r_err.z)) * std::acos(clampf(r_err.w, -1.0f, 1.0f)) * 2.0f;
                t_total += transform_vector(X_wp, ang_err) * joint_attach_ke + w_err * joint_attach_kd * angular_damping_scale;
            }
            int i_0 = lin_axis_count + qd_start;
            int i_0_q = lin_axis_count + target_q_start;
            if (ang_axis_count == 1) {
                vec3 axis = ld3(m->joint_axis, i_0);
                vec3 axis_p = transform_vector(X_wp, axis);
                vec3 axis_c = transform_vector(X_wc, axis);
                float q = quat_twist_angle_signed(axis, r_err);
                float qd = dot(w_err, axis_p);
                t_total = axis_p * (-joint_f[i_0] - dof_force(m, c, i_0, i_0_q, q, qd));
                vec3 swing_err = cross(axis_p, axis_c);
                t_total += swing_err * joint_attach_ke + (w_err - qd * axis_p) * joint_attach_kd * angular_damping_scale;
            }
            if (ang_axis_count == 2 || ang_axis_count == 3) {  // kernels_body.py:371-515
                quat q_pc = quat_inverse(q_p) * q_c;
                vec3 angles = quat_decompose(q_pc);
                vec3 orig_axis_0 = ld3(m->joint_axis, i_0), orig_axis_1 = ld3(m->joint_axis, i_0 + 1);
                vec3 orig_axis_2 = ang_axis_count == 3 ? ld3(m->joint_axis, i_0 + 2) : cross(orig_axis_0, orig_axis_1);
                vec3 axis_0 = orig_axis_0;
                quat q_0 = quat_from_axis_angle(axis_0, angles.x);
                vec3 axis_1 = quat_rotate(q_0, orig_axis_1);
                quat q_1 = quat_from_axis_angle(axis_1, angles.y);
                vec3 axis_2 = quat_rotate(q_1 * q_0, orig_axis_2);
                axis_0 = transform_vector(X_wp, axis_0);
                axis_1 = transform_vector(X_wp, axis_1);
                axis_2 = transform_vector(X_wp, axis_2);
                t_total += axis_0 * (-joint_f[i_0] - dof_force(m, c, i_0, i_0_q, angles.x, dot(axis_0, w_err)));
                t_total += axis_1 * (-joint_f[i_0 + 1] - dof_force(m, c, i_0 + 1, i_0_q + 1, angles.y, dot(axis_1, w_err)));
                if (ang_axis_count == 3) {
                    t_total += axis_2 * (-joint_f[i_0 + 2] - dof_force(m, c, i_0 + 2, i_0_q + 2, angles.z, dot(axis_2, w_err)));
                } else {  // last axis (fixed): a stiff attachment spring
                    t_total += axis_2 * -orc::joint_force(angles.z, dot(axis_2, w_err), 0.0f, 0.0f, joint_attach_ke,
                                                          joint_attach_kd * angular_damping_scale, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
        }
        if (c_parent >= 0) adds(body_f, c_parent, spatial(f_total, t_total + cross(r_p, f_total)));
        subs(body_f, c_child, spatial(f_total, t_total + cross(r_c, f_total)));
    }
}

namespace orc {
void eval_body_contact(const o_model* m, const o_contacts* ct, const float* body_q, const float* body_qd,
                       float friction_smoothing, float* body_f) {
    int count = ct->rigid_contact_count[0];
    for (int tid = 0; tid < ct->rigid_contact_max; ++tid) {
        if (tid >= count) break;
        float ke = 0.0f, kd = 0.0f, kf = 0.0f, ka = 0.0f, mu = 0.0f;
        int mat_nonzero = 0;
        float margin_a = ct->margin0[tid], margin_b = ct->margin1[tid];
        int shape_a = ct->shape0[tid], shape_b = ct->shape1[tid];
        if (shape_a == shape_b) continue;
        int body_a = -1, body_b = -1;
        if (shape_a >= 0) {
            mat_nonzero += 1;
            ke += m->shape_material_ke[shape_a]; kd += m->shape_material_kd[shape_a]; kf += m->shape_material_kf[shape_a];
            ka += m->shape_material_ka[shape_a]; mu += m->shape_material_mu[shape_a];
            body_a = m->shape_body[shape_a];
        }
        if (shape_b >= 0) {
            mat_nonzero += 1;
            ke += m->shape_material_ke[shape_b]; kd += m->shape_material_kd[shape_b]; kf += m->shape_material_kf[shape_b];
            ka += m->shape_material_ka[shape_b]; mu += m->shape_material_mu[shape_b];
            body_b = m->shape_body[shape_b];
        }
        if (mat_nonzero > 0) {
            ke /= float(mat_nonzero); kd /= float(mat_nonzero); kf /= float(mat_nonzero);
            ka /= float(mat_nonzero); mu /= float(mat_nonzero);
        }
        // per-contact stiffness / damping / friction (kernels_contact.py:452-459)
        if (ct->stiffness) {
            float contact_ke = ct->stiffness[tid];
            ke = contact_ke > 0.0f ? contact_ke : ke;
            float contact_kd = ct->damping[tid];
            kd = contact_kd > 0.0f ? contact_kd : kd;
            float contact_mu = ct->friction_scale[tid];
            mu = contact_mu > 0.0f ? mu * contact_mu : mu;
        }
        vec3 n = -ld3(ct->normal, tid);
        vec3 bx_a = ld3(ct->point0, tid), bx_b = ld3(ct->point1, tid);
        vec3 r_a(0.0f), r_b(0.0f);
        if (body_a >= 0) {
            transform X = ldx(body_q, body_a);
            bx_a = transform_point(X, bx_a) - margin_a * n;
            r_a = bx_a - transform_point(X, ld3(m->body_com, body_a));
        }
        if (body_b >= 0) {
            transform X = ldx(body_q, body_b);
            bx_b = transform_point(X, bx_b) + margin_b * n;
            r_b = bx_b - transform_point(X, ld3(m->body_com, body_b));
        }
        float d = dot(n, bx_a - bx_b);
        if (d >= ka) continue;
        vec3 bv_a(0.0f), bv_b(0.0f);
        if (body_a >= 0) {
            spatial s = lds(body_qd, body_a);
            bv_a = s.top + cross(s.bottom, r_a);
        }
        if (body_b >= 0) {
            spatial s = lds(body_qd, body_b);
            bv_b = s.top + cross(s.bottom, r_b);
        }
        vec3 v = bv_a - bv_b;
        float vn = dot(n, v);
        vec3 vt = v - n * vn;
        float fn = d * ke;
        float fd = wmin(vn, 0.0f) * kd * (d < 0.0f ? 1.0f : 0.0f);  // wp.step(d)
        vec3 ft(0.0f);
        if (d < 0.0f) {
            float vs = norm_huber(vt, friction_smoothing);
            if (vs > 0.0f) {
                vec3 fr = vt / vs;
                ft = fr * wmin(kf * vs, -mu * (fn + fd));
            }
        }
        vec3 f_total = n * (fn + fd) + ft;
        if (body_a >= 0) subs(body_f, body_a, spatial(f_total, cross(r_a, f_total)));
        if (body_b >= 0) adds(body_f, body_b, spatial(f_total, cross(r_b, f_total)));
    }
}
}  // namespace orc

extern "C" void o_semi_implicit_step(const o_model* m, const o_semi_implicit_params* p, o_state* s_in, o_state* s_out,
                                     const o_control* c, const o_contacts* contacts, float dt) {
    const int B = m->body_count;
    if (B == 0) return;
    std::vector<float> body_f_work(s_in->body_f, s_in->body_f + 6 * B);
    if (m->joint_count)
        eval_body_joints(m, c, s_in->body_q, s_in->body_qd, p->joint_attach_ke, p->joint_attach_kd, body_f_work.data());
    if (contacts && contacts->rigid_contact_max)
        eval_body_contact(m, contacts, s_in->body_q, s_in->body_qd, p->friction_smoothing, body_f_work.data());
    o_integrate_bodies(m, s_in->body_q, s_in->body_qd, body_f_work.data(), p->angular_damping, dt, s_out->body_q,
                       s_out->body_qd);
}
