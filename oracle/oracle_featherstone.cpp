// TEST INFRASTRUCTURE ONLY -- CPU oracle: SolverFeatherstone rigid path (generalized coordinates, CRBA + Cholesky).
// Literal restatement (ascending-tid serial execution, one "thread" per articulation / joint as in the reference) of
//   compute_spatial_inertia / compute_com_transforms         newton/_src/solvers/featherstone/kernels.py:21-52
//   transform_spatial_inertia                                 kernels.py:66-139
//   jcalc_transform / jcalc_motion / jcalc_tau / jcalc_integrate   kernels.py:142-630
//   compute_link_transform / eval_rigid_fk                    kernels.py:633-728
//   spatial_cross / spatial_cross_dual / compute_link_velocity / eval_rigid_id   kernels.py:731-866,1241-1317
//   accumulate_free_distance_joint_f_to_body_force            kernels.py:893-921
//   convert_free_distance_joint_qd_public_to_internal / _internal_to_public / joint_f_public_to_internal
//                                                             kernels.py:924-975,1015-1088
//   eval_rigid_tau / eval_rigid_jacobian / eval_rigid_mass    kernels.py:1320-1501
//   dense_gemm / dense_cholesky / dense_subs                  kernels.py:1504-1565,1690-1797
//   integrate_generalized_joints                              kernels.py:1849-1893
//   eval_single_articulation_fk_with_velocity_conversion      kernels.py:1987-2150
//   SolverFeatherstone.step                                   solver_featherstone.py:462-1066
//   eval_body_contact (shared with SolverSemiImplicit)        semi_implicit/kernels_contact.py:381-556
//   transform_twist / velocity_at_point                       newton/_src/math/spatial.py:53-130
// Scope: PRISMATIC, REVOLUTE, BALL, FIXED, FREE / DISTANCE (root and descendant), D6 (up to three angular axes), kinematic
// roots, update_mass_matrix_interval.  Pinned by execution of the reference solver source (tests/test_reference_vectors.py: fs/*); the Warp builtins underneath stay restated (wp_builtins.h).
#include <vector>

#include "oracle_common.h"

using namespace orc;

namespace orc {
float joint_force(float q, float qd, float joint_target_q, float joint_target_qd, float target_ke, float target_kd,
                  float limit_lower, float limit_upper, float limit_ke, float limit_kd, float damping);
void eval_body_contact(const o_model* m, const o_contacts* ct, const float* body_q, const float* body_qd,
                       float friction_smoothing, float* body_f);
}  // namespace orc

namespace {

struct mat66 {
    float a[6][6];
    mat66() {
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) a[i][j] = 0.0f;
    }
};
mat66 mul(const mat66& A, const mat66& B) {
    mat66 C;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            float sum = 0.0f;
            for (int k = 0; k < 6; ++k) sum += A.a[i][k] * B.a[k][j];
            C.a[i][j] = sum;
        }
    return C;
}
mat66 transpose(const mat66& A) {
    mat66 T;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) T.a[i][j] = A.a[j][i];
    return T;
}
float sget(const spatial& s, int i) { return i < 3 ? s.top[i] : s.bottom[i - 3]; }
spatial mul(const mat66& A, const spatial& v) {
    float r[6];
    for (int i = 0; i < 6; ++i) {
        float sum = 0.0f;
        for (int j = 0; j < 6; ++j) sum += A.a[i][j] * sget(v, j);
        r[i] = sum;
    }
    return spatial(vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]));
}
float sdot(const spatial& a, const spatial& b) {
    float s = 0.0f;
    for (int i = 0; i < 6; ++i) s += sget(a, i) * sget(b, i);
    return s;
}

// math/spatial.py:82-104 (Newton layout (linear, angular))
spatial transform_twist(const transform& t, const spatial& x) {
    vec3 w = quat_rotate(t.q, x.bottom);
    vec3 v = quat_rotate(t.q, x.top) + cross(t.p, w);
    return spatial(v, w);
}
vec3 com_twist_to_point_velocity(const spatial& qd, const transform& X_wb, vec3 com, vec3 point) {
    return velocity_at_point(qd, point - transform_point(X_wb, com));
}
spatial origin_twist_to_com_twist(const spatial& qd, const transform& X_wb, vec3 com) {
    return spatial(velocity_at_point(qd, transform_vector(X_wb, com)), qd.bottom);
}

// kernels.py:66-139
mat66 transform_spatial_inertia(const transform& t, const mat66& I) {
    transform t_inv = transform_inverse(t);
    quat q = t_inv.q;
    vec3 p = t_inv.p;
    vec3 r1 = quat_rotate(q, vec3(1.0f, 0.0f, 0.0f));
    vec3 r2 = quat_rotate(q, vec3(0.0f, 1.0f, 0.0f));
    vec3 r3 = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
    mat33 R = matrix_from_cols(r1, r2, r3);
    mat33 S = skew(p) * R;
    mat66 T;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T.a[i][j] = R(i, j);
            T.a[i][j + 3] = S(i, j);
            T.a[i + 3][j + 3] = R(i, j);
        }
    return mul(mul(transpose(T), I), T);
}

spatial spatial_cross(const spatial& a, const spatial& b) {
    vec3 w = cross(a.bottom, b.bottom);
    vec3 v = cross(a.bottom, b.top) + cross(a.top, b.bottom);
    return spatial(v, w);
}
spatial spatial_cross_dual(const spatial& a, const spatial& b) {
    vec3 w = cross(a.bottom, b.bottom) + cross(a.top, b.top);
    vec3 v = cross(a.bottom, b.top);
    return spatial(v, w);
}

int dof_end(const o_model* m, int j) { return j + 1 < m->joint_count ? m->joint_qd_start[j + 1] : m->dof_count; }

// rotation and transported angular axes of a D6 joint with 2 / 3 angular axes (transform_2d/3d_rotational_axes,
// compute_2d/3d_rotational_dofs: newton/_src/sim/articulation.py:36-83,127-178)
quat d6_multi_angular(const o_model* m, int ang, int ia, const float* joint_q, int iq, vec3& a0, vec3& a1, vec3& a2) {
    vec3 axis_0 = ld3(m->joint_axis, ia), axis_1 = ld3(m->joint_axis, ia + 1);
    if (ang == 2) {
        quat q_off = quat_from_matrix(matrix_from_cols(axis_0, axis_1, cross(axis_0, axis_1)));
        vec3 local_0 = quat_rotate(q_off, vec3(1.0f, 0.0f, 0.0f)), local_1 = quat_rotate(q_off, vec3(0.0f, 1.0f, 0.0f));
        a0 = local_0;
        quat q_0 = quat_from_axis_angle(a0, joint_q[iq]);
        a1 = quat_rotate(q_0, local_1);
        a2 = vec3();
        quat q_1 = quat_from_axis_angle(a1, joint_q[iq + 1]);
        return q_1 * q_0;
    }
    vec3 axis_2 = ld3(m->joint_axis, ia + 2);
    a0 = axis_0;
    quat q_0 = quat_from_axis_angle(a0, joint_q[iq]);
    a1 = quat_rotate(q_0, axis_1);
    quat q_1 = quat_from_axis_angle(a1, joint_q[iq + 1]);
    a2 = quat_rotate(q_1 * q_0, axis_2);
    quat q_2 = quat_from_axis_angle(a2, joint_q[iq + 2]);
    return q_2 * q_1 * q_0;
}

// kernels.py:142-239
transform jcalc_transform(const o_model* m, int type, int axis_start, int lin, int ang, const float* joint_q, int q_start) {
    if (type == PRISMATIC) return transform(ld3(m->joint_axis, axis_start) * joint_q[q_start], quat_identity());
    if (type == REVOLUTE) return transform(vec3(), quat_from_axis_angle(ld3(m->joint_axis, axis_start), joint_q[q_start]));
    if (type == BALL)
        return transform(vec3(), quat(joint_q[q_start], joint_q[q_start + 1], joint_q[q_start + 2], joint_q[q_start + 3]));
    if (type == FIXED) return transform_identity();
    if (type == FREE || type == DISTANCE)
        return transform(vec3(joint_q[q_start], joint_q[q_start + 1], joint_q[q_start + 2]),
                         quat(joint_q[q_start + 3], joint_q[q_start + 4], joint_q[q_start + 5], joint_q[q_start + 6]));
    if (type == D6) {
        vec3 pos(0.0f);
        quat rot = quat_identity();
        if (lin > 0) pos += ld3(m->joint_axis, axis_start + 0) * joint_q[q_start + 0];
        if (lin > 1) pos += ld3(m->joint_axis, axis_start + 1) * joint_q[q_start + 1];
        if (lin > 2) pos += ld3(m->joint_axis, axis_start + 2) * joint_q[q_start + 2];
        if (ang == 1) rot = quat_from_axis_angle(ld3(m->joint_axis, axis_start + lin), joint_q[q_start + lin]);
        if (ang >= 2) {
            vec3 a0, a1, a2;
            rot = d6_multi_angular(m, ang, axis_start + lin, joint_q, q_start + lin, a0, a1, a2);
        }
        return transform(pos, rot);
    }
    return transform_identity();
}

// kernels.py:242-380; returns v_j_s; c_app_s = the apparent derivative of the motion subspace (non-zero only for D6 joints
// with >= 2 angular axes, whose transported axes depend on the joint coordinates)
spatial jcalc_motion(const o_model* m, int type, int lin, int ang, const transform& X_sc, const float* joint_q, int q_start,
                     const float* joint_qd, int qd_start, spatial* joint_S_s, spatial& c_app_s) {
    c_app_s = spatial();
    if (type == PRISMATIC) {
        spatial S_s = transform_twist(X_sc, spatial(ld3(m->joint_axis, qd_start), vec3()));
        joint_S_s[qd_start] = S_s;
        return S_s * joint_qd[qd_start];
    }
    if (type == REVOLUTE) {
        spatial S_s = transform_twist(X_sc, spatial(vec3(), ld3(m->joint_axis, qd_start)));
        joint_S_s[qd_start] = S_s;
        return S_s * joint_qd[qd_start];
    }
    if (type == D6) {
        spatial v_j_s;
        for (int k = 0; k < 3; ++k)
            if (lin > k) {
                spatial S_s = transform_twist(X_sc, spatial(ld3(m->joint_axis, qd_start + k), vec3()));
                v_j_s = v_j_s + S_s * joint_qd[qd_start + k];
                joint_S_s[qd_start + k] = S_s;
            }
        int iqd = qd_start + lin;
        if (ang == 1) {
            spatial S_s = transform_twist(X_sc, spatial(vec3(), ld3(m->joint_axis, iqd)));
            v_j_s = v_j_s + S_s * joint_qd[iqd];
            joint_S_s[iqd] = S_s;
        }
        if (ang >= 2) {
            vec3 a0, a1, a2;
            d6_multi_angular(m, ang, iqd, joint_q, q_start + lin, a0, a1, a2);
            float qd0 = joint_qd[iqd], qd1 = joint_qd[iqd + 1];
            spatial S_0 = transform_twist(X_sc, spatial(vec3(), a0)), S_1 = transform_twist(X_sc, spatial(vec3(), a1));
            vec3 c_app_ang;
            if (ang == 2) {
                v_j_s = v_j_s + (S_0 * qd0 + S_1 * qd1);
                joint_S_s[iqd] = S_0;
                joint_S_s[iqd + 1] = S_1;
                c_app_ang += cross(a0, a1) * (qd0 * qd1);
            } else {
                float qd2 = joint_qd[iqd + 2];
                spatial S_2 = transform_twist(X_sc, spatial(vec3(), a2));
                v_j_s = v_j_s + (S_0 * qd0 + S_1 * qd1 + S_2 * qd2);
                joint_S_s[iqd] = S_0;
                joint_S_s[iqd + 1] = S_1;
                joint_S_s[iqd + 2] = S_2;
                c_app_ang += cross(a0, a1) * (qd0 * qd1);
                c_app_ang += cross(a0, a2) * (qd0 * qd2);
                c_app_ang += cross(a1, a2) * (qd1 * qd2);
            }
            c_app_s = transform_twist(X_sc, spatial(vec3(), c_app_ang));
        }
        return v_j_s;
    }
    if (type == BALL) {
        spatial S_0 = transform_twist(X_sc, spatial(vec3(), vec3(1.0f, 0.0f, 0.0f)));
        spatial S_1 = transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 1.0f, 0.0f)));
        spatial S_2 = transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 0.0f, 1.0f)));
        joint_S_s[qd_start + 0] = S_0;
        joint_S_s[qd_start + 1] = S_1;
        joint_S_s[qd_start + 2] = S_2;
        return S_0 * joint_qd[qd_start + 0] + S_1 * joint_qd[qd_start + 1] + S_2 * joint_qd[qd_start + 2];
    }
    if (type == FIXED) return spatial();
    if (type == FREE || type == DISTANCE) {
        spatial v_j_s = transform_twist(X_sc, spatial(vec3(joint_qd[qd_start + 0], joint_qd[qd_start + 1], joint_qd[qd_start + 2]),
                                                      vec3(joint_qd[qd_start + 3], joint_qd[qd_start + 4], joint_qd[qd_start + 5])));
        joint_S_s[qd_start + 0] = transform_twist(X_sc, spatial(vec3(1.0f, 0.0f, 0.0f), vec3()));
        joint_S_s[qd_start + 1] = transform_twist(X_sc, spatial(vec3(0.0f, 1.0f, 0.0f), vec3()));
        joint_S_s[qd_start + 2] = transform_twist(X_sc, spatial(vec3(0.0f, 0.0f, 1.0f), vec3()));
        joint_S_s[qd_start + 3] = transform_twist(X_sc, spatial(vec3(), vec3(1.0f, 0.0f, 0.0f)));
        joint_S_s[qd_start + 4] = transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 1.0f, 0.0f)));
        joint_S_s[qd_start + 5] = transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 0.0f, 1.0f)));
        return v_j_s;
    }
    return spatial();
}

// kernels.py:383-461
void jcalc_tau(const o_model* m, const o_control* c, int type, const spatial* joint_S_s, const float* joint_q, const float* joint_qd,
               const float* joint_f, int coord_start, int dof_start, int target_q_start, int lin, int ang, const spatial& body_f_s,
               float* tau) {
    if (type == BALL) {
        for (int i = 0; i < 3; ++i) {
            int j = dof_start + i;
            float passive_f = -m->joint_damping[j] * joint_qd[j];
            tau[j] = -sdot(joint_S_s[j], body_f_s) + joint_f[j] + passive_f;
        }
        return;
    }
    if (type == FREE || type == DISTANCE) {
        for (int i = 0; i < 6; ++i) tau[dof_start + i] = -sdot(joint_S_s[dof_start + i], body_f_s) + joint_f[dof_start + i];
        return;
    }
    if (type == PRISMATIC || type == REVOLUTE || type == D6) {
        int axis_count = lin + ang;
        for (int i = 0; i < axis_count; ++i) {
            int j = dof_start + i;
            float q = joint_q[coord_start + i];
            float qd = joint_qd[j];
            float drive_f = joint_force(q, qd, c->joint_target_q[target_q_start + i], c->joint_target_qd[j], m->joint_target_ke[j],
                                        m->joint_target_kd[j], m->joint_limit_lower[j], m->joint_limit_upper[j],
                                        m->joint_limit_ke[j], m->joint_limit_kd[j], m->joint_damping[j]);
            tau[j] = -sdot(joint_S_s[j], body_f_s) + drive_f + joint_f[j];
        }
    }
}

// kernels.py:464-630
void jcalc_integrate(const o_model* m, int parent, const transform& joint_X_c, vec3 body_com_child, int type, const float* joint_q,
                     const float* joint_qd, const float* joint_qdd, int coord_start, int dof_start, int lin, int ang, float dt,
                     float* joint_q_new, float* joint_qd_new) {
    if (type == FIXED) return;
    if (type == PRISMATIC || type == REVOLUTE) {
        float qd_new = joint_qd[dof_start] + joint_qdd[dof_start] * dt;
        float q_new = joint_q[coord_start] + qd_new * dt;
        joint_qd_new[dof_start] = qd_new;
        joint_q_new[coord_start] = q_new;
        return;
    }
    if (type == BALL) {
        vec3 m_j(joint_qdd[dof_start], joint_qdd[dof_start + 1], joint_qdd[dof_start + 2]);
        vec3 w_j(joint_qd[dof_start], joint_qd[dof_start + 1], joint_qd[dof_start + 2]);
        quat r_j(joint_q[coord_start], joint_q[coord_start + 1], joint_q[coord_start + 2], joint_q[coord_start + 3]);
        vec3 w_j_new = w_j + m_j * dt;
        quat drdt_j = quat(w_j_new.x, w_j_new.y, w_j_new.z, 0.0f) * r_j * 0.5f;
        quat r_j_new = normalize(r_j + drdt_j * dt);
        joint_q_new[coord_start + 0] = r_j_new.x;
        joint_q_new[coord_start + 1] = r_j_new.y;
        joint_q_new[coord_start + 2] = r_j_new.z;
        joint_q_new[coord_start + 3] = r_j_new.w;
        joint_qd_new[dof_start + 0] = w_j_new.x;
        joint_qd_new[dof_start + 1] = w_j_new.y;
        joint_qd_new[dof_start + 2] = w_j_new.z;
        return;
    }
    if ((type == FREE || type == DISTANCE) && parent >= 0) {
        // descendants stay in the internal parent-origin coordinates during the step (kernels.py:574-611); the public COM
        // convention comes back at the solver boundary
        vec3 a_s(joint_qdd[dof_start], joint_qdd[dof_start + 1], joint_qdd[dof_start + 2]);
        vec3 m_s(joint_qdd[dof_start + 3], joint_qdd[dof_start + 4], joint_qdd[dof_start + 5]);
        vec3 v_s(joint_qd[dof_start], joint_qd[dof_start + 1], joint_qd[dof_start + 2]);
        vec3 w_s(joint_qd[dof_start + 3], joint_qd[dof_start + 4], joint_qd[dof_start + 5]);
        w_s = w_s + m_s * dt;
        v_s = v_s + a_s * dt;
        vec3 p_s(joint_q[coord_start], joint_q[coord_start + 1], joint_q[coord_start + 2]);
        vec3 dpdt_s = v_s + cross(w_s, p_s);
        quat r_s(joint_q[coord_start + 3], joint_q[coord_start + 4], joint_q[coord_start + 5], joint_q[coord_start + 6]);
        quat drdt_s = quat(w_s.x, w_s.y, w_s.z, 0.0f) * r_s * 0.5f;
        vec3 p_s_new = p_s + dpdt_s * dt;
        quat r_s_new = normalize(r_s + drdt_s * dt);
        joint_q_new[coord_start + 0] = p_s_new.x;
        joint_q_new[coord_start + 1] = p_s_new.y;
        joint_q_new[coord_start + 2] = p_s_new.z;
        joint_q_new[coord_start + 3] = r_s_new.x;
        joint_q_new[coord_start + 4] = r_s_new.y;
        joint_q_new[coord_start + 5] = r_s_new.z;
        joint_q_new[coord_start + 6] = r_s_new.w;
        joint_qd_new[dof_start + 0] = v_s.x;
        joint_qd_new[dof_start + 1] = v_s.y;
        joint_qd_new[dof_start + 2] = v_s.z;
        joint_qd_new[dof_start + 3] = w_s.x;
        joint_qd_new[dof_start + 4] = w_s.y;
        joint_qd_new[dof_start + 5] = w_s.z;
        return;
    }
    if (type == FREE || type == DISTANCE) {  // root (parent < 0)
        vec3 a_parent(joint_qdd[dof_start], joint_qdd[dof_start + 1], joint_qdd[dof_start + 2]);
        vec3 alpha(joint_qdd[dof_start + 3], joint_qdd[dof_start + 4], joint_qdd[dof_start + 5]);
        vec3 v_parent(joint_qd[dof_start], joint_qd[dof_start + 1], joint_qd[dof_start + 2]);
        vec3 omega(joint_qd[dof_start + 3], joint_qd[dof_start + 4], joint_qd[dof_start + 5]);
        vec3 p(joint_q[coord_start], joint_q[coord_start + 1], joint_q[coord_start + 2]);
        quat r(joint_q[coord_start + 3], joint_q[coord_start + 4], joint_q[coord_start + 5], joint_q[coord_start + 6]);
        vec3 r_com_joint = transform_point(transform_inverse(joint_X_c), body_com_child);
        vec3 x_com = p + quat_rotate(r, r_com_joint);
        vec3 v_com = v_parent + cross(omega, x_com);
        vec3 a_com = a_parent + cross(alpha, x_com) + cross(omega, v_com);
        vec3 omega_new = omega + alpha * dt;
        vec3 v_com_new = v_com + a_com * dt;
        quat drdt = quat(omega_new.x, omega_new.y, omega_new.z, 0.0f) * r * 0.5f;
        quat r_new = normalize(r + drdt * dt);
        vec3 x_com_new = x_com + v_com_new * dt;
        vec3 p_new = x_com_new - quat_rotate(r_new, r_com_joint);
        vec3 v_parent_new = v_com_new - cross(omega_new, x_com_new);
        joint_q_new[coord_start + 0] = p_new.x;
        joint_q_new[coord_start + 1] = p_new.y;
        joint_q_new[coord_start + 2] = p_new.z;
        joint_q_new[coord_start + 3] = r_new.x;
        joint_q_new[coord_start + 4] = r_new.y;
        joint_q_new[coord_start + 5] = r_new.z;
        joint_q_new[coord_start + 6] = r_new.w;
        joint_qd_new[dof_start + 0] = v_parent_new.x;
        joint_qd_new[dof_start + 1] = v_parent_new.y;
        joint_qd_new[dof_start + 2] = v_parent_new.z;
        joint_qd_new[dof_start + 3] = omega_new.x;
        joint_qd_new[dof_start + 4] = omega_new.y;
        joint_qd_new[dof_start + 5] = omega_new.z;
        return;
    }
    if (type == D6) {
        for (int i = 0; i < lin + ang; ++i) {
            float qd_new = joint_qd[dof_start + i] + joint_qdd[dof_start + i] * dt;
            float q_new = joint_q[coord_start + i] + qd_new * dt;
            joint_qd_new[dof_start + i] = qd_new;
            joint_q_new[coord_start + i] = q_new;
        }
    }
}

// FREE/DISTANCE anchor offset shared by the public<->internal velocity conversions (kernels.py:924-1066)
vec3 free_joint_com_offset(const o_model* m, int joint_id, const float* body_q) {
    int parent = m->joint_parent[joint_id], child = m->joint_child[joint_id];
    transform X_wpj = ldx(m->joint_X_p, joint_id);
    if (parent >= 0) X_wpj = ldx(body_q, parent) * X_wpj;
    vec3 x_child_com_world = transform_point(ldx(body_q, child), ld3(m->body_com, child));
    return quat_rotate_inv(X_wpj.q, x_child_com_world - X_wpj.p);
}

}  // namespace

// test probe: when armed, the next step copies the joint-space inertia H of articulation 0 (before the armature is added
// and before factorisation) into the caller's buffer -- pinned against the closed forms of
// newton/tests/test_jacobian_mass_matrix.py:413-470
static float* g_probe_H = nullptr;
static int g_probe_cap = 0;
extern "C" void o_featherstone_probe_H(float* out, int cap) {
    g_probe_H = out;
    g_probe_cap = cap;
}

extern "C" void o_featherstone_step(const o_model* m, const o_featherstone_params* prm, o_state* s_in, o_state* s_out,
                                    const o_control* c, const o_contacts* contacts, float dt) {
    const int B = m->body_count, J = m->joint_count, D = m->dof_count;
    if (B == 0 || J == 0) return;
    // _allocate_model_aux_vars (solver_featherstone.py:359-389)
    std::vector<mat66> body_I_m(B), body_I_s(B);
    std::vector<transform> body_X_com(B), body_q_com(B);
    for (int b = 0; b < B; ++b) {
        mat33 I = ldm(m->body_inertia, b);
        float mass = m->body_mass[b];
        for (int i = 0; i < 3; ++i) {
            body_I_m[b].a[i][i] = mass;
            for (int j = 0; j < 3; ++j) body_I_m[b].a[i + 3][j + 3] = I(i, j);
        }
        body_X_com[b] = transform(ld3(m->body_com, b), quat_identity());
    }
    std::vector<float> joint_qdd(D, 0.0f), joint_tau(D, 0.0f), qd_internal_in(D), qd_internal_out(D), joint_f_internal(D);
    std::vector<spatial> joint_S_s(D), body_qd_fk(B), body_v_s(B), body_a_s(B), body_f_s(B), body_ft_s(B);
    std::vector<vec3> body_solve_origin(B);
    float* body_q = s_in->body_q;

    // eval_rigid_fk: refreshes state_in.body_q from state_in.joint_q (solver_featherstone.py:492-514)
    for (int a = 0; a < m->articulation_count; ++a)
        for (int i = m->articulation_start[a]; i < m->articulation_end[a]; ++i) {
            int parent = m->joint_parent[i], child = m->joint_child[i];
            transform X_wpj = ldx(m->joint_X_p, i);
            if (parent >= 0) X_wpj = ldx(body_q, parent) * X_wpj;
            transform X_j = jcalc_transform(m, m->joint_type[i], m->joint_qd_start[i], m->joint_dof_dim[2 * i],
                                            m->joint_dof_dim[2 * i + 1], s_in->joint_q, m->joint_q_start[i]);
            transform X_wcj = X_wpj * X_j;
            transform X_wc = X_wcj * transform_inverse(ldx(m->joint_X_c, i));
            stx(body_q, child, X_wc);
            body_q_com[child] = X_wc * body_X_com[child];
        }

    // descendant_body_q_prev (solver_featherstone.py:481,514-516): the poses of the start-of-step FK
    std::vector<float> body_q_prev(body_q, body_q + 7 * B);

    // body_f_ext = state_in.body_f + FREE/DISTANCE joint_f routed as COM wrenches
    std::vector<float> body_f(s_in->body_f, s_in->body_f + 6 * B);
    for (int j = 0; j < J; ++j) {
        int t = m->joint_type[j];
        if (t != FREE && t != DISTANCE) continue;
        int qs = m->joint_qd_start[j];
        adds(body_f.data(), m->joint_child[j],
             spatial(vec3(c->joint_f[qs], c->joint_f[qs + 1], c->joint_f[qs + 2]),
                     vec3(c->joint_f[qs + 3], c->joint_f[qs + 4], c->joint_f[qs + 5])));
    }

    // convert_free_distance_joint_qd_public_to_internal / joint_f_public_to_internal
    for (int j = 0; j < J; ++j) {
        int qs = m->joint_qd_start[j], qe = dof_end(m, j), t = m->joint_type[j];
        if (t != FREE && t != DISTANCE) {
            for (int i = qs; i < qe; ++i) {
                qd_internal_in[i] = s_in->joint_qd[i];
                joint_f_internal[i] = c->joint_f[i];
            }
            continue;
        }
        vec3 r = free_joint_com_offset(m, j, body_q);
        vec3 v_com(s_in->joint_qd[qs], s_in->joint_qd[qs + 1], s_in->joint_qd[qs + 2]);
        vec3 omega(s_in->joint_qd[qs + 3], s_in->joint_qd[qs + 4], s_in->joint_qd[qs + 5]);
        vec3 v_int = v_com - cross(omega, r);
        qd_internal_in[qs + 0] = v_int.x; qd_internal_in[qs + 1] = v_int.y; qd_internal_in[qs + 2] = v_int.z;
        qd_internal_in[qs + 3] = omega.x; qd_internal_in[qs + 4] = omega.y; qd_internal_in[qs + 5] = omega.z;
        for (int i = qs; i < qe; ++i) joint_f_internal[i] = 0.0f;
    }

    // eval_rigid_id (kernels.py:1241-1317) with compute_link_velocity (kernels.py:764-866)
    for (int a = 0; a < m->articulation_count; ++a) {
        int start = m->articulation_start[a], end = m->articulation_end[a];
        vec3 solve_origin;
        if (start < end) {
            int rt = m->joint_type[start];
            if (rt == FREE || rt == DISTANCE) solve_origin = body_q_com[m->joint_child[start]].p;
        }
        for (int i = start; i < end; ++i) {
            int type = m->joint_type[i], child = m->joint_child[i], parent = m->joint_parent[i];
            int qd_start = m->joint_qd_start[i];
            transform X_wpj = ldx(m->joint_X_p, i);
            if (parent >= 0) X_wpj = ldx(body_q, parent) * X_wpj;
            transform X_wpj_s(X_wpj.p - solve_origin, X_wpj.q);
            spatial c_app_s;
            spatial v_j_s = jcalc_motion(m, type, m->joint_dof_dim[2 * i], m->joint_dof_dim[2 * i + 1], X_wpj_s, s_in->joint_q,
                                         m->joint_q_start[i], qd_internal_in.data(), qd_start, joint_S_s.data(), c_app_s);
            spatial v_parent_s, a_parent_s;
            if (parent >= 0) {
                v_parent_s = body_v_s[parent];
                a_parent_s = body_a_s[parent];
            }
            spatial v_s = v_parent_s + v_j_s;
            spatial a_s = a_parent_s + spatial_cross(v_s, v_j_s) + c_app_s;
            transform X_sm = body_q_com[child];
            vec3 x_com_s = X_sm.p - solve_origin;
            body_solve_origin[child] = solve_origin;
            const mat66& I_m = body_I_m[child];
            float mass = I_m.a[0][0];
            int world_idx = m->body_world[child];
            if (world_idx < 0) world_idx = m->world_count;  // gravity[-1] = last entry
            vec3 f_g = mass * ld3(m->gravity, world_idx);
            spatial f_g_s(f_g, cross(x_com_s, f_g));
            transform X_sm_s(x_com_s, X_sm.q);
            mat66 I_s = transform_spatial_inertia(X_sm_s, I_m);
            spatial f_b_s = mul(I_s, a_s) + spatial_cross_dual(v_s, mul(I_s, v_s));
            vec3 omega_world = v_s.bottom;
            vec3 v_com_world = v_s.top + cross(omega_world, x_com_s);
            body_qd_fk[child] = spatial(v_com_world, omega_world);
            body_v_s[child] = v_s;
            body_a_s[child] = a_s;
            body_f_s[child] = f_b_s - f_g_s;
            body_I_s[child] = I_s;
        }
    }

    // eval_body_contact on (state_in.body_q, body_qd_fk) -> body_f (solver_featherstone.py:646-676)
    if (contacts && contacts->rigid_contact_max) {
        std::vector<float> qd_fk(6 * B);
        for (int b = 0; b < B; ++b) sts(qd_fk.data(), b, body_qd_fk[b]);
        eval_body_contact(m, contacts, body_q, qd_fk.data(), prm->friction_smoothing, body_f.data());
    }

    // zero_kinematic_body_forces (kernels.py:54-63, solver_featherstone.py:682-689)
    auto kinematic_joint = [&](int j) { return (m->body_flags[m->joint_child[j]] & BODY_KINEMATIC) != 0; };
    for (int b = 0; b < B; ++b)
        if (m->body_flags[b] & BODY_KINEMATIC) sts(body_f.data(), b, spatial());

    // eval_rigid_tau (kernels.py:1320-1419)
    for (int a = 0; a < m->articulation_count; ++a) {
        int start = m->articulation_start[a], end = m->articulation_end[a];
        for (int offset = 0; offset < end - start; ++offset) {
            int i = end - offset - 1;
            int parent = m->joint_parent[i], child = m->joint_child[i];
            spatial f_ext_public = lds(body_f.data(), child);
            vec3 force = f_ext_public.top, torque_com = f_ext_public.bottom;
            vec3 x_com_s = body_q_com[child].p - body_solve_origin[child];
            spatial f_ext = -spatial(force, torque_com + cross(x_com_s, force));
            sts(body_f.data(), child, f_ext);
            spatial f_s = body_f_s[child] + body_ft_s[child] + f_ext;
            jcalc_tau(m, c, m->joint_type[i], joint_S_s.data(), s_in->joint_q, qd_internal_in.data(), joint_f_internal.data(),
                      m->joint_q_start[i], m->joint_qd_start[i], m->joint_target_q_start[i], m->joint_dof_dim[2 * i],
                      m->joint_dof_dim[2 * i + 1], f_s, joint_tau.data());
            if (parent >= 0) body_ft_s[parent] = body_ft_s[parent] + f_s;
        }
    }

    // compute_body_parent_f (kernels.py:2371-2416, solver_featherstone.py:691-757): incoming joint wrench at the body COM
    if (s_out->body_parent_f) {
        std::fill(s_out->body_parent_f, s_out->body_parent_f + 6 * B, 0.0f);
        if (m->articulation_count)
            for (int tid = 0; tid < B; ++tid) {
                spatial f_s = body_f_s[tid] + body_ft_s[tid] + lds(body_f.data(), tid);
                vec3 f_lin = f_s.top, f_ang_at_origin = f_s.bottom;
                vec3 r_com = body_q_com[tid].p - body_solve_origin[tid];
                sts(s_out->body_parent_f, tid, spatial(f_lin, f_ang_at_origin - cross(r_com, f_lin)));
            }
    }

    // J, M, P = M J, H = J^T P, L = chol(H + diag(armature)), solve (kernels.py:1422-1565,1655-1846); every
    // update_mass_matrix_interval-th step only (solver_featherstone.py:767), the factors are reused in between
    const bool rebuild = prm->mass_matrix_cache == nullptr || prm->update_mass_matrix != 0;
    size_t cache_off = 0;
    for (int a = 0; a < m->articulation_count; ++a) {
        int joint_start = m->articulation_start[a], joint_end = m->articulation_end[a];
        int joint_count = joint_end - joint_start;
        int dof_start = m->joint_qd_start[joint_start];
        int dof_stop = joint_end < J ? m->joint_qd_start[joint_end] : D;
        int n = dof_stop - dof_start, rows = 6 * joint_count;
        std::vector<float> Jm(size_t(rows) * n, 0.0f), M(size_t(rows) * rows, 0.0f), P(size_t(rows) * n), H(size_t(n) * n),
            L(size_t(n) * n, 0.0f);
        float* cached = prm->mass_matrix_cache ? prm->mass_matrix_cache + cache_off : nullptr;
        cache_off += size_t(n) * n;
        if (!rebuild) {
            for (size_t i = 0; i < size_t(n) * n; ++i) L[i] = cached[i];
        } else {
        for (int i = 0; i < joint_count; ++i) {
            int row_start = i * 6;
            int j = joint_start + i;
            while (j != -1) {
                int jds = m->joint_qd_start[j], jde = dof_end(m, j);
                for (int dof = 0; dof < jde - jds; ++dof) {
                    int col = (jds - dof_start) + dof;
                    const spatial& S = joint_S_s[jds + dof];
                    for (int k = 0; k < 6; ++k) Jm[size_t(row_start + k) * n + col] = sget(S, k);
                }
                j = m->joint_ancestor[j];
            }
        }
        // spatial_mass indexes body_I_s by JOINT index (joint_start + l): body l of the articulation (kernels.py:1466-1480)
        for (int l = 0; l < joint_count; ++l) {
            const mat66& I = body_I_s[joint_start + l];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) M[size_t(l * 6 + i) * rows + (l * 6 + j)] = I.a[i][j];
        }
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < n; ++j) {
                float sum = 0.0f;
                for (int k = 0; k < rows; ++k) sum += M[size_t(i) * rows + k] * Jm[size_t(k) * n + j];
                P[size_t(i) * n + j] = sum;
            }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                float sum = 0.0f;
                for (int k = 0; k < rows; ++k) sum += Jm[size_t(k) * n + i] * P[size_t(k) * n + j];
                H[size_t(i) * n + j] = sum;
            }
        if (a == 0 && g_probe_H) {
            for (int i = 0; i < n * n && i < g_probe_cap; ++i) g_probe_H[i] = H[i];
            g_probe_H = nullptr;
        }
        // joint_armature_effective: 1e10 on the dofs of joints whose child is kinematic (solver_featherstone.py:269-282)
        std::vector<float> R(m->joint_armature + dof_start, m->joint_armature + dof_start + n);
        for (int j = joint_start; j < joint_end; ++j)
            if (kinematic_joint(j))
                for (int d = m->joint_qd_start[j]; d < dof_end(m, j); ++d) R[d - dof_start] = 1.0e10f;
        for (int j = 0; j < n; ++j) {
            float s = H[size_t(j) * n + j] + R[j];
            for (int k = 0; k < j; ++k) {
                float r = L[size_t(j) * n + k];
                s -= r * r;
            }
            s = std::sqrt(s);
            float invS = 1.0f / s;
            L[size_t(j) * n + j] = s;
            for (int i = j + 1; i < n; ++i) {
                s = H[size_t(i) * n + j];
                for (int k = 0; k < j; ++k) s -= L[size_t(i) * n + k] * L[size_t(j) * n + k];
                L[size_t(i) * n + j] = s * invS;
            }
        }
        if (cached)
            for (size_t i = 0; i < size_t(n) * n; ++i) cached[i] = L[i];
        }  // rebuild
        const float* b = joint_tau.data() + dof_start;
        float* x = joint_qdd.data() + dof_start;
        for (int i = 0; i < n; ++i) {
            float s = b[i];
            for (int j = 0; j < i; ++j) s -= L[size_t(i) * n + j] * x[j];
            x[i] = s / L[size_t(i) * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            float s = x[i];
            for (int j = i + 1; j < n; ++j) s -= L[size_t(j) * n + i] * x[j];
            x[i] = s / L[size_t(i) * n + i];
        }
    }

    // zero_kinematic_joint_qdd (kernels.py:1932-1948)
    for (int j = 0; j < J; ++j)
        if (kinematic_joint(j))
            for (int d = m->joint_qd_start[j]; d < dof_end(m, j); ++d) joint_qdd[d] = 0.0f;

    // integrate_generalized_joints (kernels.py:1849-1893)
    for (int j = 0; j < J; ++j)
        jcalc_integrate(m, m->joint_parent[j], ldx(m->joint_X_c, j), ld3(m->body_com, m->joint_child[j]), m->joint_type[j],
                        s_in->joint_q, qd_internal_in.data(), joint_qdd.data(), m->joint_q_start[j], m->joint_qd_start[j],
                        m->joint_dof_dim[2 * j], m->joint_dof_dim[2 * j + 1], dt, s_out->joint_q, qd_internal_out.data());

    // copy_kinematic_joint_state (kernels.py:1951-1976): prescribed joint state passes through the solve
    for (int j = 0; j < J; ++j) {
        if (!kinematic_joint(j)) continue;
        int q_end = j + 1 < J ? m->joint_q_start[j + 1] : m->coord_count;
        for (int i = m->joint_q_start[j]; i < q_end; ++i) s_out->joint_q[i] = s_in->joint_q[i];
        for (int d = m->joint_qd_start[j]; d < dof_end(m, j); ++d) qd_internal_out[d] = qd_internal_in[d];
    }

    // eval_fk_with_velocity_conversion (kernels.py:1987-2150) -> state_out.body_q / body_qd
    float* out_q = s_out->body_q;
    float* out_qd = s_out->body_qd;
    auto fk_with_velocity_conversion = [&](int joint_start, int joint_end) {
        for (int i = joint_start; i < joint_end; ++i) {
            int parent = m->joint_parent[i], child = m->joint_child[i], type = m->joint_type[i];
            int q_start = m->joint_q_start[i], qd_start = m->joint_qd_start[i];
            int lin = m->joint_dof_dim[2 * i], ang = m->joint_dof_dim[2 * i + 1];
            const float* jq = s_out->joint_q;
            const float* jqd = qd_internal_out.data();
            transform X_j = jcalc_transform(m, type, qd_start, lin, ang, jq, q_start);
            spatial v_j;
            if (type == PRISMATIC) v_j = spatial(ld3(m->joint_axis, qd_start) * jqd[qd_start], vec3());
            if (type == REVOLUTE) v_j = spatial(vec3(), ld3(m->joint_axis, qd_start) * jqd[qd_start]);
            if (type == BALL) v_j = spatial(vec3(), vec3(jqd[qd_start], jqd[qd_start + 1], jqd[qd_start + 2]));
            if (type == FREE || type == DISTANCE)
                v_j = spatial(vec3(jqd[qd_start], jqd[qd_start + 1], jqd[qd_start + 2]),
                              vec3(jqd[qd_start + 3], jqd[qd_start + 4], jqd[qd_start + 5]));
            if (type == D6) {
                vec3 vel_v(0.0f), vel_w(0.0f);
                for (int k = 0; k < 3; ++k)
                    if (lin > k) vel_v += ld3(m->joint_axis, qd_start + k) * jqd[qd_start + k];
                if (ang == 1) vel_w = jqd[qd_start + lin] * ld3(m->joint_axis, qd_start + lin);
                if (ang >= 2) {
                    vec3 a0, a1, a2;
                    d6_multi_angular(m, ang, qd_start + lin, jq, q_start + lin, a0, a1, a2);
                    vel_w = a0 * jqd[qd_start + lin] + a1 * jqd[qd_start + lin + 1];
                    if (ang == 3) vel_w = vel_w + a2 * jqd[qd_start + lin + 2];
                }
                v_j = spatial(vel_v, vel_w);
            }
            transform X_wpj = ldx(m->joint_X_p, i);
            transform X_wp;
            if (parent >= 0) {
                X_wp = ldx(out_q, parent);
                X_wpj = X_wp * X_wpj;
            }
            transform X_wcj = X_wpj * X_j;
            transform X_wc = X_wcj * transform_inverse(ldx(m->joint_X_c, i));
            vec3 x_child_origin = X_wc.p;
            vec3 v_parent_origin, w_parent;
            if (parent >= 0) {
                spatial v_wp = lds(out_qd, parent);
                w_parent = v_wp.bottom;
                v_parent_origin = com_twist_to_point_velocity(v_wp, X_wp, ld3(m->body_com, parent), x_child_origin);
            }
            vec3 linear_joint_world = transform_vector(X_wpj, v_j.top);
            vec3 angular_joint_world = transform_vector(X_wpj, v_j.bottom);
            vec3 linear_joint_origin;
            if (type == FREE || type == DISTANCE) {
                spatial v_j_world = transform_twist(X_wpj, v_j);
                linear_joint_origin = velocity_at_point(v_j_world, x_child_origin);
                angular_joint_world = v_j_world.bottom;
            } else {
                vec3 child_origin_offset_world = x_child_origin - X_wcj.p;
                linear_joint_origin = linear_joint_world + cross(angular_joint_world, child_origin_offset_world);
            }
            spatial v_wc_origin(v_parent_origin + linear_joint_origin, w_parent + angular_joint_world);
            stx(out_q, child, X_wc);
            sts(out_qd, child, origin_twist_to_com_twist(v_wc_origin, X_wc, ld3(m->body_com, child)));
        }
    };
    for (int a = 0; a < m->articulation_count; ++a) fk_with_velocity_conversion(m->articulation_start[a], m->articulation_end[a]);

    // descendant FREE / DISTANCE joints (solver_featherstone.py:229-265,1006-1046): the child pose is re-integrated from its
    // world COM twist, the joint coordinates are rebuilt from the poses, and the rest of the articulation is refreshed
    auto descendant_free = [&](int j) {
        int t = m->joint_type[j];
        return (t == FREE || t == DISTANCE) && m->joint_parent[j] >= 0 && !kinematic_joint(j);
    };
    bool any_descendant = false;
    for (int j = 0; j < J; ++j) any_descendant = any_descendant || descendant_free(j);
    if (any_descendant) {
        // correct_free_distance_body_pose_from_world_twist (kernels.py:1897-1929) on the poses of the start-of-step FK
        for (int j = 0; j < J; ++j) {
            if (!descendant_free(j)) continue;
            int child = m->joint_child[j];
            transform X_wb = ldx(body_q_prev.data(), child);
            vec3 com = ld3(m->body_com, child);
            spatial qd_com_world = lds(out_qd, child);
            quat q = X_wb.q;
            vec3 x_com = transform_point(X_wb, com);
            vec3 v_com = qd_com_world.top, w = qd_com_world.bottom;
            quat drdt = quat(w.x, w.y, w.z, 0.0f) * q * 0.5f;
            quat q_new = normalize(q + drdt * dt);
            vec3 x_com_new = x_com + v_com * dt;
            vec3 x_origin_new = x_com_new - quat_rotate(q_new, com);
            stx(out_q, child, transform(x_origin_new, q_new));
        }
        // reconstruct_free_distance_joint_q_from_body_pose (kernels.py:978-1012)
        for (int j = 0; j < J; ++j) {
            if (!descendant_free(j)) continue;
            int parent = m->joint_parent[j], child = m->joint_child[j];
            transform X_wpj = ldx(m->joint_X_p, j);
            if (parent >= 0) X_wpj = ldx(out_q, parent) * X_wpj;
            transform X_wcj = ldx(out_q, child) * ldx(m->joint_X_c, j);
            vec3 x_err_c = quat_rotate_inv(X_wpj.q, X_wcj.p - X_wpj.p);
            quat q_pc = quat_inverse(X_wpj.q) * X_wcj.q;
            int q_start = m->joint_q_start[j];
            s_out->joint_q[q_start + 0] = x_err_c.x;
            s_out->joint_q[q_start + 1] = x_err_c.y;
            s_out->joint_q[q_start + 2] = x_err_c.z;
            s_out->joint_q[q_start + 3] = q_pc.x;
            s_out->joint_q[q_start + 4] = q_pc.y;
            s_out->joint_q[q_start + 5] = q_pc.z;
            s_out->joint_q[q_start + 6] = q_pc.w;
        }
        // eval_fk_with_velocity_conversion_from_joint_starts (kernels.py:2332-2368): from the first such joint of each articulation
        for (int a = 0; a < m->articulation_count; ++a)
            for (int j = m->articulation_start[a]; j < m->articulation_end[a]; ++j)
                if (descendant_free(j)) {
                    fk_with_velocity_conversion(j, m->articulation_end[a]);
                    break;
                }
    }

    // convert_free_distance_joint_qd_internal_to_public (kernels.py:1015-1066) on state_out.body_q
    for (int j = 0; j < J; ++j) {
        int qs = m->joint_qd_start[j], qe = dof_end(m, j), t = m->joint_type[j];
        if (t != FREE && t != DISTANCE) {
            for (int i = qs; i < qe; ++i) s_out->joint_qd[i] = qd_internal_out[i];
            continue;
        }
        vec3 r = free_joint_com_offset(m, j, out_q);
        vec3 v_int(qd_internal_out[qs], qd_internal_out[qs + 1], qd_internal_out[qs + 2]);
        vec3 omega(qd_internal_out[qs + 3], qd_internal_out[qs + 4], qd_internal_out[qs + 5]);
        vec3 v_com = v_int + cross(omega, r);
        s_out->joint_qd[qs + 0] = v_com.x; s_out->joint_qd[qs + 1] = v_com.y; s_out->joint_qd[qs + 2] = v_com.z;
        s_out->joint_qd[qs + 3] = omega.x; s_out->joint_qd[qs + 4] = omega.y; s_out->joint_qd[qs + 5] = omega.z;
    }
}
