// TEST INFRASTRUCTURE ONLY -- CPU oracle: forward kinematics.
// Literal restatement of eval_single_articulation_fk / eval_articulation_fk
//   newton/_src/sim/articulation.py:14-33,236-470 (PRISMATIC/REVOLUTE/BALL/FREE/DISTANCE/FIXED; D6 with up to three angular axes)
#include "oracle_common.h"
using namespace orc;

static vec3 com_twist_to_point_velocity(const spatial& qd, const transform& X_wb, vec3 com, vec3 point) {
    return velocity_at_point(qd, point - transform_point(X_wb, com));
}
static spatial origin_twist_to_com_twist(const spatial& qd, const transform& X_wb, vec3 com) {
    return spatial(velocity_at_point(qd, transform_vector(X_wb, com)), qd.bottom);
}
static spatial com_twist_to_origin_twist(const spatial& qd, const transform& X_wb, vec3 com) {
    return spatial(qd.top - cross(qd.bottom, transform_vector(X_wb, com)), qd.bottom);
}

extern "C" void o_eval_fk(const o_model* m, const float* joint_q, const float* joint_qd, float* body_q, float* body_qd) {
    for (int a = 0; a < m->articulation_count; ++a) {
        for (int i = m->articulation_start[a]; i < m->articulation_end[a]; ++i) {
            if (m->joint_articulation[i] == -1) continue;
            int parent = m->joint_parent[i], child = m->joint_child[i];
            int type = m->joint_type[i];
            if (type == ROD) continue;
            transform X_pj = ldx(m->joint_X_p, i), X_cj = ldx(m->joint_X_c, i);
            int q_start = m->joint_q_start[i], qd_start = m->joint_qd_start[i];
            int lin_axis_count = m->joint_dof_dim[2 * i], ang_axis_count = m->joint_dof_dim[2 * i + 1];
            transform X_j;
            spatial v_j;
            if (type == PRISMATIC) {
                vec3 axis = ld3(m->joint_axis, qd_start);
                X_j = transform(axis * joint_q[q_start], quat_identity());
                v_j = spatial(axis * joint_qd[qd_start], vec3());
            }
            if (type == REVOLUTE) {
                vec3 axis = ld3(m->joint_axis, qd_start);
                X_j = transform(vec3(), quat_from_axis_angle(axis, joint_q[q_start]));
                v_j = spatial(vec3(), axis * joint_qd[qd_start]);
            }
            if (type == BALL) {
                quat r(joint_q[q_start], joint_q[q_start + 1], joint_q[q_start + 2], joint_q[q_start + 3]);
                X_j = transform(vec3(), r);
                v_j = spatial(vec3(), vec3(joint_qd[qd_start], joint_qd[qd_start + 1], joint_qd[qd_start + 2]));
            }
            if (type == FREE || type == DISTANCE) {
                X_j = transform(vec3(joint_q[q_start], joint_q[q_start + 1], joint_q[q_start + 2]),
                                quat(joint_q[q_start + 3], joint_q[q_start + 4], joint_q[q_start + 5], joint_q[q_start + 6]));
                v_j = spatial(vec3(joint_qd[qd_start], joint_qd[qd_start + 1], joint_qd[qd_start + 2]),
                              vec3(joint_qd[qd_start + 3], joint_qd[qd_start + 4], joint_qd[qd_start + 5]));
            }
            if (type == D6) {
                vec3 pos(0.0f), vel_v(0.0f), vel_w(0.0f);
                quat rot = quat_identity();
                for (int k = 0; k < 3; ++k)
                    if (lin_axis_count > k) {
                        vec3 axis = ld3(m->joint_axis, qd_start + k);
                        pos += axis * joint_q[q_start + k];
                        vel_v += axis * joint_qd[qd_start + k];
                    }
                int iq = q_start + lin_axis_count, iqd = qd_start + lin_axis_count;
                if (ang_axis_count == 1) {
                    vec3 axis = ld3(m->joint_axis, iqd);
                    rot = quat_from_axis_angle(axis, joint_q[iq]);
                    vel_w = joint_qd[iqd] * axis;
                }
                if (ang_axis_count == 2) {  // compute_2d_rotational_dofs (articulation.py:36-83)
                    vec3 axis_0 = ld3(m->joint_axis, iqd), axis_1 = ld3(m->joint_axis, iqd + 1);
                    quat q_off = quat_from_matrix(matrix_from_cols(axis_0, axis_1, cross(axis_0, axis_1)));
                    vec3 local_0 = quat_rotate(q_off, vec3(1.0f, 0.0f, 0.0f)), local_1 = quat_rotate(q_off, vec3(0.0f, 1.0f, 0.0f));
                    vec3 a0 = local_0;
                    quat q_0 = quat_from_axis_angle(a0, joint_q[iq]);
                    vec3 a1 = quat_rotate(q_0, local_1);
                    quat q_1 = quat_from_axis_angle(a1, joint_q[iq + 1]);
                    rot = q_1 * q_0;
                    vel_w = a0 * joint_qd[iqd] + a1 * joint_qd[iqd + 1];
                }
                if (ang_axis_count == 3) {  // compute_3d_rotational_dofs (articulation.py:127-178)
                    vec3 axis_0 = ld3(m->joint_axis, iqd), axis_1 = ld3(m->joint_axis, iqd + 1), axis_2 = ld3(m->joint_axis, iqd + 2);
                    quat q_0 = quat_from_axis_angle(axis_0, joint_q[iq]);
                    vec3 axis_1_w = quat_rotate(q_0, axis_1);
                    quat q_1 = quat_from_axis_angle(axis_1_w, joint_q[iq + 1]);
                    vec3 axis_2_w = quat_rotate(q_1 * q_0, axis_2);
                    quat q_2 = quat_from_axis_angle(axis_2_w, joint_q[iq + 2]);
                    rot = q_2 * q_1 * q_0;
                    vel_w = axis_0 * joint_qd[iqd] + axis_1_w * joint_qd[iqd + 1] + axis_2_w * joint_qd[iqd + 2];
                }
                X_j = transform(pos, rot);
                v_j = spatial(vel_v, vel_w);
            }
            transform X_wpj = X_pj;
            transform X_wp;
            if (parent >= 0) {
                X_wp = ldx(body_q, parent);
                X_wpj = X_wp * X_wpj;
            }
            transform X_wcj = X_wpj * X_j;
            transform X_wc = X_wcj * transform_inverse(X_cj);

            vec3 x_child_origin = X_wc.p;
            vec3 v_parent_origin, w_parent;
            if (parent >= 0) {
                spatial v_wp = lds(body_qd, parent);
                w_parent = v_wp.bottom;
                v_parent_origin = com_twist_to_point_velocity(v_wp, X_wp, ld3(m->body_com, parent), x_child_origin);
            }
            vec3 linear_joint_world = transform_vector(X_wpj, v_j.top);
            vec3 angular_joint_world = transform_vector(X_wpj, v_j.bottom);
            vec3 linear_joint_origin;
            if (type == FREE || type == DISTANCE) {
                spatial v_joint_origin =
                    com_twist_to_origin_twist(spatial(linear_joint_world, angular_joint_world), X_wc, ld3(m->body_com, child));
                linear_joint_origin = v_joint_origin.top;
            } else {
                vec3 child_origin_offset_world = x_child_origin - X_wcj.p;
                linear_joint_origin = linear_joint_world + cross(angular_joint_world, child_origin_offset_world);
            }
            spatial v_wc_origin(v_parent_origin + linear_joint_origin, w_parent + angular_joint_world);
            stx(body_q, child, X_wc);
            sts(body_qd, child, origin_twist_to_com_twist(v_wc_origin, X_wc, ld3(m->body_com, child)));
        }
    }
}
