// nt_semi_implicit.hpp -- SolverSemiImplicit phases (penalty joints, penalty contacts) and its step kernel.
// Included by nt_kernels.hip inside its anonymous namespace, in this order: nt_layout.hpp, nt_collide.hpp, nt_xpbd.hpp,
// nt_semi_implicit.hpp, nt_featherstone.hpp (one translation unit; the split is for reading, not for separate compilation).
// No include guard: nt_kernels.hip includes the PHASES (NT_SI_PHASES_ONLY) once per arithmetic namespace -- SolverFeatherstone shares
// si_contact_item / si_dof_force -- and the kernel (NT_SI_KERNEL_ONLY) once, in namespace ieee.
#ifndef NT_SI_KERNEL_ONLY

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
            xform X_pj = c.plxf(c.L.jp, 0, nj, j), X_cj = c.plxf(c.L.jp, 7, nj, j);
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
                if (ang >= 2) {  // kernels_body.py:371-515: decompose the relative rotation, transport the axes like FK
                    int i_0 = lin + qd_start, i_0_q = lin + tq_start;
                    vec3 angles = quat_decompose(r_err);  // r_err = inverse(q_p) * q_c
                    vec3 orig_axis_0 = c.dof_axis(i_0), orig_axis_1 = c.dof_axis(i_0 + 1);
                    vec3 orig_axis_2 = ang == 3 ? c.dof_axis(i_0 + 2) : cross(orig_axis_0, orig_axis_1);
                    vec3 axis_0 = orig_axis_0;
                    quat q_0 = quat_from_axis_angle(axis_0, angles.x);
                    vec3 axis_1 = quat_rotate(q_0, orig_axis_1);
                    quat q_1 = quat_from_axis_angle(axis_1, angles.y);
                    vec3 axis_2 = quat_rotate(q_1 * q_0, orig_axis_2);
                    axis_0 = xform_vector(X_wp, axis_0);
                    axis_1 = xform_vector(X_wp, axis_1);
                    axis_2 = xform_vector(X_wp, axis_2);
                    t_total += axis_0 * (-c.l(c.L.cf, 0, 1, i_0) - si_dof_force(c, i_0, i_0_q, angles.x, dot(axis_0, w_err)));
                    t_total += axis_1 * (-c.l(c.L.cf, 0, 1, i_0 + 1) - si_dof_force(c, i_0 + 1, i_0_q + 1, angles.y, dot(axis_1, w_err)));
                    if (ang == 3)
                        t_total += axis_2 * (-c.l(c.L.cf, 0, 1, i_0 + 2) - si_dof_force(c, i_0 + 2, i_0_q + 2, angles.z, dot(axis_2, w_err)));
                    else  // last axis (fixed)
                        t_total += axis_2 * -si_joint_force(angles.z, dot(axis_2, w_err), 0.0f, 0.0f, ke_att, kd_att * ads, 0.0f, 0.0f,
                                                            0.0f, 0.0f, 0.0f);
                }
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
// FUSED: the collide phases of the SAME kernel just ran (SolverFeatherstone's rollout): a pair's live contacts fill its slots from the
// front and their number is in L.pm, the (type-sorted) shapes of a pair are static -- neither the liveness test nor the shape ids need the
// Contacts buffers in HBM (one dependent global round trip less per contact phase, and no id -> local shape search)
template <int EPB, bool FUSED = false>
NT_DI void si_contact_item(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp;
    const float* D = ct.data;
    float has_a = 0.0f, has_b = 0.0f, a_is_pair_a = 1.0f;
    vec3 f_total, tq_a, tq_b;
    int gid_a = -1, gid_b = -1, fused_a = -1, fused_b = -1;
    if constexpr (FUSED) {
        const int p = slot / cpp, k = slot - p * cpp;
        if (k < (int)c.l(c.L.pm, 0, m.np, p)) {
            fused_a = c.T.pair_a[p];
            fused_b = c.T.pair_b[p];
            if (c.T.shape_type[fused_a] > c.T.shape_type[fused_b]) { const int t_ = fused_a; fused_a = fused_b; fused_b = t_; }  // narrow_phase.py:525-528
            gid_a = 0; gid_b = 1;  // (live: any two distinct non-negative ids)
        }
    } else {
        const size_t gi = (size_t)slot * c.ES + c.env;
        gid_a = ct.shape0[gi]; gid_b = ct.shape1[gi];
    }
    if (gid_a != gid_b) {
        float ke = 0.0f, kd = 0.0f, kf = 0.0f, ka = 0.0f, mu = 0.0f;
        int mat_nonzero = 0, shape_a = -1, shape_b = -1, body_a = -1, body_b = -1;
        if (gid_a >= 0) {
            shape_a = FUSED ? fused_a : c.local_shape_id(gid_a);
            mat_nonzero += 1;
            ke += c.shape_f(shape_a, SP_KE); kd += c.shape_f(shape_a, SP_KD); kf += c.shape_f(shape_a, SP_KF);
            ka += c.shape_f(shape_a, SP_KA); mu += c.shape_f(shape_a, SP_MU);
            body_a = c.T.shape_body[shape_a];
        }
        if (gid_b >= 0) {
            shape_b = FUSED ? fused_b : c.local_shape_id(gid_b);
            mat_nonzero += 1;
            ke += c.shape_f(shape_b, SP_KE); kd += c.shape_f(shape_b, SP_KD); kf += c.shape_f(shape_b, SP_KF);
            ka += c.shape_f(shape_b, SP_KA); mu += c.shape_f(shape_b, SP_MU);
            body_b = c.T.shape_body[shape_b];
        }
        if (mat_nonzero > 0) {
            ke /= float(mat_nonzero); kd /= float(mat_nonzero); kf /= float(mat_nonzero);
            ka /= float(mat_nonzero); mu /= float(mat_nonzero);
        }
        if (ct.prop) {  // per-contact stiffness / damping / friction scale (kernels_contact.py:452-459), e.g. hydroelastic faces
            const float contact_ke = ct.prop[c.g(0, ncs, slot)], contact_kd = ct.prop[c.g(1, ncs, slot)],
                        contact_mu = ct.prop[c.g(2, ncs, slot)];
            ke = contact_ke > 0.0f ? contact_ke : ke;
            kd = contact_kd > 0.0f ? contact_kd : kd;
            mu = contact_mu > 0.0f ? mu * contact_mu : mu;
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
#endif  // !NT_SI_KERNEL_ONLY
#ifndef NT_SI_PHASES_ONLY
template <int EPB>
__global__ void __launch_bounds__(EPB <= 8 ? 256 : 512) semi_implicit_step_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    Ctx<EPB> c(a, lds, make_layout(a.m, false).rows_semi);
    c.L.bf = c.L.si_bf;  // this solver's force scratch: body_f_tmp | joint wrenches | contact wrenches
    load_state(c, a.s_in);
    load_params(c, true);
    __syncthreads();
    if (c.valid) {
        const nt_model& m = a.m;
        stage_rows(c, c.L.bf, a.s_in.body_f, 6, m.nb);
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
#endif
