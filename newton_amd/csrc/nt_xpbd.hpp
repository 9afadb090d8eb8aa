// nt_xpbd.hpp -- SolverXPBD phases (joint forces, integrate, contacts, apply, joints, restitution, optional reporting) and the
// step control flow do_xpbd_step.  The collide / step / rollout kernels are nt_xpbd_kernels.hpp.
// No include guard: nt_kernels.hip includes this file once per arithmetic namespace -- `ieee` (the shared integrator / force
// helpers SolverSemiImplicit and SolverFeatherstone use; literal operation order) and `fused` (what the XPBD kernels run:
// NT_XPBD_FAST_MATH + contraction, see the top of nt_kernels.hip).

// ------------------------------------------------------------------------------------------------
// XPBD: apply_joint_forces (xpbd/kernels.py:945-1075)
// ------------------------------------------------------------------------------------------------
// Division inside the XPBD projection phases (contacts / joints / apply / integrate).  Namespace ieee: IEEE division, the literal
// operation of the reference.  Namespace fused (NT_XPBD_FAST_MATH, what the XPBD kernels run): v_rcp_f32 + multiply and
// v_sqrt_f32 -- 1 ulp instead of correctly rounded, 2 instructions instead of ~11 per division.
#ifdef NT_XPBD_FAST_MATH
NT_DI float xrcp(float x) { return __builtin_amdgcn_rcpf(x); }
NT_DI float xdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
NT_DI float xrcp(float x) { return 1.0f / x; }
NT_DI float xdiv(float a, float b) { return a / b; }
#endif
#ifdef NT_XPBD_FAST_MATH
NT_DI float xsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32 (1 ulp) instead of the correctly rounded expansion
#else
NT_DI float xsqrt(float x) { return sqrtf(x); }
#endif
NT_DI float xlength(vec3 a) { return xsqrt(dot(a, a)); }
NT_DI float xlength(quat a) { return xsqrt(dot(a, a)); }
NT_DI vec3 xdiv(vec3 a, float s) {
#ifdef NT_XPBD_FAST_MATH
    const float r = __builtin_amdgcn_rcpf(s);
    return vec3(a.x * r, a.y * r, a.z * r);
#else
    return a / s;
#endif
}
NT_DI vec3 xnormalize(vec3 a) {
    float l = xlength(a);
    if (l > 0.0f) return xdiv(a, l);
    return vec3();
}
NT_DI quat xnormalize(quat q) {
    float l = xlength(q);
    if (l > 0.0f) return q * xrcp(l);
    return quat(0.f, 0.f, 0.f, 1.f);
}
template <int EPB>
NT_DI void joint_force_item(const Ctx<EPB>& c, const int j);
template <int EPB>
NT_DI void seed_body_forces(const Ctx<EPB>& c, bool forces_are_zero) {
    // body threads seed body_f_tmp with state_in.body_f (solver_xpbd.py:423: wp.clone)
    const int nb = c.a.m.nb;
    if (forces_are_zero) {  // (rows, not elements: the pad row is cleared with the rest)
        for (int r = c.slot; r < 7 * nb; r += c.nslot) c.lds[(c.L.bf.off + r) * Ctx<EPB>::N + c.e] = 0.0f;
    } else {
        stage_rows(c, c.L.bf, c.a.s_in.body_f, 6, nb);
    }
}
template <int EPB>
NT_DI void phase_joint_forces(const Ctx<EPB>& c, bool forces_are_zero) {
    if (!c.valid) return;
    seed_body_forces(c, forces_are_zero);
    for (int j = c.slot; j < c.a.m.nj; j += c.nslot) joint_force_item(c, j);
}
template <int EPB>
NT_DI void joint_force_item(const Ctx<EPB>& c, const int j) {
    const nt_model& m = c.a.m;
    const int nj = m.nj;
    {
        vec3 fp, tp, fc, tc;  // parent wrench (subtracted), child wrench (added)
        int type = c.T.joint_type[j];
        if (c.T.joint_enabled[j] && type != JT_FIXED && type != JT_ROD) {
            int id_c = c.T.joint_child[j], id_p = c.T.joint_parent[j];
            xform X_pj = c.plxf(c.L.jp, 0, nj, j);
            xform X_cj = c.plxf(c.L.jp, 7, nj, j);
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
    vec3 gravity = c.gravity();
    const float dt = c.a.dt;

    vec3 x0 = q.p;
    quat r0 = q.q;
    vec3 x_com = x0 + quat_rotate(r0, com);
    vec3 v1 = v0 + (f0 * inv_mass + gravity * nonzero(inv_mass)) * dt;
    vec3 x1 = x_com + v1 * dt;
    vec3 wb = quat_rotate_inv(r0, w0);
    vec3 tb = quat_rotate_inv(r0, t0) - cross(wb, inertia * wb);
    vec3 w1 = quat_rotate(r0, wb + inv_inertia * tb * dt);
    quat r1 = xnormalize(r0 + quat(w1, 0.0f) * r0 * 0.5f * dt);
    w1 *= 1.0f - c.a.angular_damping * dt;
    c.st_lxf(c.L.bq, nb, b, xform(x1 - quat_rotate(r1, com), r1));
    c.st_lv3(c.L.bqd, 0, nb, b, v1);
    c.st_lv3(c.L.bqd, 3, nb, b, w1);
    c.update_body_derived(b);
}
// ------------------------------------------------------------------------------------------------
// XPBD body phases on LINEAR and ANGULAR lanes.
// Inside a SolverXPBD step the linear state of a body is its world COM (rows 0..2 of the body-derived tile L.bd): integrate_bodies and
// apply_body_deltas both move x_com = p + R(q) com and only then subtract R(q1) com again (solver.py:100-118, xpbd/kernels.py:915-921),
// so between integrate and the step's last apply the COM is advanced directly (x_com += dp dt, no dependence on the new rotation) and
// the body origin p (L.bq rows 0..2) is rebuilt once, by the last apply: p = x_com - R(q1) com, followed by the same
// world_com = xform_point((p, q1), com) every kernel entry computes -- so the state a rollout carries from substep to substep is
// bit for bit the state the call-by-call loop reloads from HBM.  Consumers inside the step (contact / joint rows) read (world COM, q).
// With x_com as the state the linear half (forces / corrections -> v, x_com) and the angular half (-> omega, q, W = R I^-1 R^T) of a
// body are independent programs: when the workgroup has idle waves (13 bodies x 16 environments fill 4 of its 8) they run side by side
// on different waves -- slots [0, nb) angular, slots [S0, S0 + nb) linear, S0 on a wave boundary.  A wave issues one dependent VALU
// instruction per ~8 cycles while the SIMD can start one every 2 (profiles/r05a_valu_issue.jsonl), so two half-length programs on two
// waves take about half the time of one.  Same arithmetic per quantity in both forms: split and unsplit tiles agree bit for bit.
// ------------------------------------------------------------------------------------------------
// does an apply phase follow integrate_bodies in this step (else integrate rebuilds the body origins itself)
NT_DI bool xpbd_applies_follow(const KArgs& a) { return a.p.iterations > 0 && (a.has_contacts != 0 || a.m.nj > 0); }
template <int EPB>
NT_DI int body_lane_split(const Ctx<EPB>& c) {  // S0 if the linear lanes fit behind the angular ones, else 0 (one lane per body)
    const int spw = 64 / Ctx<EPB>::N > 0 ? 64 / Ctx<EPB>::N : 1;
    const int S0 = ((c.a.m.nb + spw - 1) / spw) * spw;
    return c.lane_split && S0 + c.a.m.nb <= c.nslot ? S0 : 0;
}
// integrate_bodies (solver.py:63-170) of SolverXPBD.  need_p: no apply phase follows in this step -- rebuild the body origin here
// (both halves on one lane).
template <int EPB>
NT_DI void xpbd_integrate_item(const Ctx<EPB>& c, const int b, const bool do_lin, const bool do_ang, const bool need_p) {
    const nt_model& m = c.a.m;
    const int nb = m.nb, nj = m.nj;
    vec3 f0, t0;
    if (do_lin) f0 = c.lv3(c.L.bf, 0, nb, b);
    if (do_ang) t0 = c.lv3(c.L.bf, 3, nb, b);
    for (int i = c.T.body_joint_start[b]; i < c.T.body_joint_start[b + 1]; ++i) {
        const int code = c.T.body_joint_list[i];
        const int j = code >> 1, row = (code & 1) ? 6 : 0;
        if (do_lin) {
            const vec3 f = c.lv3(c.L.jf, row, nj, j);
            if (code & 1) f0 += f;
            else f0 -= f;
        }
        if (do_ang) {
            const vec3 t = c.lv3(c.L.jf, row + 3, nj, j);
            if (code & 1) t0 += t;
            else t0 -= t;
        }
    }
    if (c.T.body_flags[b] & BODY_KINEMATIC) return;  // pass through unchanged
    const float dt = c.a.dt;
    vec3 x1;
    quat r1;
    if (do_lin) {
        const vec3 v0 = c.body_v(b);
        const float inv_mass = c.inv_mass(b);
        const vec3 v1 = v0 + (f0 * inv_mass + c.gravity() * nonzero(inv_mass)) * dt;
        x1 = c.world_com(b) + v1 * dt;
        c.st_lv3(c.L.bqd, 0, nb, b, v1);
        c.st_lv3(c.L.bd, 0, nb, b, x1);
    }
    if (do_ang) {
        const quat r0 = c.body_rot(b);
        const vec3 w0 = c.body_w(b);
        const mat33 inertia = c.inertia(b), inv_inertia = c.inv_inertia(b);
        const vec3 wb = quat_rotate_inv(r0, w0);
        const vec3 tb = quat_rotate_inv(r0, t0) - cross(wb, inertia * wb);
        vec3 w1 = quat_rotate(r0, wb + inv_inertia * tb * dt);
        r1 = xnormalize(r0 + quat(w1, 0.0f) * r0 * 0.5f * dt);
        w1 *= 1.0f - c.a.angular_damping * dt;
        c.l(c.L.bq, 3, nb, b) = r1.x; c.l(c.L.bq, 4, nb, b) = r1.y; c.l(c.L.bq, 5, nb, b) = r1.z; c.l(c.L.bq, 6, nb, b) = r1.w;
        c.st_lv3(c.L.bqd, 3, nb, b, w1);
        c.update_body_w(b, r1);
    }
    if (need_p) {
        const xform X(x1 - quat_rotate(r1, c.com(b)), r1);
        c.st_lv3(c.L.bq, 0, nb, b, X.p);
        c.update_world_com(b, X);
    }
}
template <int EPB>
NT_DI void phase_xpbd_integrate(const Ctx<EPB>& c, const bool need_p) {
    if (!c.valid) return;
    const int nb = c.a.m.nb, S0 = need_p ? 0 : body_lane_split(c);
    if (S0) {
        if (c.slot < nb) xpbd_integrate_item(c, c.slot, false, true, false);
        else if (c.slot >= S0 && c.slot < S0 + nb) xpbd_integrate_item(c, c.slot - S0, true, false, false);
    } else {
        for (int b = c.tslot; b < nb; b += c.nslot) xpbd_integrate_item(c, b, true, true, need_p);
    }
}
template <int EPB>
NT_DI void phase_body_derived(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int b = c.tslot; b < c.a.m.nb; b += c.nslot) c.update_body_derived(b);
}
template <int EPB, bool SEMI>
NT_DI void phase_integrate(const Ctx<EPB>& c) {
    if (!c.valid) return;
    for (int b = c.tslot; b < c.a.m.nb; b += c.nslot) integrate_item<EPB, SEMI>(c, b);
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
    delta_lambda = denom > 0.0f ? xdiv(delta_lambda, dt * denom) : delta_lambda;
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
    if (denom + alpha > 0.0f) delta_lambda = xdiv(delta_lambda, (dt + gamma) * denom + xdiv(alpha, dt));
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
    if (denom + alpha > 0.0f) delta_lambda = xdiv(delta_lambda, (dt + gamma) * denom + xdiv(alpha, dt));
    return delta_lambda;
}

// ------------------------------------------------------------------------------------------------
// where the per-contact correction records (CW_FLOATS rows per contact slot) live: in LDS (default), or -- for pair-heavy
// scenes whose records do not fit the CU's LDS (nt_model.contact_scratch_in_hbm) -- in nt_contacts.cw in HBM, one contiguous record
// per (environment, slot): that tile runs ONE environment per workgroup, so the env-major SoA of the other buffers would put every
// float of a record (and every slot of a pair) in its own cache line and DRAM page -- 35 lines per pair in the apply phase instead of 4
// ------------------------------------------------------------------------------------------------
// NC: record stride of the LDS copy (NC_CWX for the position solve, NC_CW for the restitution pass)
struct CwLds {
    template <int NC, int EPB>
    static NT_DI float& at(const Ctx<EPB>& c, int comp, int ncs, int slot) { return c.l(Fld<NC>{c.L.cw.off}, comp, ncs, slot); }
};
struct CwHbm {
    template <int NC, int EPB>
    static NT_DI float& at(const Ctx<EPB>& c, int comp, int ncs, int slot) {
        static_assert(NC <= CW_FLOATS, "nt_contacts.cw holds CW_FLOATS floats per (environment, slot)");
        return c.a.ct.cw[((size_t)c.env * ncs + slot) * NC + comp];
    }
};
template <class CW, int NC, int EPB>
NT_DI vec3 cw_v3(const Ctx<EPB>& c, int comp, int ncs, int slot) {
    return vec3(CW::template at<NC>(c, comp, ncs, slot), CW::template at<NC>(c, comp + 1, ncs, slot),
                CW::template at<NC>(c, comp + 2, ncs, slot));
}
template <class CW, int NC, int EPB>
NT_DI void cw_st3(const Ctx<EPB>& c, int comp, int ncs, int slot, vec3 v) {
    CW::template at<NC>(c, comp, ncs, slot) = v.x;
    CW::template at<NC>(c, comp + 1, ncs, slot) = v.y;
    CW::template at<NC>(c, comp + 2, ncs, slot) = v.z;
}

// ------------------------------------------------------------------------------------------------
// XPBD: solve_body_contact_positions (xpbd/kernels.py:2164-2399); one lane per contact slot.
// ------------------------------------------------------------------------------------------------
// The geometry of one contact as the position solve reads it: a fixed slot of the environment tile (env-major SoA in HBM) or a
// flat row of the SDF legs (Newton's AoS arrays).  Offsets are only fetched when friction is active, like the reference.
template <int EPB>
struct SlotRecord {
    const Ctx<EPB>& c;
    const float* D;
    int ncs, slot;
    NT_DI vec3 point0() const { return c.gv3(D, CD_POINT0, ncs, slot); }
    NT_DI vec3 point1() const { return c.gv3(D, CD_POINT1, ncs, slot); }
    NT_DI vec3 offset0() const { return c.gv3(D, CD_OFFSET0, ncs, slot); }
    NT_DI vec3 offset1() const { return c.gv3(D, CD_OFFSET1, ncs, slot); }
    NT_DI vec3 normal() const { return c.gv3(D, CD_NORMAL, ncs, slot); }
    NT_DI float margins() const { return D[c.g(CD_MARGIN0, ncs, slot)] + D[c.g(CD_MARGIN1, ncs, slot)]; }
};
template <int EPB>
struct LdsRecord {  // the slot's record in L.cr (LDS-record tiles of the fused rollout)
    const Ctx<EPB>& c;
    int ncs, slot;
    NT_DI vec3 point0() const { return c.lv3(c.L.cr, CD_POINT0, ncs, slot); }
    NT_DI vec3 point1() const { return c.lv3(c.L.cr, CD_POINT1, ncs, slot); }
    NT_DI vec3 offset0() const { return c.lv3(c.L.cr, CD_OFFSET0, ncs, slot); }
    NT_DI vec3 offset1() const { return c.lv3(c.L.cr, CD_OFFSET1, ncs, slot); }
    NT_DI vec3 normal() const { return c.lv3(c.L.cr, CD_NORMAL, ncs, slot); }
    NT_DI float margins() const { return c.l(c.L.cr, CD_MARGIN0, ncs, slot) + c.l(c.L.cr, CD_MARGIN1, ncs, slot); }
};
template <int EPB>
struct AosRecord {  // the slot's line of nt_contacts.cr (pair-heavy fused rollout)
    const float* o;
    NT_DI vec3 point0() const { return vec3(o[CD_POINT0], o[CD_POINT0 + 1], o[CD_POINT0 + 2]); }
    NT_DI vec3 point1() const { return vec3(o[CD_POINT1], o[CD_POINT1 + 1], o[CD_POINT1 + 2]); }
    NT_DI vec3 offset0() const { return vec3(o[CD_OFFSET0], o[CD_OFFSET0 + 1], o[CD_OFFSET0 + 2]); }
    NT_DI vec3 offset1() const { return vec3(o[CD_OFFSET1], o[CD_OFFSET1 + 1], o[CD_OFFSET1 + 2]); }
    NT_DI vec3 normal() const { return vec3(o[CD_NORMAL], o[CD_NORMAL + 1], o[CD_NORMAL + 2]); }
    NT_DI float margins() const { return o[CD_MARGIN0] + o[CD_MARGIN1]; }
};
struct FlatRecord {
    const nt_flat_rows& f;
    int r;
    static NT_DI vec3 ld(const float* p, int r) { return vec3(p[3 * (size_t)r], p[3 * (size_t)r + 1], p[3 * (size_t)r + 2]); }
    NT_DI vec3 point0() const { return ld(f.point0, r); }
    NT_DI vec3 point1() const { return ld(f.point1, r); }
    NT_DI vec3 offset0() const { return ld(f.offset0, r); }
    NT_DI vec3 offset1() const { return ld(f.offset1, r); }
    NT_DI vec3 normal() const { return ld(f.normal, r); }
    NT_DI float margins() const { return f.margin0[r] + f.margin1[r]; }
};

// solve_body_contact_positions (xpbd/kernels.py:2164-2399) for one live contact between body_a / body_b (-1: static); returns
// false when the contact is separated (no correction).  lin_delta_b is the exact negation of lin_delta_a.
// a contact record fetched as a whole: the seventeen loads of a record leave together (one memory round trip; fetched field by field
// where the solve first needs it -- the offsets inside the friction branch -- a slot record cost three dependent L2 round trips per
// contact phase on a wave that has nothing else to run)
struct LoadedRecord {
    vec3 p0, p1, o0, o1, n;
    float m;
    NT_DI vec3 point0() const { return p0; }
    NT_DI vec3 point1() const { return p1; }
    NT_DI vec3 offset0() const { return o0; }
    NT_DI vec3 offset1() const { return o1; }
    NT_DI vec3 normal() const { return n; }
    NT_DI float margins() const { return m; }
};
template <class REC>
NT_DI LoadedRecord load_record(const REC& r) {
    LoadedRecord L;
    L.p0 = r.point0(); L.p1 = r.point1(); L.n = r.normal(); L.o0 = r.offset0(); L.o1 = r.offset1(); L.m = r.margins();
    return L;
}
// Written load-first and branch-free: every LDS operand of both bodies and both shapes is fetched from clamped indices before
// the first use (a static side is selected to identity / zero afterwards: its rotation leaves a vector unchanged and its zero W tile
// makes every quadratic form vanish), the optional rows are computed unconditionally and their multiplier selected to zero.  A
// phase whose lanes own a SIMD alone cannot hide an LDS round trip (68 cycles, profiles/r05a_valu_issue.jsonl); the branchy form
// took ~45 dependent round trips per contact, this one four.
template <int EPB, class REC>
NT_DI bool contact_solve(const Ctx<EPB>& c, const REC& rec, int shape_a, int shape_b, int body_a, int body_b, vec3& lin_delta_a,
                         vec3& ang_delta_a, vec3& ang_delta_b, const bool full_a) {
    const float dt = c.a.dt, relaxation = c.a.p.rigid_contact_relaxation;
    const bool ha = body_a >= 0, hb = body_b >= 0, sha = shape_a >= 0, shb = shape_b >= 0;
    const int ba = ha ? body_a : 0, bb = hb ? body_b : 0, sa = sha ? shape_a : 0, sb = shb ? shape_b : 0;
    // a body inside the step is (world COM, rotation): the origin p is stale between integrate and the last apply (see the body
    // phases above).  A contact point p + R x becomes x_com + R (x - com); its lever arm about the COM is the rotated part alone.
    quat q_a = c.body_rot(ba), q_b = c.body_rot(bb);
    vec3 wc_a = c.world_com(ba), wc_b = c.world_com(bb), com_a = c.com(ba), com_b = c.com(bb);
    vec3 omega_a = c.body_w(ba), omega_b = c.body_w(bb), vel_a = c.body_v(ba), vel_b = c.body_v(bb);
    float m_inv_a = c.inv_mass(ba), m_inv_b = c.inv_mass(bb);
    typename Ctx<EPB>::Wsym W_a = c.w_tile(ba), W_b = c.w_tile(bb);
    bool kin_a = (c.T.body_flags[ba] & BODY_KINEMATIC) != 0, kin_b = (c.T.body_flags[bb] & BODY_KINEMATIC) != 0;
    float mu_a = c.shape_f_sel(sa, SP_MU), mu_b = c.shape_f_sel(sb, SP_MU);
    float mut_a = c.shape_f_sel(sa, SP_MU_TORSIONAL), mut_b = c.shape_f_sel(sb, SP_MU_TORSIONAL);
    float mur_a = c.shape_f_sel(sa, SP_MU_ROLLING), mur_b = c.shape_f_sel(sb, SP_MU_ROLLING);
    const vec3 point0 = rec.point0(), point1 = rec.point1(), n = rec.normal();
    const vec3 offset_a = rec.offset0(), offset_b = rec.offset1();
    const float margins = rec.margins();
    if (!ha) {  // static side: identity pose at the origin, no mass, no velocity
        q_a = quat(0.0f, 0.0f, 0.0f, 1.0f); wc_a = vec3(0.0f); com_a = vec3(0.0f); omega_a = vec3(0.0f); vel_a = vec3(0.0f);
        m_inv_a = 0.0f; W_a = typename Ctx<EPB>::Wsym{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; kin_a = false;
    }
    if (!hb) {
        q_b = quat(0.0f, 0.0f, 0.0f, 1.0f); wc_b = vec3(0.0f); com_b = vec3(0.0f); omega_b = vec3(0.0f); vel_b = vec3(0.0f);
        m_inv_b = 0.0f; W_b = typename Ctx<EPB>::Wsym{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}; kin_b = false;
    }
    if (!sha) { mu_a = 0.0f; mut_a = 0.0f; mur_a = 0.0f; }
    if (!shb) { mu_b = 0.0f; mut_b = 0.0f; mur_b = 0.0f; }
    float mu = mu_a + mu_b, mu_torsional = mut_a + mut_b, mu_rolling = mur_a + mur_b;
    if (sha && shb) {  // the mean of one or two shapes' coefficients: x / 2 == x * 0.5 exactly, x / 1 == x
        mu *= 0.5f;
        mu_torsional *= 0.5f;
        mu_rolling *= 0.5f;
    }
    // A static side (ground, fixed geometry: most contacts of a walking scene) has the identity pose, a zero W tile and zero velocities:
    // its contact point is the record's point, its quadratic forms and velocity terms vanish and its corrections are never applied
    // (has_a / has_b clear) -- the arithmetic below is skipped for it, by a branch the wave takes only if one of its contacts has a
    // moving body on that side (operands stay fetched up front: the loads above leave as one batch either way).  full_a: the
    // caller reports Contacts.force, which carries (lin_delta_a, ang_delta_a) of a static shape0 too (xpbd/kernels.py:2394-2395).
    const bool ca = ha || full_a;
    vec3 r_a, r_b, angular_a, angular_b;
    vec3 bx_a = point0, bx_b = point1;
    float wq_a = 0.0f, wq_b = 0.0f;
    if (ca) {
        r_a = quat_rotate(q_a, point0 - com_a);
        bx_a = wc_a + r_a;
    }
    if (hb) {
        r_b = quat_rotate(q_b, point1 - com_b);
        bx_b = wc_b + r_b;
    }
    float d = dot(n, bx_b - bx_a) - margins;
    if (!(d < 0.0f)) return false;
    vec3 lin_delta_b;
    if (ca) {
        angular_a = -cross(r_a, n);
        wq_a = W_a.quad(angular_a);
    }
    if (hb) {
        angular_b = cross(r_b, n);
        wq_b = W_b.quad(angular_b);
    }

    float lambda_n = contact_constraint_delta(d, m_inv_a, m_inv_b, -n, n, wq_a, wq_b, relaxation, dt);
    lin_delta_a = -n * lambda_n;
    lin_delta_b = n * lambda_n;
    ang_delta_a = angular_a * lambda_n;
    ang_delta_b = angular_b * lambda_n;

    {  // friction row (applies when mu > 0 and the tangential error is non-zero)
        bx_a = point0 + offset_a;
        bx_b = point1 + offset_b;
        if (ca) {
            r_a = quat_rotate(q_a, bx_a - com_a);
            bx_a = wc_a + r_a;
        }
        if (hb) {
            r_b = quat_rotate(q_b, bx_b - com_b);
            bx_b = wc_b + r_b;
        }
        vec3 delta = bx_b - bx_a;
        vec3 friction_delta = delta - dot(n, delta) * n;
        if (kin_a || kin_b) {  // a kinematic body drags the contact along: its tangential velocity enters the error
            vec3 rel_v_kin_t(0.0f);
            vec3 v_a = velocity_at_point(spatial(vel_a, omega_a), r_a);
            vec3 t_a = v_a - dot(n, v_a) * n;
            if (kin_a) rel_v_kin_t = rel_v_kin_t - t_a;
            vec3 v_b = velocity_at_point(spatial(vel_b, omega_b), r_b);
            vec3 t_b = v_b - dot(n, v_b) * n;
            if (kin_b) rel_v_kin_t = rel_v_kin_t + t_b;
            friction_delta += rel_v_kin_t * dt;
        }
        const float err = xlength(friction_delta);
        const vec3 perp = vsel(err > 0.0f, xdiv(friction_delta, err), vec3());  // xnormalize, as a select
        wq_a = 0.0f; wq_b = 0.0f;
        angular_a = vec3(); angular_b = vec3();
        if (ca) {
            angular_a = -cross(r_a, perp);
            wq_a = W_a.quad(angular_a);
        }
        if (hb) {
            angular_b = cross(r_b, perp);
            wq_b = W_b.quad(angular_b);
        }
        float lambda_fr = contact_constraint_delta(err, m_inv_a, m_inv_b, -perp, perp, wq_a, wq_b, relaxation, dt);
        lambda_fr = fmaxw(lambda_fr, -lambda_n * mu);
        if (!(mu > 0.0f && err > 0.0f)) lambda_fr = 0.0f;
        lin_delta_a -= perp * lambda_fr;
        lin_delta_b += perp * lambda_fr;
        ang_delta_a += angular_a * lambda_fr;
        ang_delta_b += angular_b * lambda_fr;
    }
    vec3 delta_omega = omega_b - omega_a;
    const vec3 lin0(0.0f);
    {  // torsional friction about the normal (v^T W v is even in v: one quadratic form per body serves -n and n)
        float err = dot(delta_omega, n) * dt;
        wq_a = ca ? W_a.quad(n) : 0.0f;
        wq_b = hb ? W_b.quad(n) : 0.0f;
        float lt = contact_constraint_delta(err, m_inv_a, m_inv_b, lin0, lin0, wq_a, wq_b, relaxation, dt);
        lt = clampf(lt, -lambda_n * mu_torsional, lambda_n * mu_torsional);
        if (!(mu_torsional > 0.0f && fabsf(err) > 0.0f)) lt = 0.0f;
        ang_delta_a -= n * lt;
        ang_delta_b += n * lt;
    }
    {  // rolling friction about the tangential relative spin
        delta_omega -= dot(n, delta_omega) * n;
        const float len = xlength(delta_omega);
        const float err = len * dt;
        const vec3 roll_n = vsel(len > 0.0f, xdiv(delta_omega, len), vec3());  // xnormalize, as a select
        wq_a = ca ? W_a.quad(roll_n) : 0.0f;
        wq_b = hb ? W_b.quad(roll_n) : 0.0f;
        float lr = contact_constraint_delta(err, m_inv_a, m_inv_b, lin0, lin0, wq_a, wq_b, relaxation, dt);
        lr = fmaxw(lr, -lambda_n * mu_rolling);
        if (!(mu_rolling > 0.0f && err > 0.0f)) lr = 0.0f;
        ang_delta_a -= roll_n * lr;
        ang_delta_b += roll_n * lr;
    }
    return true;
}

// FUSED: the collide phase of the same kernel left the live-contact count of every pair in LDS, and the (type-sorted)
// shape order of a pair is static, so neither the liveness test nor the shape ids need the global contact arrays.
// live_pair >= 0 (fused, staged tiles): the slot is live contact (live_pair, slot - live_pair * cpp) straight from the compacted
// list, its shapes and bodies come from the pair descriptor -- one LDS round instead of the prefix search and the
// pair -> shape -> type / body chain.
template <int EPB, bool FUSED, class CW = CwLds>
NT_DI void contact_item(const Ctx<EPB>& c, const int slot, const int live_pair = -1) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp;
    float has_a = 0.0f, has_b = 0.0f, a_is_pair_a = 1.0f;
    vec3 lin_delta_a, ang_delta_a, ang_delta_b;

    bool live;
    int shape_a = -1, shape_b = -1, body_a = -1, body_b = -1;
    bool swapped = false, described = false;
    LoadedRecord rec;
    if (FUSED && live_pair >= 0) {
        if (c.lds_records) rec = load_record(LdsRecord<EPB>{c, ncs, slot});
        else rec = load_record(SlotRecord<EPB>{c, ct.data, ncs, slot});  // (a listed contact is live: fetch before anything depends on LDS)
        const int* d = c.T.pair_desc + 4 * live_pair;
        shape_a = d[0]; shape_b = d[1]; body_a = d[2];
        const int w = d[3];
        swapped = (w & (1 << 30)) != 0;
        body_b = (w << 2) >> 2;  // sign-extend the 30-bit body index
        live = body_a != body_b;
        described = true;
    } else if (FUSED) {
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
    if (live && !described) {
        body_a = shape_a >= 0 ? c.T.shape_body[shape_a] : -1;
        body_b = shape_b >= 0 ? c.T.shape_body[shape_b] : -1;
        live = body_a != body_b;
    }
    if (live && !described) {
        if (FUSED && c.big && c.aos_records) rec = load_record(AosRecord<EPB>{ct.cr + ((size_t)c.env * ncs + slot) * NT_CR_STRIDE});
        else rec = load_record(SlotRecord<EPB>{c, ct.data, ncs, slot});
    }
    if (live && contact_solve(c, rec, shape_a, shape_b, body_a, body_b, lin_delta_a, ang_delta_a, ang_delta_b,
                              !FUSED && c.a.rep.contact_impulse != nullptr)) {
        has_a = body_a >= 0 ? 1.0f : 0.0f;
        has_b = body_b >= 0 ? 1.0f : 0.0f;
        if (described) a_is_pair_a = swapped ? 0.0f : 1.0f;
        else a_is_pair_a = (shape_a == c.T.pair_a[slot / cpp]) ? 1.0f : 0.0f;
    }
    // lin_delta_b == -lin_delta_a bit for bit (IEEE negation commutes with every rounding above), so only one is stored
    cw_st3<CW, NC_CWX>(c, 0, ncs, slot, lin_delta_a);
    cw_st3<CW, NC_CWX>(c, CWX_ANG_A, ncs, slot, ang_delta_a);
    cw_st3<CW, NC_CWX>(c, CWX_ANG_B, ncs, slot, ang_delta_b);
    CW::template at<NC_CWX>(c, CWX_FLAGS, ncs, slot) = has_a + 2.0f * has_b + 4.0f * a_is_pair_a;
}
// The same solve for a row of the SDF legs (nt_contacts.flat): the record goes to flat.cw[row][10] (lin_a, ang_a, ang_b,
// flags: bit 0 the row corrects shape0's body, bit 1 shape1's body).  The launch-by-launch step kernels only.
template <int EPB>
NT_DI void flat_contact_item(const Ctx<EPB>& c, const int r) {
    const nt_flat_rows& f = c.a.ct.flat;
    float flags = 0.0f;
    vec3 lin_delta_a, ang_delta_a, ang_delta_b;
    const int gid_a = f.shape0[r], gid_b = f.shape1[r];
    if (gid_a != gid_b) {
        const int shape_a = gid_a >= 0 ? c.local_shape_id(gid_a) : -1, shape_b = gid_b >= 0 ? c.local_shape_id(gid_b) : -1;
        const int body_a = shape_a >= 0 ? c.T.shape_body[shape_a] : -1, body_b = shape_b >= 0 ? c.T.shape_body[shape_b] : -1;
        if (body_a != body_b &&
            contact_solve(c, load_record(FlatRecord{f, r}), shape_a, shape_b, body_a, body_b, lin_delta_a, ang_delta_a, ang_delta_b,
                          c.a.rep.contact_impulse != nullptr))
            flags = (body_a >= 0 ? 1.0f : 0.0f) + (body_b >= 0 ? 2.0f : 0.0f);
    }
    float* o = f.cw + CWX_FLOATS * (size_t)r;
    o[0] = lin_delta_a.x; o[1] = lin_delta_a.y; o[2] = lin_delta_a.z;
    o[CWX_ANG_A] = ang_delta_a.x; o[CWX_ANG_A + 1] = ang_delta_a.y; o[CWX_ANG_A + 2] = ang_delta_a.z;
    o[CWX_ANG_B] = ang_delta_b.x; o[CWX_ANG_B + 1] = ang_delta_b.y; o[CWX_ANG_B + 2] = ang_delta_b.z;
    o[CWX_FLAGS] = flags;
}
// flags of an XPBD correction record: does it touch the body on `side` of its pair (0: owner of pair_a's shape), and as
// the contact's shape0 ("a") or shape1?
struct CwxSide { bool has, is_a; };
template <class CW, int EPB>
NT_DI CwxSide cwx_side(const Ctx<EPB>& c, int ncs, int slot, int side) {
    const int f = (int)CW::template at<NC_CWX>(c, CWX_FLAGS, ncs, slot);
    CwxSide r;
    r.is_a = (side == 0) == ((f & 4) != 0);  // this body is the contact's "a" iff (side == 0) == (shape0 is pair_a's shape)
    r.has = (f & (r.is_a ? 1 : 2)) != 0;
    return r;
}
// ROWS: the Contacts may carry rows of the SDF legs (nt_contacts.flat) behind the slots -- the launch-by-launch step kernels; the fused
// rollouts never do and compile without that code
template <int EPB, bool FUSED, class CW = CwLds, bool ROWS = !FUSED>
NT_DI void phase_contacts(const Ctx<EPB>& c) {
    if (!c.valid) return;
    const int np = c.a.m.np, cpp = c.a.m.cpp;
    if constexpr (FUSED) {
        // lane i takes the environment's i-th live contact (L.px: exclusive prefix of the per-pair live counts); records
        // of dead slots are never written and never read (apply_item stops at the pair's live count)
        const int total = (int)c.l(c.L.px, 0, 1, np);
        if (c.L.has_lt) {  // the compacted list written with the prefix: no search
            for (int i = c.tslot; i < total; i += c.nslot) {
                const int e = *reinterpret_cast<const int*>(&c.l(c.L.lt, 0, 1, i));
                contact_item<EPB, FUSED, CW>(c, (e >> 4) * cpp + (e & 15), e >> 4);
            }
        } else {
            for (int i = c.tslot; i < total; i += c.nslot) {
                int lo = 0, hi = np;  // the last pair whose prefix is <= i
                while (hi - lo > 1) {
                    int mid = (lo + hi) >> 1;
                    if ((int)c.l(c.L.px, 0, 1, mid) <= i) lo = mid;
                    else hi = mid;
                }
                contact_item<EPB, FUSED, CW>(c, lo * cpp + (i - (int)c.l(c.L.px, 0, 1, lo)));
            }
        }
    } else {
        for (int s = c.slot; s < np * cpp; s += c.nslot) contact_item<EPB, FUSED, CW>(c, s);
    }
    if constexpr (ROWS) {
        if (const nt_flat_rows& f = c.a.ct.flat; f.row_start)  // rows of the SDF legs, appended after the slots like the
            for (int r = f.row_start[c.env] + c.slot; r < f.row_start[c.env + 1]; r += c.nslot)  // reference's later launches
                flat_contact_item(c, r);
    }
}

// ------------------------------------------------------------------------------------------------
// XPBD: apply_body_deltas (xpbd/kernels.py:864-933).  FROM_CONTACTS: sum contact corrections (+ contact counts)
// in ascending contact order; otherwise sum joint corrections in ascending joint order.
// ------------------------------------------------------------------------------------------------
// do_lin / do_ang: the halves this lane runs (both on tiles without idle waves).  last: the step's last apply -- the angular lane
// (it holds q1) also rebuilds the body origin p and the entry-form world COM, the linear lane leaves the COM alone.
constexpr int NT_APPLY_SLOTS = 5;   // contact slots of a pair fetched together (cpp <= 5)
constexpr int NT_APPLY_JOINTS = 4;  // incident joints fetched together; longer lists finish in a loop
template <int EPB, bool FROM_CONTACTS, class CW = CwLds, bool FUSED = false, bool ROWS = !FUSED>
NT_DI void apply_item(const Ctx<EPB>& c, const int b, const bool do_lin, const bool do_ang, const bool last) {
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    const bool need_p = last && do_ang;
    const bool want_lin = do_lin || need_p;
    float inv_m = c.inv_mass(b);
    if (inv_m == 0.0f) {  // pass-through; an immovable body that integrate_bodies still moved (v != 0) gets its origin back
        if (need_p && !(c.T.body_flags[b] & BODY_KINEMATIC)) {
            const quat q = c.body_rot(b);
            const xform X(c.world_com(b) - quat_rotate(q, c.com(b)), q);
            c.st_lv3(c.L.bq, 0, nb, b, X.p);
            c.update_world_com(b, X);
        }
        return;
    }

    vec3 dlin, dang;
    float inv_weight = 0.0f;
    if (FROM_CONTACTS) {
        const int cpp = m.cpp, ncs = m.np * cpp;
        for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
            int code = c.T.body_pair_list[i];
            int p = code >> 1, side = code & 1;  // side 0: this body owns pair_a's shape
            const int live = FUSED ? (int)c.l(c.L.pm, 0, m.np, p) : cpp;  // fused: only live slots carry a record
            // the pair's slots together: one LDS round for the flags, one for the records (a load / wait / branch per slot cost the
            // feet of the standing quadruped eight dependent rounds per apply); summed in ascending slot order as before
            // (named records, not arrays: the optimiser leaves conditionally written struct arrays in scratch memory; loads
            // unconditional from a clamped slot and predicated at the add: branches would split the batch into one round trip each)
            struct Slot { vec3 lin, ang; bool has, isa; };
            auto slot_of = [&](int k) { return p * cpp + (k < cpp ? k : cpp - 1); };
            auto fetch_flags = [&](int k) { return (int)CW::template at<NC_CWX>(c, CWX_FLAGS, ncs, slot_of(k)); };
            auto fetch = [&](int k, int fl) {
                Slot s;
                if (k >= live) fl = 0;
                s.isa = (side == 0) == ((fl & 4) != 0);  // this body is the contact's "a" iff (side == 0) == (shape0 is pair_a's shape)
                s.has = (fl & (s.isa ? 1 : 2)) != 0;
                if (want_lin) s.lin = cw_v3<CW, NC_CWX>(c, 0, ncs, slot_of(k));
                if (do_ang) s.ang = cw_v3<CW, NC_CWX>(c, s.isa ? CWX_ANG_A : CWX_ANG_B, ncs, slot_of(k));
                return s;
            };
            auto add = [&](const Slot& s) {
                if (s.has) {
                    if (want_lin) dlin += s.isa ? s.lin : -s.lin;
                    if (do_ang) dang += s.ang;
                    inv_weight += 1.0f;
                }
            };
            static_assert(NT_APPLY_SLOTS == 5, "five named slots below");
            if (live > 0) {
                const int f0 = fetch_flags(0), f1 = fetch_flags(1), f2 = fetch_flags(2), f3 = fetch_flags(3), f4 = fetch_flags(4);
                const Slot s0 = fetch(0, f0), s1 = fetch(1, f1), s2 = fetch(2, f2), s3 = fetch(3, f3), s4 = fetch(4, f4);
                add(s0); add(s1); add(s2); add(s3); add(s4);
            }
        }
        if constexpr (ROWS) {
            if (const nt_flat_rows& f = c.a.ct.flat; f.row_start) {  // then the SDF legs' rows, ascending row order
                const int* bs = f.body_blk_start + (size_t)c.env * (nb + 1) + b;
                for (int i = bs[0]; i < bs[1]; ++i) {
                    const int code = f.body_blk_list[2 * (size_t)i], count = f.body_blk_list[2 * (size_t)i + 1];
                    const int r0 = code >> 1, side = code & 1;
                    for (int r = r0; r < r0 + count; ++r) {
                        const float* w = f.cw + CWX_FLOATS * (size_t)r;
                        if (((int)w[CWX_FLAGS] & (side ? 2 : 1)) == 0) continue;
                        const vec3 lin(w[0], w[1], w[2]);
                        const float* a = w + (side ? CWX_ANG_B : CWX_ANG_A);
                        dlin += side ? -lin : lin;
                        dang += vec3(a[0], a[1], a[2]);
                        inv_weight += 1.0f;
                    }
                }
            }
        }
    } else {
        // ascending joint order: the body's entries of L.ji are contiguous (joint lanes write them in incidence order, the
        // angular-row term already signed), so they are summed from ONE base address with immediate offsets -- the first
        // NT_APPLY_JOINTS as one batch of loads (clamped entry, predicated add), longer lists in a loop
        const int nj2 = 2 * m.nj;
        const int i0 = c.T.body_joint_start[b], n = c.T.body_joint_start[b + 1] - i0;
        struct Inc { vec3 lin, ang, t; };
        auto fetch = [&](int k) {
            Inc r;
            const int i = i0 + k;  // (entries past the body's own are loaded and ignored: rows of the same scratch block)
            if (want_lin) r.lin = c.lv3(c.L.ji, 0, nj2, i);
            if (do_ang) {
                r.ang = c.lv3(c.L.ji, 3, nj2, i);
                r.t = c.lv3(c.L.ji, 6, nj2, i);
            }
            return r;
        };
        auto add = [&](int k, const Inc& r) {
            if (k < n) {
                if (want_lin) dlin += r.lin;
                if (do_ang) dang += r.ang + r.t;
            }
        };
        static_assert(NT_APPLY_JOINTS == 4, "four named entries below");
        if (n > 0) {
            const Inc r0 = fetch(0), r1 = fetch(1), r2 = fetch(2), r3 = fetch(3);
            add(0, r0); add(1, r1); add(2, r2); add(3, r3);
        }
        for (int i = i0 + NT_APPLY_JOINTS; i < i0 + n; ++i) {
            if (want_lin) dlin += c.lv3(c.L.ji, 0, nj2, i);
            if (do_ang) dang += c.lv3(c.L.ji, 3, nj2, i) + c.lv3(c.L.ji, 6, nj2, i);
        }
    }
    const float dt = c.a.dt;
    float weight = 1.0f;
    if (FROM_CONTACTS && c.a.p.rigid_contact_con_weighting) {
        if (inv_weight > 0.0f) weight = xrcp(inv_weight);
    }
    vec3 dp;
    if (want_lin) dp = dlin * (inv_m * weight);
    quat q1;
    if (do_ang) {
        const mat33 inv_I = c.inv_inertia(b), body_I = c.inertia(b);
        const quat q0 = c.body_rot(b);
        const vec3 w0 = c.body_w(b);
        vec3 dq = dang * weight;
        vec3 wb = quat_rotate_inv(q0, w0);
        vec3 dwb = inv_I * quat_rotate_inv(q0, dq);
        vec3 tb = cross(dwb, body_I * (wb + dwb)) + cross(wb, body_I * dwb);
        vec3 dw1 = quat_rotate(q0, dwb - (dt * inv_I) * tb);
        q1 = q0 + 0.5f * quat(dw1 * dt, 0.0f) * q0;
        q1 = xnormalize(q1);
        vec3 w1 = w0 + dw1;
        if (xlength(w1) < 1e-4f) w1 = vec3(0.0f);
        c.l(c.L.bq, 3, nb, b) = q1.x; c.l(c.L.bq, 4, nb, b) = q1.y; c.l(c.L.bq, 5, nb, b) = q1.z; c.l(c.L.bq, 6, nb, b) = q1.w;
        c.st_lv3(c.L.bqd, 3, nb, b, w1);
        c.update_body_w(b, q1);
    }
    if (do_lin) {
        vec3 v1 = c.body_v(b) + dp;
        if (xlength(v1) < 1e-4f) v1 = vec3(0.0f);
        c.st_lv3(c.L.bqd, 0, nb, b, v1);
        if (!last) c.st_lv3(c.L.bd, 0, nb, b, c.world_com(b) + dp * dt);
    }
    if (need_p) {
        const xform X((c.world_com(b) + dp * dt) - quat_rotate(q1, c.com(b)), q1);
        c.st_lv3(c.L.bq, 0, nb, b, X.p);
        c.update_world_com(b, X);
    }
}
template <int EPB, bool FROM_CONTACTS, class CW = CwLds, bool FUSED = false, bool ROWS = !FUSED>
NT_DI void phase_apply(const Ctx<EPB>& c, const bool last) {
    if (!c.valid) return;
    const int nb = c.a.m.nb, S0 = body_lane_split(c);
    if (S0) {
        if (c.slot < nb) apply_item<EPB, FROM_CONTACTS, CW, FUSED, ROWS>(c, c.slot, false, true, last);
        else if (c.slot >= S0 && c.slot < S0 + nb) apply_item<EPB, FROM_CONTACTS, CW, FUSED, ROWS>(c, c.slot - S0, true, false, last);
    } else {
        for (int b = c.tslot; b < nb; b += c.nslot) apply_item<EPB, FROM_CONTACTS, CW, FUSED, ROWS>(c, b, true, true, last);
    }
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
    if (ke_w.x > 0.0f) tp.x = xdiv(tp.x, ke_w.x);
    if (ke_w.y > 0.0f) tp.y = xdiv(tp.y, ke_w.y);
    if (ke_w.z > 0.0f) tp.z = xdiv(tp.z, ke_w.z);
    if (kd_w.x > 0.0f) tv.x = xdiv(tv.x, kd_w.x);
    if (kd_w.y > 0.0f) tv.y = xdiv(tv.y, kd_w.y);
    if (kd_w.z > 0.0f) tv.z = xdiv(tv.z, kd_w.z);
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
        xform X_pj = c.plxf(c.L.jp, 0, nj, j);
        xform X_cj = c.plxf(c.L.jp, 7, nj, j);
        xform X_wp = X_pj;
        vec3 world_com_p = X_pj.p;  // transform_point(pose_p = X_pj, com_p = 0) for world-attached joints
        vec3 vel_p(0.0f), omega_p(0.0f);
        // a body inside the step is (world COM, rotation), its origin is stale (see the body phases): the joint frame
        // pose * X_j = (p + R x_j, q q_j) is built as (x_com + R (x_j - com), q q_j)
        if (id_p >= 0) {
            const quat qb = c.body_rot(id_p);
            world_com_p = c.world_com(id_p);
            X_wp = xform(world_com_p + quat_rotate(qb, X_pj.p - c.com(id_p)), qb * X_pj.q);
            vel_p = c.body_v(id_p);
            omega_p = c.body_w(id_p);
        }
        vec3 world_com_c = c.world_com(id_c);
        const quat qbc = c.body_rot(id_c);
        xform X_wc(world_com_c + quat_rotate(qbc, X_cj.p - c.com(id_c)), qbc * X_cj.q);
        vec3 vel_c = c.body_v(id_c), omega_c = c.body_w(id_c);
        // (the two W tiles once, not six floats per row and body: a world-attached parent gets the zero tile)
        typename Ctx<EPB>::Wsym W_p = c.w_tile(id_p >= 0 ? id_p : 0);
        const typename Ctx<EPB>::Wsym W_c = c.w_tile(id_c);
        if (id_p < 0) W_p = typename Ctx<EPB>::Wsym{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        auto wq_p = [&](vec3 v) { return W_p.quad(v); };
        auto wq_c = [&](vec3 v) { return W_c.quad(v); };

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
                        linear_c = xdiv(anchor_delta, d);
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
                    if (ke > 0.0f) compliance = xrcp(ke);
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
            const vec3 dvel = vel_c - vel_p;  // linear_p = -linear_c: the two linear velocity terms are one dot product
            vec3 lin_c_sum;                   // ... and the parent's linear correction the exact negation of the child's
#pragma unroll
            for (int dim = 0; dim < 3; ++dim) {
                float e = vget(rel_p, dim);
                vec3 linear_c = mat_col(frame_p, dim);
                vec3 linear_p = -linear_c;
                vec3 angular_p = -cross(r_p, linear_c);
                vec3 angular_c = cross(r_c, linear_c);
                float derr = dot(linear_c, dvel) + dot(angular_p, omega_p) + dot(angular_c, omega_c);
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
                    if (st > 0.0f) { err = e - target_pos; compliance = xrcp(st); damping = dm; }
                    else if (dm > 0.0f) { compliance = xrcp(dm); damping = dm; }
                }
                if (fabsf(err) > 1e-9f || fabsf(derr_rel) > 1e-9f) {
                    float d_lambda = positional_correction(err, derr_rel, m_inv_p, m_inv_c, linear_p, linear_c, wq_p(angular_p),
                                                           wq_c(angular_c), 0.0f, compliance, damping, dt);
                    ang_delta_p += angular_p * (d_lambda * P.joint_angular_relaxation);
                    lin_c_sum += linear_c * (d_lambda * P.joint_linear_relaxation);
                    ang_delta_c += angular_c * (d_lambda * P.joint_angular_relaxation);
                }
            }
            lin_delta_c = lin_c_sum;
            lin_delta_p = -lin_c_sum;
        }
    }
    const int ip = c.T.joint_inc[2 * j], ic = c.T.joint_inc[2 * j + 1];  // the (joint, side) entries of the two bodies' lists
    if (ip >= 0) {
        c.st_lv3(c.L.ji, 0, 2 * nj, ip, lin_delta_p);
        c.st_lv3(c.L.ji, 3, 2 * nj, ip, ang_delta_p);
    }
    if (ic >= 0) {
        c.st_lv3(c.L.ji, 0, 2 * nj, ic, lin_delta_c);
        c.st_lv3(c.L.ji, 3, 2 * nj, ic, ang_delta_c);
    }
}

template <int EPB>
NT_DI void joint_angular_item(const Ctx<EPB>& c, const int j) {
    const nt_model& m = c.a.m;
    const int nj = m.nj;
    const nt_xpbd_params& P = c.a.p;
    const float dt = c.a.dt;
    vec3 t0, t1, t2;  // angular_c * d_lambda for the three angular rows (parent gets the negation); stored as their sum
    int id_p, id_c;
    float m_inv_p, m_inv_c;
    const int type = c.T.joint_type[j];
    bool angular_type = type == JT_FIXED || type == JT_PRISMATIC || type == JT_REVOLUTE || type == JT_D6;
    if (angular_type && joint_live(c, j, id_p, id_c, m_inv_p, m_inv_c)) {
        xform X_pj = c.plxf(c.L.jp, 0, nj, j);
        xform X_cj = c.plxf(c.L.jp, 7, nj, j);
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
        typename Ctx<EPB>::Wsym W_p = c.w_tile(id_p >= 0 ? id_p : 0);  // (once, not per row)
        const typename Ctx<EPB>::Wsym W_c = c.w_tile(id_c);
        if (id_p < 0) W_p = typename Ctx<EPB>::Wsym{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

        if (dot(q_p, q_c) < 0.0f) q_c = q_c * -1.0f;
        quat rel_q = quat_inverse(q_p) * q_c;
        // twist about x / swing: the twist quaternion (x, 0, 0, w) / |(x, w)| and the products with it written out -- a general
        // quaternion product multiplies by its literal zeros (IEEE rules forbid folding x * 0), and its norm is the `s` below
        float s = xsqrt(rel_q.x * rel_q.x + rel_q.w * rel_q.w);
        float invs = xrcp(s);
        const bool has_twist = s > 0.0f;
        const float tx = has_twist ? rel_q.x * invs : 0.0f, tw = has_twist ? rel_q.w * invs : 1.0f;  // qtwist = (tx, 0, 0, tw)
        // qswing = rel_q * conj(qtwist)
        const quat qswing(rel_q.x * tw - rel_q.w * tx, rel_q.y * tw - rel_q.z * tx, rel_q.z * tw + rel_q.y * tx, rel_q.w * tw + rel_q.x * tx);
        float invscube = invs * invs * invs;
        float err_0 = 2.0f * asinf(clampf(tx, -1.0f, 1.0f));
        float err_1 = qswing.y, err_2 = qswing.z;
        float g0x = invs - rel_q.x * rel_q.x * invscube, g0w = -(rel_q.w * rel_q.x) * invscube;  // grad_0 = (g0x, 0, 0, g0w)
        quat grad_1(-rel_q.w * (rel_q.w * rel_q.z + rel_q.x * rel_q.y) * invscube, rel_q.w * invs, -rel_q.x * invs,
                    rel_q.x * (rel_q.w * rel_q.z + rel_q.x * rel_q.y) * invscube);
        quat grad_2(rel_q.w * (rel_q.w * rel_q.y - rel_q.x * rel_q.z) * invscube, rel_q.x * invs, rel_q.w * invs,
                    rel_q.x * (rel_q.z * rel_q.x - rel_q.w * rel_q.y) * invscube);
        {
            const float k = xdiv(2.0f, fabsf(tw));
            g0x *= k; g0w *= k;
        }
        float swing_sq = qswing.w * qswing.w;
        if (swing_sq + 1.0e-4f < 1.0f) {
            float d = xsqrt(1.0f - qswing.w * qswing.w);
            float theta = 2.0f * acosf(clampf(qswing.w, -1.0f, 1.0f));
            float scale = xdiv(theta, d);
            err_1 *= scale;
            err_2 *= scale;
            grad_1 = grad_1 * scale;
            grad_2 = grad_2 * scale;
        }
        AxisData A = gather_axes(c, ang_count, axis_start + lin_count, target_axis_start + lin_count);
        // angular_p = -angular_c in every row: the two bodies' quadratic forms are one form of the summed tiles (v^T W v is even in
        // v), and the two velocity terms one dot product with the relative spin
        const typename Ctx<EPB>::Wsym W_pc{W_p.xx + W_c.xx, W_p.xy + W_c.xy, W_p.xz + W_c.xz, W_p.yy + W_c.yy, W_p.yz + W_c.yz, W_p.zz + W_c.zz};
        const vec3 domega = omega_c - omega_p;
        const quat qci = quat_inverse(q_c);
#pragma unroll
        for (int dim = 0; dim < 3; ++dim) {
            float e = dim == 0 ? err_0 : (dim == 1 ? err_1 : err_2);
            // the vector part of 0.5 q_p * grad * conj(q_c)
            quat pg;
            if (dim == 0) pg = quat(q_p.w * g0x + g0w * q_p.x, g0w * q_p.y + q_p.z * g0x, g0w * q_p.z - g0x * q_p.y, q_p.w * g0w - q_p.x * g0x);
            else pg = q_p * (dim == 1 ? grad_1 : grad_2);
            const quat quat_c = pg * qci;
            vec3 angular_c(0.5f * quat_c.x, 0.5f * quat_c.y, 0.5f * quat_c.z);
            float derr = dot(angular_c, domega);
            float err = 0.0f;
            float compliance = P.joint_angular_compliance;
            float damping = 0.0f;
            const float tv = vget(A.target_vel, dim);
            float derr_rel = derr;
            if (tv != 0.0f) derr_rel -= tv * xlength(angular_c);
            float lower = vget(A.lower, dim), upper = vget(A.upper, dim);
            if (e < lower) err = e - lower;
            else if (e > upper) err = e - upper;
            else {
                float target_pos = clampf(vget(A.target_pos, dim), lower, upper);
                float st = vget(A.stiffness, dim), dm = vget(A.damping, dim);
                if (st > 0.0f) { err = e - target_pos; compliance = xrcp(st); damping = dm; }
                else if (dm > 0.0f) { damping = dm; compliance = xrcp(dm); }
            }
            float d_lambda = angular_correction(err, derr_rel, W_pc.quad(angular_c), 0.0f, 0.0f, compliance, damping, dt) *
                             P.joint_angular_relaxation;
            vec3 t = angular_c * d_lambda;
            if (dim == 0) t0 = t;
            else if (dim == 1) t1 = t;
            else t2 = t;
        }
    }
    const vec3 t = (t0 + t1) + t2;  // (angular_p = -angular_c: the parent's entry gets the negation)
    const int ip = c.T.joint_inc[2 * j], ic = c.T.joint_inc[2 * j + 1];
    if (ip >= 0) c.st_lv3(c.L.ji, 6, 2 * nj, ip, -t);
    if (ic >= 0) c.st_lv3(c.L.ji, 6, 2 * nj, ic, t);
}

// apply_rigid_restitution (xpbd/kernels.py:2583-2728) for one contact: the velocity deltas of its two bodies
struct RestitutionOut {
    vec3 lin_a, ang_a, lin_b, ang_b;
    float has_a, has_b;
    int shape_a;
};
// gid_a / gid_b: Newton shape ids of the contact; px_a / px_b: point + offset in the bodies' frames; n: world normal
template <int EPB>
NT_DI RestitutionOut restitution_solve(const Ctx<EPB>& c, int gid_a, int gid_b, vec3 px_a, vec3 px_b, vec3 n) {
    const int nb = c.a.m.nb;
    const float dt = c.a.dt;
    RestitutionOut o;
    o.has_a = 0.0f; o.has_b = 0.0f; o.shape_a = -1;
    if (gid_a == gid_b) return o;
    int shape_a = gid_a >= 0 ? c.local_shape_id(gid_a) : -1;
    int shape_b = gid_b >= 0 ? c.local_shape_id(gid_b) : -1;
    o.shape_a = shape_a;
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
    if (body_a == body_b) return o;
    float m_inv_a = 0.0f, m_inv_b = 0.0f;
    mat33 I_inv_a, I_inv_b;
    xform X_a_prev, X_b_prev;
    vec3 com_a(0.0f), com_b(0.0f);
    auto prev_q = [&](int b) { return c.lxf(c.L.xiq, 0, nb, b); };
    auto prev_qd = [&](int b) { return spatial(c.lv3(c.L.xiqd, 0, nb, b), c.lv3(c.L.xiqd, 3, nb, b)); };
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
    vec3 bx_a = xform_point(X_a_prev, px_a);
    vec3 bx_b = xform_point(X_b_prev, px_b);
    float d = dot(n, bx_b - bx_a);
    if (!(d < 0.0f)) return o;
    vec3 r_a = bx_a - xform_point(X_a_prev, com_a);
    vec3 r_b = bx_b - xform_point(X_b_prev, com_b);
    vec3 gravity = c.gravity();
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
            o.lin_a = n * m_inv_a * dv_a;
            o.ang_a = quat_rotate(X_a_prev.q, I_inv_a * rxn_a * dv_a);
            o.has_a = 1.0f;
        }
        if (body_b >= 0) {
            o.lin_b = n * m_inv_b * dv;
            o.ang_b = quat_rotate(X_b_prev.q, I_inv_b * rxn_b * dv);
            o.has_b = 1.0f;
        }
    }
    return o;
}
// ... for one contact slot; velocity deltas go to the per-contact record
template <int EPB, class CW = CwLds>
NT_DI void restitution_item(const Ctx<EPB>& c, const int slot) {
    const nt_model& m = c.a.m;
    const nt_contacts& ct = c.a.ct;
    const int cpp = m.cpp, ncs = m.np * cpp;
    const float* D = ct.data;
    size_t gi = (size_t)slot * c.ES + c.env;
    const int gid_a = ct.shape0[gi], gid_b = ct.shape1[gi];
    RestitutionOut o;
    o.has_a = 0.0f; o.has_b = 0.0f; o.shape_a = -1;
    if (gid_a != gid_b)
        o = restitution_solve(c, gid_a, gid_b, c.gv3(D, CD_POINT0, ncs, slot) + c.gv3(D, CD_OFFSET0, ncs, slot),
                              c.gv3(D, CD_POINT1, ncs, slot) + c.gv3(D, CD_OFFSET1, ncs, slot), c.gv3(D, CD_NORMAL, ncs, slot));
    const float a_is_pair_a = (o.has_a != 0.0f || o.has_b != 0.0f) ? ((o.shape_a == c.T.pair_a[slot / cpp]) ? 1.0f : 0.0f) : 1.0f;
    cw_st3<CW, NC_CW>(c, 0, ncs, slot, o.lin_a);
    cw_st3<CW, NC_CW>(c, 3, ncs, slot, o.ang_a);
    cw_st3<CW, NC_CW>(c, 6, ncs, slot, o.lin_b);
    cw_st3<CW, NC_CW>(c, 9, ncs, slot, o.ang_b);
    CW::template at<NC_CW>(c, 12, ncs, slot) = o.has_a;
    CW::template at<NC_CW>(c, 13, ncs, slot) = o.has_b;
    CW::template at<NC_CW>(c, 14, ncs, slot) = a_is_pair_a;
}
// ... and for one row of the SDF legs: nt_flat_rows.restitution[row][14] = lin_a, ang_a, lin_b, ang_b, has_a, has_b
constexpr int FLAT_REST_FLOATS = 14;
template <int EPB>
NT_DI void restitution_flat_item(const Ctx<EPB>& c, const int r) {
    const nt_flat_rows& f = c.a.ct.flat;
    const RestitutionOut o = restitution_solve(c, f.shape0[r], f.shape1[r], FlatRecord::ld(f.point0, r) + FlatRecord::ld(f.offset0, r),
                                               FlatRecord::ld(f.point1, r) + FlatRecord::ld(f.offset1, r), FlatRecord::ld(f.normal, r));
    float* w = f.restitution + FLAT_REST_FLOATS * (size_t)r;
    w[0] = o.lin_a.x; w[1] = o.lin_a.y; w[2] = o.lin_a.z; w[3] = o.ang_a.x; w[4] = o.ang_a.y; w[5] = o.ang_a.z;
    w[6] = o.lin_b.x; w[7] = o.lin_b.y; w[8] = o.lin_b.z; w[9] = o.ang_b.x; w[10] = o.ang_b.y; w[11] = o.ang_b.z;
    w[12] = o.has_a; w[13] = o.has_b;
}
// apply_body_delta_velocities (xpbd/kernels.py:936-942): body lane sums its contacts' velocity deltas in contact order
// ROWS: the launch-by-launch step kernel, whose Contacts may carry rows of the SDF legs (the fused rollouts never do)
template <int EPB, class CW = CwLds, bool ROWS = false>
NT_DI void restitution_apply_item(const Ctx<EPB>& c, const int b) {
    const nt_model& m = c.a.m;
    const int cpp = m.cpp, ncs = m.np * cpp, nb = m.nb;
    vec3 dv, dw;
    for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
        int code = c.T.body_pair_list[i];
        int p = code >> 1, side = code & 1;
        for (int k = 0; k < cpp; ++k) {
            int slot = p * cpp + k;
            bool is_a = (side == 0) == (CW::template at<NC_CW>(c, 14, ncs, slot) != 0.0f);
            if (CW::template at<NC_CW>(c, is_a ? 12 : 13, ncs, slot) != 0.0f) {
                dv += cw_v3<CW, NC_CW>(c, is_a ? 0 : 6, ncs, slot);
                dw += cw_v3<CW, NC_CW>(c, is_a ? 3 : 9, ncs, slot);
            }
        }
    }
    if constexpr (ROWS)
    if (const nt_flat_rows& f = c.a.ct.flat; f.row_start && f.restitution) {  // then the SDF legs' rows, ascending row order
        const int* bs = f.body_blk_start + (size_t)c.env * (nb + 1) + b;
        for (int i = bs[0]; i < bs[1]; ++i) {
            const int code = f.body_blk_list[2 * (size_t)i], count = f.body_blk_list[2 * (size_t)i + 1];
            const int r0 = code >> 1, side = code & 1;  // side 0: the body of the rows' shape0
            for (int r = r0; r < r0 + count; ++r) {
                const float* w = f.restitution + FLAT_REST_FLOATS * (size_t)r;
                if (w[side ? 13 : 12] != 0.0f) {
                    dv += vec3(w[side ? 6 : 0], w[side ? 7 : 1], w[side ? 8 : 2]);
                    dw += vec3(w[side ? 9 : 3], w[side ? 10 : 4], w[side ? 11 : 5]);
                }
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
        const int ic = c.T.joint_inc[2 * j + 1];  // the child body's entry of this joint
        vec3 jl = c.lv3(c.L.ji, 0, 2 * nj, ic);
        vec3 ja = c.lv3(c.L.ji, 3, 2 * nj, ic) + c.lv3(c.L.ji, 6, 2 * nj, ic);
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
// number of active contacts on body b in this iteration (constraint_inv_weight[b], xpbd/kernels.py:2287-2291): the slots, and the rows
// of the SDF legs that correct it (the same rows apply_item sums)
template <int EPB, class CW = CwLds>
NT_DI float report_body_contact_count(const Ctx<EPB>& c, int b) {
    const nt_model& m = c.a.m;
    const int cpp = m.cpp, ncs = m.np * cpp;
    float n = 0.0f;
    if (const nt_flat_rows& f = c.a.ct.flat; f.row_start) {
        const int* bs = f.body_blk_start + (size_t)c.env * (m.nb + 1) + b;
        for (int i = bs[0]; i < bs[1]; ++i) {
            const int code = f.body_blk_list[2 * (size_t)i], count = f.body_blk_list[2 * (size_t)i + 1];
            const int r0 = code >> 1, side = code & 1;
            for (int r = r0; r < r0 + count; ++r)
                if ((int)f.cw[CWX_FLOATS * (size_t)r + CWX_FLAGS] & (side ? 2 : 1)) n += 1.0f;
        }
    }
    for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
        int code = c.T.body_pair_list[i];
        int p = code >> 1, side = code & 1;
        for (int k = 0; k < cpp; ++k) {
            int slot = p * cpp + k;
            if (cwx_side<CW>(c, ncs, slot, side).has) n += 1.0f;
        }
    }
    return n;
}
// contact_impulse[slot] (+)= (lin_delta_a, ang_delta_a) * weight   (accumulate_weighted_contact_impulse)
template <int EPB, class CW = CwLds>
NT_DI void report_contact_iteration(const Ctx<EPB>& c, bool first) {
    if (!c.valid) return;
    const nt_model& m = c.a.m;
    const int cpp = m.cpp, ncs = m.np * cpp;
    float* I = c.a.rep.contact_impulse;
    for (int slot = c.slot; slot < ncs; slot += c.nslot) {
        const int flags = (int)CW::template at<NC_CWX>(c, CWX_FLAGS, ncs, slot);
        float has_a = (flags & 1) ? 1.0f : 0.0f, has_b = (flags & 2) ? 1.0f : 0.0f;
        vec3 lin, ang;
        if (has_a != 0.0f || has_b != 0.0f) {
            float weight = 1.0f;
            if (c.a.p.rigid_contact_con_weighting) {
                const int p = slot / cpp;
                int sa = c.T.pair_a[p], sb = c.T.pair_b[p];
                if (!(flags & 4)) { int t = sa; sa = sb; sb = t; }
                int body_a = c.T.shape_body[sa], body_b = c.T.shape_body[sb];
                float n_a = body_a >= 0 ? report_body_contact_count<EPB, CW>(c, body_a) : 0.0f;
                float n_b = body_b >= 0 ? report_body_contact_count<EPB, CW>(c, body_b) : 0.0f;
                float n_sum = n_a + n_b;
                if (n_sum > 0.0f) {
                    if (n_a == 0.0f) weight = 1.0f / n_b;
                    else if (n_b == 0.0f) weight = 1.0f / n_a;
                    else weight = 2.0f / n_sum;
                }
            }
            lin = cw_v3<CW, NC_CWX>(c, 0, ncs, slot) * weight;
            ang = cw_v3<CW, NC_CWX>(c, CWX_ANG_A, ncs, slot) * weight;
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

// the same accumulation for the rows of the SDF legs: flat.impulse[row] (+)= (lin_delta_a, ang_delta_a) * weight
template <int EPB, class CW = CwLds>
NT_DI void report_flat_rows_iteration(const Ctx<EPB>& c, bool first) {
    const nt_flat_rows& f = c.a.ct.flat;
    if (!c.valid || !f.row_start || !f.impulse) return;
    for (int r = f.row_start[c.env] + c.slot; r < f.row_start[c.env + 1]; r += c.nslot) {
        const float* w = f.cw + CWX_FLOATS * (size_t)r;
        const int flags = (int)w[CWX_FLAGS];
        const bool has = (flags & 3) != 0;
        vec3 lin, ang;
        if (has) {
            float weight = 1.0f;
            if (c.a.p.rigid_contact_con_weighting) {
                const int gid_a = f.shape0[r], gid_b = f.shape1[r];
                const int shape_a = gid_a >= 0 ? c.local_shape_id(gid_a) : -1, shape_b = gid_b >= 0 ? c.local_shape_id(gid_b) : -1;
                const int body_a = shape_a >= 0 ? c.T.shape_body[shape_a] : -1, body_b = shape_b >= 0 ? c.T.shape_body[shape_b] : -1;
                float n_a = body_a >= 0 ? report_body_contact_count<EPB, CW>(c, body_a) : 0.0f;
                float n_b = body_b >= 0 ? report_body_contact_count<EPB, CW>(c, body_b) : 0.0f;
                float n_sum = n_a + n_b;
                if (n_sum > 0.0f) {
                    if (n_a == 0.0f) weight = 1.0f / n_b;
                    else if (n_b == 0.0f) weight = 1.0f / n_a;
                    else weight = 2.0f / n_sum;
                }
            }
            lin = vec3(w[0], w[1], w[2]) * weight;
            ang = vec3(w[CWX_ANG_A], w[CWX_ANG_A + 1], w[CWX_ANG_A + 2]) * weight;
        }
        float* I = f.impulse + 6 * (size_t)r;
        if (first) {
            I[0] = lin.x; I[1] = lin.y; I[2] = lin.z; I[3] = ang.x; I[4] = ang.y; I[5] = ang.z;
        } else if (has) {
            I[0] += lin.x; I[1] += lin.y; I[2] += lin.z; I[3] += ang.x; I[4] += ang.y; I[5] += ang.z;
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
    const int spw = 64 / Ctx<EPB>::N > 0 ? 64 / Ctx<EPB>::N : 1;
    const int A0 = ((nj + spw - 1) / spw) * spw;
    NT_SKIP_DECL(c.a);  // (measurement builds: 32 skips the linear-row lanes, 64 the angular-row lanes)
    for (int i = c.slot; i < A0 + nj; i += c.nslot) {
        if (i < nj) { if (!NT_SKIP(32)) joint_linear_item(c, i); }
        else if (i >= A0) { if (!NT_SKIP(64)) joint_angular_item(c, i - A0); }
    }
}

// SolverXPBD.step control flow (solver_xpbd.py:329-862), rigid-only model
// PROLOGUE_DONE: the caller (do_fused_substep) already saved the pre-step state, applied the joint forces and integrated
// ROWS: see phase_contacts (a step kernel that compacts its live slots like the fused rollouts AND walks the rows of the SDF legs)
template <int EPB, bool FUSED, class CW = CwLds, bool PROLOGUE_DONE = false, bool ROWS = !FUSED>
NT_DI void do_xpbd_step(const Ctx<EPB>& c, bool forces_are_zero) {
    const nt_model& m = c.a.m;
    NT_SKIP_DECL(c.a);
    const bool restitution = c.a.p.enable_restitution && c.a.has_contacts;
    const bool vel_from_delta = c.a.p.compute_body_velocity_from_position_delta != 0;
    if (!PROLOGUE_DONE && (restitution || vel_from_delta) && c.valid)  // body_q_init / body_qd_init: the state the step starts from
        for (int r = c.slot; r < 14 * m.nb; r += c.nslot) c.lds[(c.L.xiq.off + r) * Ctx<EPB>::N + c.e] = c.lds[(c.L.bq.off + r) * Ctx<EPB>::N + c.e];
    const bool rep_joints = !FUSED && c.a.rep.joint_impulse != nullptr;
    const bool rep_contacts = !FUSED && c.a.rep.contact_impulse != nullptr && c.a.has_contacts;
    if (!PROLOGUE_DONE && !NT_SKIP(2)) {
        phase_joint_forces(c, forces_are_zero);
        __syncthreads();
        NT_TICK(3);
        if (rep_joints) report_joint_forces(c);
        phase_xpbd_integrate(c, !xpbd_applies_follow(c.a));
        __syncthreads();
        NT_TICK(4);
    }
    const int iterations = c.a.p.iterations;
    for (int it = 0; it < iterations; ++it) {
        const bool last_it = it == iterations - 1;
        if (c.a.has_contacts) {
            if (!NT_SKIP(4)) phase_contacts<EPB, FUSED, CW, ROWS>(c);
            __syncthreads();
            NT_TICK(5);
            if (rep_contacts) {
                report_contact_iteration<EPB, CW>(c, it == 0);
                if constexpr (!FUSED) report_flat_rows_iteration<EPB, CW>(c, it == 0);
            }
            if (!NT_SKIP(16)) phase_apply<EPB, true, CW, FUSED, ROWS>(c, last_it && m.nj <= 0);
            __syncthreads();
            NT_TICK(6);
        }
        if (m.nj > 0) {
            if (!NT_SKIP(8)) phase_joints(c);
            __syncthreads();
            NT_TICK(7);
            if (rep_joints) report_joint_iteration(c);
            if (!NT_SKIP(16)) phase_apply<EPB, false>(c, last_it);
            __syncthreads();
            NT_TICK(8);
        }
    }
    if (vel_from_delta) {  // update_body_velocities (xpbd/kernels.py:2547-2579, solver_xpbd.py:767-783)
        if (c.valid)
            for (int b = c.slot; b < m.nb; b += c.nslot) {
                const xform pose = c.body_q(b), prev = c.lxf(c.L.xiq, 0, m.nb, b);
                const vec3 com = c.com(b);
                const vec3 x_com = pose.p + quat_rotate(pose.q, com), x_com_prev = prev.p + quat_rotate(prev.q, com);
                const vec3 v = (x_com - x_com_prev) / c.a.dt;
                const quat dq = pose.q * quat_inverse(prev.q);
                vec3 omega = (2.0f / c.a.dt) * vec3(dq.x, dq.y, dq.z);
                if (dq.w < 0.0f) omega = -omega;
                c.st_lv3(c.L.bqd, 0, m.nb, b, v);
                c.st_lv3(c.L.bqd, 3, m.nb, b, omega);
            }
        __syncthreads();
    }
    if (restitution) {  // solver_xpbd.py:784-858
        if (c.valid)
            for (int s = c.slot; s < m.np * m.cpp; s += c.nslot) restitution_item<EPB, CW>(c, s);
        if constexpr (ROWS) {
            if (const nt_flat_rows& f = c.a.ct.flat; c.valid && f.row_start && f.restitution)
                for (int r = f.row_start[c.env] + c.slot; r < f.row_start[c.env + 1]; r += c.nslot) restitution_flat_item(c, r);
        }
        __syncthreads();
        if (c.valid)
            for (int b = c.slot; b < m.nb; b += c.nslot)
                if (!(c.T.body_flags[b] & BODY_KINEMATIC)) restitution_apply_item<EPB, CW, ROWS>(c, b);
        __syncthreads();
    }
    if (!FUSED && c.a.s_out.body_parent_f) {
        __threadfence_block();  // joint lanes' accumulators -> body lanes
        __syncthreads();
        report_parent_f(c);
    }
}

