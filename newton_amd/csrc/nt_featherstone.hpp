// SolverFeatherstone for gfx950: one workgroup owns EPB environments; generalized state, motion subspaces, spatial
// inertias, the articulation's joint-space inertia H and its Cholesky factor all live in LDS (the reference streams
// J, M, P = M J, H and L through HBM: solver_featherstone.py:771-934).
// The phases (this file) are included by nt_kernels.hip in namespace ieee after nt_ctx.hpp / nt_xpbd.hpp / nt_semi_implicit.hpp (they
// use Ctx / KArgs / si_contact_item / si_dof_force).  FsLayout lives in nt_layout.hpp, the kernels in nt_featherstone_kernels.hpp.
//
// Reference (restated):
//   jcalc_transform / jcalc_motion / jcalc_tau / jcalc_integrate   newton/_src/solvers/featherstone/kernels.py:142-630
//   eval_rigid_fk / compute_link_velocity / eval_rigid_id          kernels.py:633-866,1241-1317
//   public <-> internal FREE-joint velocity conversions            kernels.py:924-1088
//   eval_rigid_tau / eval_rigid_jacobian / eval_rigid_mass         kernels.py:1320-1501
//   dense_gemm / dense_cholesky / dense_subs                       kernels.py:1504-1565,1690-1797
//   integrate_generalized_joints + FK with velocity conversion     kernels.py:1849-1893,1987-2150
//   SolverFeatherstone.step                                        solver_featherstone.py:462-1066
// Summation orders follow the reference: H[i][j] = sum over bodies (ascending) and spatial rows (ascending) of
// J[b,r,i] * (I_b S_j)[r] -- the zero entries of the dense J / M the reference multiplies through add exact zeros.
// Scope: PRISMATIC, REVOLUTE, BALL, FIXED, FREE / DISTANCE (root and descendant), D6; body l of an articulation is the child
// of its joint l (the reference's eval_rigid_mass indexes body_I_s by joint index, kernels.py:1466-1480).

template <int EPB>
struct FsCtx {
    const Ctx<EPB>& c;
    FsLayout F;
    const int *anc, *depth, *art;  // [nj] each (LDS)
    const int* dof_joint;          // [nd]
    const unsigned* pathmask;      // [nj][words]
    const unsigned* childmask;     // [nj][words] joints whose parent body is this joint's child
    const int* refresh;            // [nj] joint index >= first descendant FREE / DISTANCE joint of its articulation; [nj] = any
    int words;
    // dof tree of the tree-structured factorisation (fs_build_tables; nt_layout.hpp fs_topo_ints): 64-bit masks as (lo, hi) pairs
    const unsigned *t_below, *t_above, *t_lvl, *t_jbelow;
    const int *ddepth, *rowbase, *ent;
    int nnz, dmaxd;
    bool tree_ok;
    static constexpr int N = Ctx<EPB>::N;  // environments per workgroup (the tile code EPB may carry the uniform-parameter bit)
    NT_DI FsCtx(const Ctx<EPB>& c_, int* extra) : c(c_) {
        F = make_fs_layout(c.a.m, c.L, fs_tree_mode(c.a));
        const int nj = c.a.m.nj;
        anc = extra;
        depth = extra + nj;
        art = extra + 2 * nj;
        dof_joint = extra + 3 * nj;
        pathmask = reinterpret_cast<const unsigned*>(extra + 3 * nj + c.a.m.nd);
        words = fs_mask_words(c.a.m);
        childmask = pathmask + nj * words;
        refresh = reinterpret_cast<const int*>(childmask + nj * words);
        const int nd = c.a.m.nd;
        tree_ok = fs_tree_ok(c.a.m);
        const int* tr = extra + fs_topo_base_ints(c.a.m);
        t_below = reinterpret_cast<const unsigned*>(tr);
        t_above = t_below + 2 * nd;
        t_lvl = t_above + 2 * nd;
        t_jbelow = t_lvl + 2 * (nd + 1);
        ddepth = reinterpret_cast<const int*>(t_jbelow + 2 * nj);
        rowbase = ddepth + nd;
        nnz = tree_ok ? rowbase[nd] : 0;
        dmaxd = tree_ok ? rowbase[nd + 1] : 0;
        ent = rowbase + nd + 2;
    }
    static NT_DI unsigned long long m64(const unsigned* p, int i) { return (unsigned long long)p[2 * i] | ((unsigned long long)p[2 * i + 1] << 32); }
    // FREE / DISTANCE joint below the root whose child is dynamic (solver_featherstone.py:229-237)
    NT_DI bool descendant_free(int j) const {
        const int t = c.T.joint_type[j];
        return (t == JT_FREE || t == JT_DISTANCE) && c.T.joint_parent[j] >= 0 && !(c.T.body_flags[c.T.joint_child[j]] & BODY_KINEMATIC);
    }
    NT_DI bool any_descendant_free() const { return refresh[c.a.m.nj] != 0; }
    // is joint `a` on the root path of joint `l` (or `l` itself)?
    NT_DI bool on_path(int a, int l) const { return (pathmask[l * words + (a >> 5)] >> (a & 31)) & 1u; }
    NT_DI float& f(int off, int idx) const { return c.lds[(off + idx) * N + c.e]; }
    NT_DI vec3 v3(int off, int comp0, int n, int s) const { return c.lv3(off, comp0, n, s); }
    NT_DI spatial sp6(int off, int n, int s) const { return spatial(c.lv3(off, 0, n, s), c.lv3(off, 3, n, s)); }
    NT_DI void st6(int off, int n, int s, const spatial& x) const {
        c.st_lv3(off, 0, n, s, x.top);
        c.st_lv3(off, 3, n, s, x.bottom);
    }
};

// (linear, angular) twist transform (math/spatial.py:82-104)
NT_DI spatial fs_transform_twist(const xform& t, const spatial& x) {
    vec3 w = quat_rotate(t.q, x.bottom);
    vec3 v = quat_rotate(t.q, x.top) + cross(t.p, w);
    return spatial(v, w);
}
NT_DI spatial fs_spatial_cross(const spatial& a, const spatial& b) {
    return spatial(cross(a.bottom, b.top) + cross(a.top, b.bottom), cross(a.bottom, b.bottom));
}
NT_DI spatial fs_spatial_cross_dual(const spatial& a, const spatial& b) {
    return spatial(cross(a.bottom, b.top), cross(a.bottom, b.bottom) + cross(a.top, b.top));
}
NT_DI float fs_sget(const spatial& s, int i) { return i < 3 ? vget(s.top, i) : vget(s.bottom, i - 3); }

struct mat66 {
    float a[6][6];
};
NT_DI spatial fs_mul(const mat66& A, const spatial& v) {
    float r[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) sum += A.a[i][j] * fs_sget(v, j);
        r[i] = sum;
    }
    return spatial(vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]));
}

// rotation and transported angular axes of a D6 joint with two or three angular axes (compute_2d_rotational_dofs /
// compute_3d_rotational_dofs, newton/_src/sim/articulation.py:36-83,127-178); used by newton.eval_fk and by the
// Featherstone passes (jcalc_transform / jcalc_motion, featherstone/kernels.py:186-232,266-335)
NT_DI quat d6_multi_angular(int ang, vec3 axis_0, vec3 axis_1, vec3 axis_2, float q0, float q1, float q2, vec3& a0, vec3& a1,
                            vec3& a2) {
    if (ang == 2) {
        quat q_off = quat_from_matrix(matrix_from_cols(axis_0, axis_1, cross(axis_0, axis_1)));
        vec3 local_0 = quat_rotate(q_off, vec3(1.0f, 0.0f, 0.0f)), local_1 = quat_rotate(q_off, vec3(0.0f, 1.0f, 0.0f));
        a0 = local_0;
        quat q_0 = quat_from_axis_angle(a0, q0);
        a1 = quat_rotate(q_0, local_1);
        a2 = vec3();
        quat q_1 = quat_from_axis_angle(a1, q1);
        return q_1 * q_0;
    }
    a0 = axis_0;
    quat q_0 = quat_from_axis_angle(a0, q0);
    a1 = quat_rotate(q_0, axis_1);
    quat q_1 = quat_from_axis_angle(a1, q1);
    a2 = quat_rotate(q_1 * q_0, axis_2);
    quat q_2 = quat_from_axis_angle(a2, q2);
    return q_2 * q_1 * q_0;
}

// jcalc_transform (kernels.py:142-239)
template <int EPB>
NT_DI xform fs_joint_transform(const FsCtx<EPB>& f, int type, int qd_start, int lin, int ang, int q_off, int q_start) {
    const Ctx<EPB>& c = f.c;
    if (type == JT_PRISMATIC) return xform(c.dof_axis(qd_start) * f.f(q_off, q_start), quat_identity());
    if (type == JT_REVOLUTE) return xform(vec3(), quat_from_axis_angle(c.dof_axis(qd_start), f.f(q_off, q_start)));
    if (type == JT_BALL)
        return xform(vec3(), quat(f.f(q_off, q_start), f.f(q_off, q_start + 1), f.f(q_off, q_start + 2), f.f(q_off, q_start + 3)));
    if (type == JT_FREE || type == JT_DISTANCE)
        return xform(vec3(f.f(q_off, q_start), f.f(q_off, q_start + 1), f.f(q_off, q_start + 2)),
                     quat(f.f(q_off, q_start + 3), f.f(q_off, q_start + 4), f.f(q_off, q_start + 5), f.f(q_off, q_start + 6)));
    if (type == JT_D6) {
        vec3 pos(0.0f);
        quat rot = quat_identity();
        if (lin > 0) pos += c.dof_axis(qd_start + 0) * f.f(q_off, q_start + 0);
        if (lin > 1) pos += c.dof_axis(qd_start + 1) * f.f(q_off, q_start + 1);
        if (lin > 2) pos += c.dof_axis(qd_start + 2) * f.f(q_off, q_start + 2);
        if (ang == 1) rot = quat_from_axis_angle(c.dof_axis(qd_start + lin), f.f(q_off, q_start + lin));
        if (ang >= 2) {
            vec3 a0, a1, a2;
            rot = d6_multi_angular(ang, c.dof_axis(qd_start + lin), c.dof_axis(qd_start + lin + 1),
                                   ang == 3 ? c.dof_axis(qd_start + lin + 2) : vec3(), f.f(q_off, q_start + lin),
                                   f.f(q_off, q_start + lin + 1), ang == 3 ? f.f(q_off, q_start + lin + 2) : 0.0f, a0, a1, a2);
        }
        return xform(pos, rot);
    }
    return xform();
}

// jcalc_transform for every joint at once (the sin / cos of the joint angles are the expensive part of FK and do not
// depend on the tree level); parked in the v_s / a_s rows, which are dead during both FK passes
template <int EPB>
NT_DI void fs_joint_xform_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    xform X_j = fs_joint_transform(f, c.T.joint_type[j], c.T.joint_qd_start[j], c.T.joint_lin_count[j], c.T.joint_ang_count[j],
                                   f.F.jq, c.T.joint_q_start[j]);
    c.st_lxf(f.F.vs, c.a.m.nj, j, X_j);
}

// compute_link_transform (kernels.py:633-684): body_q[child], COM world position
template <int EPB>
NT_DI void fs_fk_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int parent = c.T.joint_parent[j], child = c.T.joint_child[j];
    xform X_wpj = c.plxf(c.L.jp, 0, m.nj, j);
    if (parent >= 0) X_wpj = c.body_q(parent) * X_wpj;
    xform X_j = c.lxf(f.F.vs, 0, m.nj, j);
    xform X_wcj = X_wpj * X_j;
    xform X_wc = X_wcj * xform_inverse(c.plxf(c.L.jp, 7, m.nj, j));
    c.st_lxf(c.L.bq, m.nb, child, X_wc);
    xform X_sm = X_wc * xform(c.com(child), quat_identity());
    c.st_lv3(f.F.qcom, 0, m.nb, child, X_sm.p);
}

// FREE/DISTANCE anchor offset r_child_com_parent (kernels.py:946-954,1037-1044), on the body poses currently in LDS
template <int EPB>
NT_DI vec3 fs_free_com_offset(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const int parent = c.T.joint_parent[j], child = c.T.joint_child[j];
    xform X_wpj = c.plxf(c.L.jp, 0, c.a.m.nj, j);
    if (parent >= 0) X_wpj = c.body_q(parent) * X_wpj;
    vec3 x_child_com_world = xform_point(c.body_q(child), c.com(child));
    return quat_rotate_inv(X_wpj.q, x_child_com_world - X_wpj.p);
}

// convert_free_distance_joint_qd_public_to_internal + joint_f_public_to_internal (kernels.py:924-975,1069-1088)
template <int EPB>
NT_DI void fs_to_internal_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int qs = c.T.joint_qd_start[j], type = c.T.joint_type[j];
    const int qe = j + 1 < m.nj ? c.T.joint_qd_start[j + 1] : m.nd;
    auto pub = [&](int i) { return f.f(f.F.qdp, i); };
    if (type != JT_FREE && type != JT_DISTANCE) {
        for (int i = qs; i < qe; ++i) {
            f.f(f.F.qdi, i) = pub(i);
            f.f(f.F.jfi, i) = c.l(c.L.cf, 0, 1, i);
        }
        return;
    }
    vec3 r = fs_free_com_offset(f, j);
    vec3 v_com(pub(qs), pub(qs + 1), pub(qs + 2)), omega(pub(qs + 3), pub(qs + 4), pub(qs + 5));
    vec3 v_int = v_com - cross(omega, r);
    f.f(f.F.qdi, qs + 0) = v_int.x; f.f(f.F.qdi, qs + 1) = v_int.y; f.f(f.F.qdi, qs + 2) = v_int.z;
    f.f(f.F.qdi, qs + 3) = omega.x; f.f(f.F.qdi, qs + 4) = omega.y; f.f(f.F.qdi, qs + 5) = omega.z;
    for (int i = qs; i < qe; ++i) f.f(f.F.jfi, i) = 0.0f;
}

// transform_spatial_inertia (kernels.py:66-139): T^T I T with T = [[R, skew(p) R], [0, R]] of the inverse transform and
// I = diag(m 1, I_b).  The reference multiplies the dense 6 x 6 matrices; here every sum keeps the reference's ascending-k order but
// only the terms whose T / I factor is not a structural zero.  Bit-identical for finite inputs: a dropped term is 0 * x = +-0, a sum
// that starts at +0 cannot become -0, and s + (+-0) = s -- 330 instead of 920 VALU operations per body and substep (contraction
// is off in this namespace, the compiler may not fold 0 * x).
NT_DI void fs_transform_spatial_inertia(const xform& t, float mass, const mat33& Ib, mat66& out) {
    xform t_inv = xform_inverse(t);
    quat q = t_inv.q;
    vec3 p = t_inv.p;
    vec3 r1 = quat_rotate(q, vec3(1.0f, 0.0f, 0.0f));
    vec3 r2 = quat_rotate(q, vec3(0.0f, 1.0f, 0.0f));
    vec3 r3 = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
    const float R[3][3] = {{r1.x, r2.x, r3.x}, {r1.y, r2.y, r3.y}, {r1.z, r2.z, r3.z}};
    const float K[3][3] = {{0.0f, -p.z, p.y}, {p.z, 0.0f, -p.x}, {-p.y, p.x, 0.0f}};
    float S[3][3];  // skew(p) R: the diagonal of skew(p) is a structural zero
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k != i) sum += K[i][k] * R[k][j];
            S[i][j] = sum;
        }
    const float Im[3][3] = {{Ib.m00, Ib.m01, Ib.m02}, {Ib.m10, Ib.m11, Ib.m12}, {Ib.m20, Ib.m21, Ib.m22}};
    // A = T^T I: A[i][j] = sum_k T[k][i] I[k][j].  Blocks: [R^T m | 0; S^T m | R^T I_b]
    float Amm[3][3], Asm[3][3], Arr[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Amm[i][j] = 0.0f + R[j][i] * mass;  // the one k with I[k][j] != 0 is k = j
            Asm[i][j] = 0.0f + S[j][i] * mass;
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) sum += R[k][i] * Im[k][j];
            Arr[i][j] = sum;
        }
    // out = A T: out[i][j] = sum_k A[i][k] T[k][j]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s00 = 0.0f, s10 = 0.0f, s01 = 0.0f, s11 = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s00 += Amm[i][k] * R[k][j];
                s10 += Asm[i][k] * R[k][j];
                s01 += Amm[i][k] * S[k][j];  // (A[i][3..5] = 0 for i < 3: the lower half of the sum is dropped)
                s11 += Asm[i][k] * S[k][j];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) s11 += Arr[i][k] * R[k][j];
            out.a[i][j] = s00;
            out.a[i + 3][j] = s10;
            out.a[i][j + 3] = s01;
            out.a[i + 3][j + 3] = s11;
        }
}

// compute_link_velocity (kernels.py:764-866) incl. jcalc_motion (kernels.py:242-380), split in two:
// (1) everything that does not depend on the parent's velocity -- motion subspace columns S, the joint velocity v_j_s
//     and the solve-frame spatial inertia I_s -- runs for all joints at once;
// do_motion: S, v_j_s, c_app_s; do_inertia: solve origin + I_s.  The two halves share nothing but their inputs, so a workgroup
// with idle slot lanes runs them side by side (fs_substep); together they are the reference's per-joint program.
template <int EPB>
NT_DI void fs_motion_pre_item(const FsCtx<EPB>& f, int j, const bool do_motion = true, const bool do_inertia = true) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nb = m.nb, nd = m.nd;
    const int type = c.T.joint_type[j], child = c.T.joint_child[j], parent = c.T.joint_parent[j];
    const int qd_start = c.T.joint_qd_start[j];
    const int lin = c.T.joint_lin_count[j], ang = c.T.joint_ang_count[j];
    // solve origin: COM of the articulation root's child when the root is FREE/DISTANCE (kernels.py:1280-1288)
    int root = j;
    while (f.anc[root] >= 0) root = f.anc[root];
    vec3 solve_origin;
    {
        int rt = c.T.joint_type[root];
        if (rt == JT_FREE || rt == JT_DISTANCE) solve_origin = f.v3(f.F.qcom, 0, nb, c.T.joint_child[root]);
    }
    if (do_inertia) {
        vec3 x_com_s = f.v3(f.F.qcom, 0, nb, child) - solve_origin;
        c.st_lv3(f.F.org, 0, nb, child, solve_origin);
        mat66 I_s;
        fs_transform_spatial_inertia(xform(x_com_s, c.body_rot(child)), c.pl(c.L.bp, BP_MASS, nb, child), c.inertia(child), I_s);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) c.l(f.F.Is, i * 6 + k, nb, child) = I_s.a[i][k];
    }
    if (!do_motion) return;
    xform X_wpj = c.plxf(c.L.jp, 0, m.nj, j);
    if (parent >= 0) X_wpj = c.body_q(parent) * X_wpj;
    xform X_sc(X_wpj.p - solve_origin, X_wpj.q);

    spatial v_j_s, c_app_s;
    auto qd = [&](int i) { return f.f(f.F.qdi, i); };
    auto put_S = [&](int d, const spatial& S_s) { f.st6(f.F.S, nd, d, S_s); };
    if (type == JT_PRISMATIC) {
        spatial S_s = fs_transform_twist(X_sc, spatial(c.dof_axis(qd_start), vec3()));
        put_S(qd_start, S_s);
        v_j_s = S_s * qd(qd_start);
    } else if (type == JT_REVOLUTE) {
        spatial S_s = fs_transform_twist(X_sc, spatial(vec3(), c.dof_axis(qd_start)));
        put_S(qd_start, S_s);
        v_j_s = S_s * qd(qd_start);
    } else if (type == JT_D6) {
        for (int k = 0; k < 3; ++k)
            if (lin > k) {
                spatial S_s = fs_transform_twist(X_sc, spatial(c.dof_axis(qd_start + k), vec3()));
                v_j_s = v_j_s + S_s * qd(qd_start + k);
                put_S(qd_start + k, S_s);
            }
        if (ang == 1) {
            int iqd = qd_start + lin;
            spatial S_s = fs_transform_twist(X_sc, spatial(vec3(), c.dof_axis(iqd)));
            v_j_s = v_j_s + S_s * qd(iqd);
            put_S(iqd, S_s);
        }
        if (ang >= 2) {  // FK-transported axes; their dependence on q gives the apparent derivative c_app
            const int iqd = qd_start + lin, iq = c.T.joint_q_start[j] + lin;
            vec3 a0, a1, a2;
            d6_multi_angular(ang, c.dof_axis(iqd), c.dof_axis(iqd + 1), ang == 3 ? c.dof_axis(iqd + 2) : vec3(), f.f(f.F.jq, iq),
                             f.f(f.F.jq, iq + 1), ang == 3 ? f.f(f.F.jq, iq + 2) : 0.0f, a0, a1, a2);
            float qd0 = qd(iqd), qd1 = qd(iqd + 1);
            spatial S_0 = fs_transform_twist(X_sc, spatial(vec3(), a0)), S_1 = fs_transform_twist(X_sc, spatial(vec3(), a1));
            vec3 c_app_ang;
            put_S(iqd, S_0);
            put_S(iqd + 1, S_1);
            if (ang == 2) {
                v_j_s = v_j_s + (S_0 * qd0 + S_1 * qd1);
                c_app_ang += cross(a0, a1) * (qd0 * qd1);
            } else {
                float qd2 = qd(iqd + 2);
                spatial S_2 = fs_transform_twist(X_sc, spatial(vec3(), a2));
                put_S(iqd + 2, S_2);
                v_j_s = v_j_s + (S_0 * qd0 + S_1 * qd1 + S_2 * qd2);
                c_app_ang += cross(a0, a1) * (qd0 * qd1);
                c_app_ang += cross(a0, a2) * (qd0 * qd2);
                c_app_ang += cross(a1, a2) * (qd1 * qd2);
            }
            c_app_s = fs_transform_twist(X_sc, spatial(vec3(), c_app_ang));
        }
    } else if (type == JT_BALL) {
        spatial S_0 = fs_transform_twist(X_sc, spatial(vec3(), vec3(1.0f, 0.0f, 0.0f)));
        spatial S_1 = fs_transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 1.0f, 0.0f)));
        spatial S_2 = fs_transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 0.0f, 1.0f)));
        put_S(qd_start + 0, S_0);
        put_S(qd_start + 1, S_1);
        put_S(qd_start + 2, S_2);
        v_j_s = S_0 * qd(qd_start + 0) + S_1 * qd(qd_start + 1) + S_2 * qd(qd_start + 2);
    } else if (type == JT_FREE || type == JT_DISTANCE) {
        v_j_s = fs_transform_twist(X_sc, spatial(vec3(qd(qd_start), qd(qd_start + 1), qd(qd_start + 2)),
                                                 vec3(qd(qd_start + 3), qd(qd_start + 4), qd(qd_start + 5))));
        put_S(qd_start + 0, fs_transform_twist(X_sc, spatial(vec3(1.0f, 0.0f, 0.0f), vec3())));
        put_S(qd_start + 1, fs_transform_twist(X_sc, spatial(vec3(0.0f, 1.0f, 0.0f), vec3())));
        put_S(qd_start + 2, fs_transform_twist(X_sc, spatial(vec3(0.0f, 0.0f, 1.0f), vec3())));
        put_S(qd_start + 3, fs_transform_twist(X_sc, spatial(vec3(), vec3(1.0f, 0.0f, 0.0f))));
        put_S(qd_start + 4, fs_transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 1.0f, 0.0f))));
        put_S(qd_start + 5, fs_transform_twist(X_sc, spatial(vec3(), vec3(0.0f, 0.0f, 1.0f))));
    }
    f.st6(f.F.ft, nb, j, v_j_s);  // parked in the (still unused) subtree-wrench rows until the level pass picks it up
    f.st6(f.F.as, nb, child, c_app_s);  // the level pass adds it to the child's acceleration (zero except multi-angular D6)
}

// (2) the velocity / acceleration recurrence, one tree level at a time (a handful of adds and two cross products);
template <int EPB>
NT_DI void fs_motion_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const int nb = c.a.m.nb;
    const int child = c.T.joint_child[j], parent = c.T.joint_parent[j];
    spatial v_j_s = f.sp6(f.F.ft, nb, j);
    spatial v_parent_s, a_parent_s;
    if (parent >= 0) {
        v_parent_s = f.sp6(f.F.vs, nb, parent);
        a_parent_s = f.sp6(f.F.as, nb, parent);
    }
    spatial v_s = v_parent_s + v_j_s;
    spatial a_s = a_parent_s + fs_spatial_cross(v_s, v_j_s) + f.sp6(f.F.as, nb, child);  // + c_app_s (pre-pass)
    f.st6(f.F.vs, nb, child, v_s);
    f.st6(f.F.as, nb, child, a_s);
}

// (3) bias forces f_b = I a + v x* (I v) minus gravity, and the FK body twist, for all bodies at once.
template <int EPB>
NT_DI void fs_motion_post_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    const int child = c.T.joint_child[j];
    spatial v_s = f.sp6(f.F.vs, nb, child), a_s = f.sp6(f.F.as, nb, child);
    vec3 x_com_s = f.v3(f.F.qcom, 0, nb, child) - f.v3(f.F.org, 0, nb, child);
    float mass = c.pl(c.L.bp, BP_MASS, nb, child);
    vec3 gravity = c.gravity();
    vec3 f_g = mass * gravity;
    spatial f_g_s(f_g, cross(x_com_s, f_g));
    mat66 I_s;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int k = 0; k < 6; ++k) I_s.a[i][k] = c.l(f.F.Is, i * 6 + k, nb, child);
    spatial f_b_s = fs_mul(I_s, a_s) + fs_spatial_cross_dual(v_s, fs_mul(I_s, v_s));
    vec3 omega_world = v_s.bottom;
    vec3 v_com_world = v_s.top + cross(omega_world, x_com_s);
    // body_qd_fk lives in the body_qd rows until the final FK overwrites them with the public output twist
    c.st_lv3(c.L.bqd, 0, nb, child, v_com_world);
    c.st_lv3(c.L.bqd, 3, nb, child, omega_world);
    f.st6(f.F.fs, nb, child, f_b_s - f_g_s);
}

// body_f_ext = state_in.body_f + FREE/DISTANCE joint_f (kernels.py:893-921) + contact wrenches in contact order
template <int EPB>
NT_DI void fs_body_force_item(const FsCtx<EPB>& f, int b, bool forces_are_zero) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    vec3 f0, t0;
    if (!forces_are_zero && c.a.s_in.body_f) {
        f0 = c.gv3(c.a.s_in.body_f, 0, nb, b);
        t0 = c.gv3(c.a.s_in.body_f, 3, nb, b);
    }
    for (int j = 0; j < m.nj; ++j) {
        if (c.T.joint_child[j] != b) continue;  // (one LDS read per joint instead of two; a body has one inbound joint)
        int type = c.T.joint_type[j];
        if (type == JT_FREE || type == JT_DISTANCE) {
            int qs = c.T.joint_qd_start[j];
            f0 += vec3(c.l(c.L.cf, 0, 1, qs), c.l(c.L.cf, 0, 1, qs + 1), c.l(c.L.cf, 0, 1, qs + 2));
            t0 += vec3(c.l(c.L.cf, 0, 1, qs + 3), c.l(c.L.cf, 0, 1, qs + 4), c.l(c.L.cf, 0, 1, qs + 5));
        }
    }
    if (c.a.has_contacts) {
        const int cpp = m.cpp, ncs = m.np * cpp;
        for (int i = c.T.body_pair_start[b]; i < c.T.body_pair_start[b + 1]; ++i) {
            int code = c.T.body_pair_list[i];
            int p = code >> 1, side = code & 1;
            for (int k = 0; k < cpp; ++k) {
                int slot = p * cpp + k;
                const Fld<NC_CW> rec{f.F.cw};  // written by the shared contact phase as slot-major 15-float records
                bool is_a = (side == 0) == (c.l(rec, 14, ncs, slot) != 0.0f);
                if (c.l(rec, is_a ? 12 : 13, ncs, slot) != 0.0f) {
                    vec3 ff = c.lv3(rec, is_a ? 0 : 6, ncs, slot), tt = c.lv3(rec, is_a ? 3 : 9, ncs, slot);
                    if (is_a) { f0 -= ff; t0 -= tt; }
                    else { f0 += ff; t0 += tt; }
                }
            }
        }
    }
    if (c.T.body_flags[b] & BODY_KINEMATIC) { f0 = vec3(0.0f); t0 = vec3(0.0f); }  // zero_kinematic_body_forces (kernels.py:54-63)
    c.st_lv3(f.F.bfx, 0, nb, b, f0);
    c.st_lv3(f.F.bfx, 3, nb, b, t0);
}

// eval_rigid_tau (kernels.py:1320-1419) in two parts.  (1) The subtree wrench of one joint, deepest level first: children of the
// joint's body were processed one level deeper.  (2) fs_tau_dof_item: the projection on the motion subspace + drives, one lane per DOF
// once every wrench is final -- the six dofs of a floating base no longer queue on their joint's lane inside the level sweep.
template <int EPB>
NT_DI void fs_tau_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    const int child = c.T.joint_child[j];
    // body_ft_s[child]: the reference walks joints in descending index and accumulates into the parent
    spatial f_t_s;
    for (int w = f.words - 1; w >= 0; --w) {  // children in descending joint order, straight from the bit mask
        unsigned bits = f.childmask[j * f.words + w];
        while (bits) {
            int hi = 31 - __builtin_clz(bits);
            bits &= ~(1u << hi);
            f_t_s = f_t_s + f.sp6(f.F.ft, nb, w * 32 + hi);
        }
    }
    vec3 force = f.v3(f.F.bfx, 0, nb, child), torque_com = f.v3(f.F.bfx, 3, nb, child);
    vec3 x_com_s = f.v3(f.F.qcom, 0, nb, child) - f.v3(f.F.org, 0, nb, child);
    spatial f_ext(-force, -(torque_com + cross(x_com_s, force)));
    spatial f_s = f.sp6(f.F.fs, nb, child) + f_t_s + f_ext;
    f.st6(f.F.ft, nb, j, f_s);
}
template <int EPB>
NT_DI void fs_tau_dof_item(const FsCtx<EPB>& f, int d) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nb = m.nb, nd = m.nd;
    const int j = f.dof_joint[d];
    const int type = c.T.joint_type[j];
    const int i = d - c.T.joint_qd_start[j];
    const spatial f_s = f.sp6(f.F.ft, nb, j), S = f.sp6(f.F.S, nd, d);
    float sdot = 0.0f;
#pragma unroll
    for (int r = 0; r < 6; ++r) sdot += fs_sget(S, r) * fs_sget(f_s, r);
    if (type == JT_BALL) {
        float passive_f = -c.dof(DP_DAMPING, d) * f.f(f.F.qdi, d);
        f.f(f.F.tau, d) = -sdot + f.f(f.F.jfi, d) + passive_f;
    } else if (type == JT_FREE || type == JT_DISTANCE) {
        f.f(f.F.tau, d) = -sdot + f.f(f.F.jfi, d);
    } else if (type == JT_PRISMATIC || type == JT_REVOLUTE || type == JT_D6) {
        float drive_f = si_dof_force(c, d, c.T.joint_tq_start[j] + i, f.f(f.F.jq, c.T.joint_q_start[j] + i), f.f(f.F.qdi, d));
        f.f(f.F.tau, d) = -sdot + drive_f + f.f(f.F.jfi, d);
    }
}

// P[b][dl] = I_b S_d for every dof d on the path root..joint(b); item = b * W + dl
template <int EPB>
NT_DI void fs_P_item(const FsCtx<EPB>& f, int item) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int W = m.max_art_dofs, nb = m.nb, nd = m.nd;
    const int l = item / W, dl = item - l * W;  // l: joint (== body) index
    const int a = f.art[l];
    const int art_j0 = m.art_start[a];
    const int d0 = c.T.joint_qd_start[art_j0];
    const int d = d0 + dl;
    const int art_j1 = m.art_start[a + 1];
    const int d1 = art_j1 < m.nj ? c.T.joint_qd_start[art_j1] : nd;
    if (d >= d1) return;
    if (!f.on_path(f.dof_joint[d], l)) return;
    const int b = l;  // body l of the articulation == child of joint l (host-checked)
    spatial S = f.sp6(f.F.S, nd, d);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) sum += c.l(f.F.Is, i * 6 + k, nb, b) * fs_sget(S, k);
        c.l(f.F.P, i, nb * W, l * W + dl) = sum;
    }
}
// H[i][jl] (lower triangle, jl <= il): sum over bodies whose path holds both dofs; item = i * W + jl
template <int EPB>
NT_DI void fs_H_item(const FsCtx<EPB>& f, int item) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int W = m.max_art_dofs, nb = m.nb, nd = m.nd;
    const int i = item / W, jl = item - i * W;
    const int ji = f.dof_joint[i];
    const int a = f.art[ji];
    const int art_j0 = m.art_start[a], art_j1 = m.art_start[a + 1];
    const int d0 = c.T.joint_qd_start[art_j0];
    const int il = i - d0;
    if (jl > il) return;
    const int jj = f.dof_joint[d0 + jl];
    spatial S_i = f.sp6(f.F.S, nd, i);
    // bodies off the common path contribute an exact zero (the dense product's J entries are zero there): every P row
    // is fetched unconditionally so the loads pipeline, and the partial dot product is selected away
    float sum = 0.0f;
#pragma unroll 2
    for (int l = ji > jj ? ji : jj; l < art_j1; ++l) {  // a body below both dofs has an index >= both joints
        const bool on = f.on_path(ji, l) && f.on_path(jj, l);
        float p0 = c.l(f.F.P, 0, nb * W, l * W + jl), p1 = c.l(f.F.P, 1, nb * W, l * W + jl), p2 = c.l(f.F.P, 2, nb * W, l * W + jl);
        float p3 = c.l(f.F.P, 3, nb * W, l * W + jl), p4 = c.l(f.F.P, 4, nb * W, l * W + jl), p5 = c.l(f.F.P, 5, nb * W, l * W + jl);
        if (on) {
            sum += S_i.top.x * p0;
            sum += S_i.top.y * p1;
            sum += S_i.top.z * p2;
            sum += S_i.bottom.x * p3;
            sum += S_i.bottom.y * p4;
            sum += S_i.bottom.z * p5;
        }
    }
    c.l(f.F.H, 0, 1, i * W + jl) = sum;
}

// ---- tree-structured mass matrix (nt_featherstone_params.dense_mass_matrix == 0) ---------------------------------------------
// The reference builds H = J^T (M J) over the dense lower triangle and factorises it densely (eval_rigid_mass / dense_cholesky,
// kernels.py:1466-1501,1690-1797).  For a kinematic tree H_ij is non-zero only when dof j lies on dof i's root path, where it is
// S_j^T I^c_i S_i with I^c the composite inertia of the subtree below i; and H = L^T D L with L unit lower triangular has exactly
// H's pattern when the elimination runs from the leaves (Featherstone, Rigid Body Dynamics Algorithms, 6.2 / 6.5).  The same
// joint-space inertia and the same solution up to rounding (the 1e-5 contract), with the work of a 4-legged 18-dof robot cut
// from 171 dense entries x 13 bodies and 18 sequential pivots to 117 entries and 9 dof-tree levels worked by the whole workgroup.

// I^c_l = sum of I_b over the bodies b of joint l's subtree (ascending b); item = l * 6 + row: the six entries of one row of the
// 6 x 6 tile per lane (round 6: one ENTRY per lane was 36 nj items of item decode + mask fetch + a dependent load each -- 15 trips on
// the 32 lanes per environment of the 16-environment tile; a row per lane is 2.4 trips with six independent sums in flight).  Every
// entry is still summed in ascending body order: the same bits.
template <int EPB>
NT_DI void fs_Ic_item(const FsCtx<EPB>& f, int item) {
    const Ctx<EPB>& c = f.c;
    const int nb = c.a.m.nb;
    const int l = item / 6, r0 = (item - l * 6) * 6;
    unsigned long long sub = f.m64(f.t_jbelow, l);
    float sum[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    while (sub) {  // two bodies per round: twelve loads issued together, the additions stay in ascending body order
        const int b0 = __ffsll((long long)sub) - 1;
        sub &= sub - 1;
        const bool two = sub != 0ull;
        const int b1 = two ? __ffsll((long long)sub) - 1 : b0;
        sub &= sub - 1;  // (0 stays 0)
        float v0[6], v1[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { v0[k] = c.l(f.F.Is, r0 + k, nb, b0); v1[k] = c.l(f.F.Is, r0 + k, nb, b1); }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            sum[k] += v0[k];
            if (two) sum[k] += v1[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) c.l(f.F.Ic, r0 + k, nb, l) = sum[k];
}
// Pd[d] = I^c_joint(d) S_d; item = d * 6 + i
template <int EPB>
NT_DI void fs_Pd_item(const FsCtx<EPB>& f, int item) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nb = m.nb, nd = m.nd;
    const int d = item / 6, i = item - d * 6;
    const int l = f.dof_joint[d];
    const spatial S = f.sp6(f.F.S, nd, d);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) sum += c.l(f.F.Ic, i * 6 + k, nb, l) * fs_sget(S, k);
    c.l(f.F.Pd, i, nd, d) = sum;
}
// H[i][j] = S_j^T Pd[i] (+ armature on the diagonal) for the entries (i, j): dof j on dof i's root path; the structural zeros of
// the lower triangle are never stored or read
template <int EPB>
NT_DI void fs_Ht_item(const FsCtx<EPB>& f, int e) {
    const Ctx<EPB>& c = f.c;
    const int nd = c.a.m.nd;
    const int en = f.ent[e], i = en & 255, j = en >> 8;
    const spatial S_j = f.sp6(f.F.S, nd, j);
    float h = 0.0f;
#pragma unroll
    for (int r = 0; r < 6; ++r) h += fs_sget(S_j, r) * c.l(f.F.Pd, r, nd, i);
    if (j == i) {
        h += c.dof(DP_ARMATURE, i);
        if (f.m64(f.t_below, i) == 0ull) h = 1.0f / h;  // a leaf dof: D_i is final as built (fs_solve_tree keeps 1 / D on the diagonal)
    }
    c.l(f.F.H, 0, 1, f.rowbase[i] + j) = h;
}
// L^T D L in place (row i keeps U[i][j] = D_i L[i][j] for the dofs j above i, the diagonal 1 / D_i) and the three substitutions.
// Every dof-tree level is one workgroup phase, leaves first: the rows of the level's dofs are final, so an entry of a shallower row
// collects their updates (bit scan of descendants & level, ascending dof order: the result does not depend on the lane count) and,
// in the same phase, y = L^-T tau collects the level's terms.  The last substitution walks root to leaves on the lanes an
// environment owns in wavefront 0 (tid = env + EPB * slot), ordered by wave-level fences like fs_solve_coop below.
// Called by every thread of the workgroup (barriers inside).
#ifndef FS_WAVE_SYNC_DEFINED
#define FS_WAVE_SYNC_DEFINED
#define FS_WAVE_SYNC()                                        \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)
#endif
template <int EPB>
NT_DI void fs_solve_tree(const FsCtx<EPB>& f, const bool factor) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nd = m.nd, W = m.max_art_dofs, maxd = f.dmaxd, nnz = f.nnz;
    float* lds = c.lds;
    const int e = c.e;
    constexpr int N = Ctx<EPB>::N;
    auto A = [&](int off) -> float& { return lds[(f.F.H + off) * N + e]; };  // off = rowbase[i] + j, j on i's root path
    auto X = [&](int i) -> float& { return lds[(f.F.qdd + i) * N + e]; };
    const bool single = m.na == 1;  // one articulation: row k of the packed lower triangle starts at k (k + 1) / 2 (no table read)
    auto row = [&](int k) { return single ? (k * (k + 1)) / 2 : f.rowbase[k]; };
    // the entries and the dof this lane owns, decoded once per step (the tables are block-shared LDS: a dependent read per level
    // and entry was most of the first version's time)
    constexpr int TE = 4;  // (entries per lane: 117 entries of the quadruped on the 32 lanes per environment of the 16-environment tile)
    const bool cached = nnz <= TE * c.nslot && nd <= c.nslot;
    int e_i[TE], e_j[TE], e_off[TE], e_dd[TE];
    unsigned long long e_below[TE];
#pragma unroll
    for (int u = 0; u < TE; ++u) {
        const int t = c.slot + u * c.nslot;
        const bool on = cached && t < nnz;
        const int en = on ? f.ent[t] : 0;
        e_i[u] = en & 255;
        e_j[u] = en >> 8;
        e_below[u] = on ? f.m64(f.t_below, e_i[u]) : 0ull;
        e_off[u] = row(e_i[u]) + e_j[u];
        e_dd[u] = e_i[u] == e_j[u] ? f.ddepth[e_i[u]] : -2;
    }
    const int my = c.slot < nd ? c.slot : 0;
    const unsigned long long my_below = (cached && c.slot < nd) ? f.m64(f.t_below, my) : 0ull;
    // one update of entry (i, j) [or of y_j when i < 0] by the dofs `ks` of the current level, two dofs per round
    auto collect = [&](float h, unsigned long long ks, int i, int j) {
        while (ks) {
            const int k0 = __ffsll((long long)ks) - 1;
            ks &= ks - 1;
            const bool two = ks != 0ull;
            const int k1 = two ? __ffsll((long long)ks) - 1 : k0;
            ks &= ks - 1;
            const int r0 = row(k0), r1 = row(k1);
            if (i >= 0) {
                const float a0 = A(r0 + i), b0 = A(r0 + j), d0 = A(r0 + k0), a1 = A(r1 + i), b1 = A(r1 + j), d1 = A(r1 + k1);
                h -= a0 * b0 * d0;  // the diagonal of a finished row holds 1 / D_k
                if (two) h -= a1 * b1 * d1;
            } else {
                const float b0 = A(r0 + j), d0 = A(r0 + k0), x0 = X(k0), b1 = A(r1 + j), d1 = A(r1 + k1), x1 = X(k1);
                h -= b0 * (d0 * x0);
                if (two) h -= b1 * (d1 * x1);
            }
        }
        return h;
    };
    if (c.valid)
        for (int i = c.slot; i < nd; i += c.nslot) X(i) = f.f(f.F.tau, i);
    __syncthreads();
    for (int d = maxd; d >= 1; --d) {
        const unsigned long long at = f.m64(f.t_lvl, d);
        // (Measured and rejected, round 6: the runs of single-dof levels -- the six dofs of a FREE root -- worked by ONE lane per environment
        // without the level barriers: factorise + solve 410 k -> 498 k cycles per launch, 69.4 -> 65.2 M env-steps/s; the 35 entries of such a
        // run are 35 dependent LDS round trips on one lane, the level-parallel form spreads them.  profiles/r06K_*.)
        if (c.valid && cached) {
#pragma unroll
            for (int u = 0; u < TE; ++u) {
                const unsigned long long ks = e_below[u] & at;
                if (!factor || !ks) continue;
                float h = collect(A(e_off[u]), ks, e_i[u], e_j[u]);
                if (e_dd[u] == d - 1) h = 1.0f / h;  // the dofs directly above this level have collected every level below: D_i is final
                A(e_off[u]) = h;
            }
            const unsigned long long ks = my_below & at;
            if (ks) X(my) = collect(X(my), ks, -1, my);  // L^T y = tau: y_j -= sum of L[k][j] y_k, L[k][j] = U[k][j] / D_k
        } else if (c.valid) {
            for (int t = c.slot; factor && t < nnz; t += c.nslot) {
                const int en = f.ent[t], i = en & 255, j = en >> 8;
                const unsigned long long ks = f.m64(f.t_below, i) & at;
                if (!ks) continue;
                const int off = row(i) + j;
                float h = collect(A(off), ks, i, j);
                if (i == j && f.ddepth[i] == d - 1) h = 1.0f / h;
                A(off) = h;
            }
            for (int j = c.slot; j < nd; j += c.nslot) {
                const unsigned long long ks = f.m64(f.t_below, j) & at;
                if (ks) X(j) = collect(X(j), ks, -1, j);
            }
        }
        __syncthreads();
    }
    // x_i = (y_i - sum over the dofs j above i of U[i][j] x_j) / D_i, root first, on wavefront 0
    const int G = (64 / Ctx<EPB>::N) < c.nslot ? (64 / Ctx<EPB>::N) : c.nslot;
    if (c.valid && c.slot < G)
        for (int d = 0; d <= maxd; ++d) {
            const unsigned long long at = f.m64(f.t_lvl, d);
            for (int i = c.slot; i < nd; i += G) {
                if (!((at >> i) & 1ull)) continue;
                const int rb = row(i);
                unsigned long long up = f.m64(f.t_above, i);
                float s = X(i);
                while (up) {  // four ancestors per round, ascending
                    int j[4];
                    bool on[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        on[q] = up != 0ull;
                        j[q] = on[q] ? __ffsll((long long)up) - 1 : i;
                        up &= up - 1;
                    }
                    float a[4], x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a[q] = A(rb + j[q]); x[q] = X(j[q]); }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (on[q]) s -= a[q] * x[q];
                }
                X(i) = s * A(rb + i);
            }
            FS_WAVE_SYNC();
        }
}

// dense_cholesky (in place over the lower triangle of H) + dense_subs for one articulation (kernels.py:1690-1797),
// worked by the G = 64 / EPB slot-lanes of the environment that share wavefront 0 (tid = env + EPB * slot): the lanes
// run in lockstep, LDS operations of one wave complete in order, so a value written by one lane is visible to the
// others at the next instruction without a workgroup barrier.  Row i belongs to lane i % G.  Every sum runs in the
// reference's order (k ascending), so the factor and the solution are bit-identical to the serial algorithm.
// compiler-level ordering of LDS traffic between lanes of one wave (no hardware barrier is needed: see below)
// (FS_WAVE_SYNC, defined above fs_solve_tree)

// factor = false: H holds the factor of an earlier step (update_mass_matrix_interval > 1), only the substitutions run
template <int EPB>
NT_DI void fs_solve_coop(const FsCtx<EPB>& f, int a, int lane, int G, const bool factor = true) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int W = m.max_art_dofs, nd = m.nd;
    const int art_j0 = m.art_start[a], art_j1 = m.art_start[a + 1];
    const int d0 = c.T.joint_qd_start[art_j0];
    const int d1 = art_j1 < m.nj ? c.T.joint_qd_start[art_j1] : nd;
    const int n = d1 - d0;
    float* lds = c.lds;
    const int e = c.e;
    auto A = [&](int i, int j) -> float& { return lds[(f.F.H + (d0 + i) * W + j) * Ctx<EPB>::N + e]; };
    auto X = [&](int i) -> float& { return lds[(f.F.qdd + d0 + i) * Ctx<EPB>::N + e]; };
    for (int j = 0; factor && j < n; ++j) {
        float s = A(j, j) + c.dof(DP_ARMATURE, d0 + j);  // every lane evaluates the pivot (no broadcast needed)
        {
            int k = 0;
            for (; k + 4 <= j; k += 4) {  // four LDS reads in flight; the subtraction order stays k ascending
                float r0 = A(j, k), r1 = A(j, k + 1), r2 = A(j, k + 2), r3 = A(j, k + 3);
                s -= r0 * r0;
                s -= r1 * r1;
                s -= r2 * r2;
                s -= r3 * r3;
            }
            for (; k < j; ++k) {
                float r = A(j, k);
                s -= r * r;
            }
        }
        s = sqrtf(s);
        float invS = 1.0f / s;
        FS_WAVE_SYNC();
        if (lane == 0) A(j, j) = s;
        for (int i = j + 1 + lane; i < n; i += G) {
            float t = A(i, j);
            int k = 0;
            for (; k + 4 <= j; k += 4) {
                float a0 = A(i, k), a1 = A(i, k + 1), a2 = A(i, k + 2), a3 = A(i, k + 3);
                float b0 = A(j, k), b1 = A(j, k + 1), b2 = A(j, k + 2), b3 = A(j, k + 3);
                t -= a0 * b0;
                t -= a1 * b1;
                t -= a2 * b2;
                t -= a3 * b3;
            }
            for (; k < j; ++k) t -= A(i, k) * A(j, k);
            A(i, j) = t * invS;
        }
        FS_WAVE_SYNC();
    }
    // forward substitution, column oriented: x_j is final once rows < j have been subtracted (ascending j per row)
    for (int i = lane; i < n; i += G) X(i) = f.f(f.F.tau, d0 + i);
    FS_WAVE_SYNC();
    for (int j = 0; j < n; ++j) {
        float xj = X(j) / A(j, j);
        FS_WAVE_SYNC();
        if (lane == 0) X(j) = xj;
        for (int i = j + 1 + lane; i < n; i += G) X(i) = X(i) - A(i, j) * xj;
        FS_WAVE_SYNC();
    }
    // back substitution: each x_i needs the ascending-j sum over the rows below it; evaluated by every lane
    for (int i = n - 1; i >= 0; --i) {
        float s = X(i);
        {
            int j = i + 1;
            for (; j + 4 <= n; j += 4) {
                float a0 = A(j, i), a1 = A(j + 1, i), a2 = A(j + 2, i), a3 = A(j + 3, i);
                float x0 = X(j), x1 = X(j + 1), x2 = X(j + 2), x3 = X(j + 3);
                s -= a0 * x0;
                s -= a1 * x1;
                s -= a2 * x2;
                s -= a3 * x3;
            }
            for (; j < n; ++j) s -= A(j, i) * X(j);
        }
        float xi = s / A(i, i);
        FS_WAVE_SYNC();
        if (lane == 0) X(i) = xi;
        FS_WAVE_SYNC();
    }
}

// jcalc_integrate (kernels.py:464-630): writes joint_q (in place) and the internal output velocity
template <int EPB>
NT_DI void fs_integrate_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int type = c.T.joint_type[j], child = c.T.joint_child[j];
    const int cs = c.T.joint_q_start[j], ds = c.T.joint_qd_start[j];
    const int lin = c.T.joint_lin_count[j], ang = c.T.joint_ang_count[j];
    const float dt = c.a.dt;
    auto q = [&](int i) -> float& { return f.f(f.F.jq, i); };
    auto qd = [&](int i) { return f.f(f.F.qdi, i); };
    auto qdd = [&](int i) { return f.f(f.F.qdd, i); };
    auto qdn = [&](int i) -> float& { return f.f(f.F.qdo, i); };
    if (type == JT_FIXED) return;
    if (c.T.body_flags[child] & BODY_KINEMATIC) {
        // zero_kinematic_joint_qdd + copy_kinematic_joint_state (kernels.py:1932-1976): the prescribed joint state passes
        // through the solve unchanged (joint_q stays, the internal speeds are copied)
        const int nd_j = (type == JT_FREE || type == JT_DISTANCE) ? 6 : (type == JT_BALL ? 3 : lin + ang);
        for (int i = 0; i < nd_j; ++i) qdn(ds + i) = qd(ds + i);
        return;
    }
    if (type == JT_PRISMATIC || type == JT_REVOLUTE) {
        float qd_new = qd(ds) + qdd(ds) * dt;
        float q_new = q(cs) + qd_new * dt;
        qdn(ds) = qd_new;
        q(cs) = q_new;
        return;
    }
    if (type == JT_BALL) {
        vec3 m_j(qdd(ds), qdd(ds + 1), qdd(ds + 2)), w_j(qd(ds), qd(ds + 1), qd(ds + 2));
        quat r_j(q(cs), q(cs + 1), q(cs + 2), q(cs + 3));
        vec3 w_j_new = w_j + m_j * dt;
        quat drdt_j = quat(w_j_new, 0.0f) * r_j * 0.5f;
        quat r_j_new = normalize(r_j + drdt_j * dt);
        q(cs) = r_j_new.x; q(cs + 1) = r_j_new.y; q(cs + 2) = r_j_new.z; q(cs + 3) = r_j_new.w;
        qdn(ds) = w_j_new.x; qdn(ds + 1) = w_j_new.y; qdn(ds + 2) = w_j_new.z;
        return;
    }
    if ((type == JT_FREE || type == JT_DISTANCE) && c.T.joint_parent[j] >= 0) {
        // descendants stay in the internal parent-origin coordinates during the step (kernels.py:574-611)
        vec3 a_s(qdd(ds), qdd(ds + 1), qdd(ds + 2)), m_s(qdd(ds + 3), qdd(ds + 4), qdd(ds + 5));
        vec3 v_s(qd(ds), qd(ds + 1), qd(ds + 2)), w_s(qd(ds + 3), qd(ds + 4), qd(ds + 5));
        w_s = w_s + m_s * dt;
        v_s = v_s + a_s * dt;
        vec3 p_s(q(cs), q(cs + 1), q(cs + 2));
        vec3 dpdt_s = v_s + cross(w_s, p_s);
        quat r_s(q(cs + 3), q(cs + 4), q(cs + 5), q(cs + 6));
        quat drdt_s = quat(w_s, 0.0f) * r_s * 0.5f;
        vec3 p_s_new = p_s + dpdt_s * dt;
        quat r_s_new = normalize(r_s + drdt_s * dt);
        q(cs) = p_s_new.x; q(cs + 1) = p_s_new.y; q(cs + 2) = p_s_new.z;
        q(cs + 3) = r_s_new.x; q(cs + 4) = r_s_new.y; q(cs + 5) = r_s_new.z; q(cs + 6) = r_s_new.w;
        qdn(ds) = v_s.x; qdn(ds + 1) = v_s.y; qdn(ds + 2) = v_s.z;
        qdn(ds + 3) = w_s.x; qdn(ds + 4) = w_s.y; qdn(ds + 5) = w_s.z;
        return;
    }
    if (type == JT_FREE || type == JT_DISTANCE) {  // root (parent < 0)
        vec3 a_parent(qdd(ds), qdd(ds + 1), qdd(ds + 2)), alpha(qdd(ds + 3), qdd(ds + 4), qdd(ds + 5));
        vec3 v_parent(qd(ds), qd(ds + 1), qd(ds + 2)), omega(qd(ds + 3), qd(ds + 4), qd(ds + 5));
        vec3 p(q(cs), q(cs + 1), q(cs + 2));
        quat r(q(cs + 3), q(cs + 4), q(cs + 5), q(cs + 6));
        vec3 r_com_joint = xform_point(xform_inverse(c.plxf(c.L.jp, 7, m.nj, j)), c.com(child));
        vec3 x_com = p + quat_rotate(r, r_com_joint);
        vec3 v_com = v_parent + cross(omega, x_com);
        vec3 a_com = a_parent + cross(alpha, x_com) + cross(omega, v_com);
        vec3 omega_new = omega + alpha * dt;
        vec3 v_com_new = v_com + a_com * dt;
        quat drdt = quat(omega_new, 0.0f) * r * 0.5f;
        quat r_new = normalize(r + drdt * dt);
        vec3 x_com_new = x_com + v_com_new * dt;
        vec3 p_new = x_com_new - quat_rotate(r_new, r_com_joint);
        vec3 v_parent_new = v_com_new - cross(omega_new, x_com_new);
        q(cs) = p_new.x; q(cs + 1) = p_new.y; q(cs + 2) = p_new.z;
        q(cs + 3) = r_new.x; q(cs + 4) = r_new.y; q(cs + 5) = r_new.z; q(cs + 6) = r_new.w;
        qdn(ds) = v_parent_new.x; qdn(ds + 1) = v_parent_new.y; qdn(ds + 2) = v_parent_new.z;
        qdn(ds + 3) = omega_new.x; qdn(ds + 4) = omega_new.y; qdn(ds + 5) = omega_new.z;
        return;
    }
    if (type == JT_D6) {
        for (int i = 0; i < lin + ang; ++i) {
            float qd_new = qd(ds + i) + qdd(ds + i) * dt;
            float q_new = q(cs + i) + qd_new * dt;
            qdn(ds + i) = qd_new;
            q(cs + i) = q_new;
        }
    }
}

// eval_single_articulation_fk_with_velocity_conversion for one joint (kernels.py:1987-2150): final body_q / body_qd
// PUBLIC = true: newton.eval_fk semantics (eval_single_articulation_fk, newton/_src/sim/articulation.py:236-420): FREE /
// DISTANCE joint_qd is the child's COM twist in the parent anchor frame.
template <int EPB, bool PUBLIC>
NT_DI void fs_fk_vel_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int nb = m.nb;
    const int parent = c.T.joint_parent[j], child = c.T.joint_child[j], type = c.T.joint_type[j];
    const int qs = c.T.joint_qd_start[j];
    const int lin = c.T.joint_lin_count[j], ang = c.T.joint_ang_count[j];
    auto qd = [&](int i) { return f.f(f.F.qdo, i); };
    xform X_j = c.lxf(f.F.vs, 0, m.nj, j);  // fs_joint_xform_item
    spatial v_j;
    if (type == JT_PRISMATIC) v_j = spatial(c.dof_axis(qs) * qd(qs), vec3());
    if (type == JT_REVOLUTE) v_j = spatial(vec3(), c.dof_axis(qs) * qd(qs));
    if (type == JT_BALL) v_j = spatial(vec3(), vec3(qd(qs), qd(qs + 1), qd(qs + 2)));
    if (type == JT_FREE || type == JT_DISTANCE)
        v_j = spatial(vec3(qd(qs), qd(qs + 1), qd(qs + 2)), vec3(qd(qs + 3), qd(qs + 4), qd(qs + 5)));
    if (type == JT_D6) {
        vec3 vel_v(0.0f), vel_w(0.0f);
        for (int k = 0; k < 3; ++k)
            if (lin > k) vel_v += c.dof_axis(qs + k) * qd(qs + k);
        if (ang == 1) vel_w = qd(qs + lin) * c.dof_axis(qs + lin);
        {
            if (ang >= 2) {
                const int cs = c.T.joint_q_start[j];
                vec3 a0, a1, a2;
                d6_multi_angular(ang, c.dof_axis(qs + lin), c.dof_axis(qs + lin + 1), ang == 3 ? c.dof_axis(qs + lin + 2) : vec3(),
                                 f.f(f.F.jq, cs + lin), f.f(f.F.jq, cs + lin + 1), ang == 3 ? f.f(f.F.jq, cs + lin + 2) : 0.0f, a0,
                                 a1, a2);
                vel_w = a0 * qd(qs + lin) + a1 * qd(qs + lin + 1);
                if (ang == 3) vel_w = vel_w + a2 * qd(qs + lin + 2);
            }
        }
        v_j = spatial(vel_v, vel_w);
    }
    xform X_wpj = c.plxf(c.L.jp, 0, m.nj, j);
    xform X_wp;
    if (parent >= 0) {
        X_wp = c.body_q(parent);
        X_wpj = X_wp * X_wpj;
    }
    xform X_wcj = X_wpj * X_j;
    xform X_wc = X_wcj * xform_inverse(c.plxf(c.L.jp, 7, m.nj, j));
    vec3 x_child_origin = X_wc.p;
    vec3 v_parent_origin, w_parent;
    if (parent >= 0) {
        w_parent = c.body_w(parent);
        vec3 r = x_child_origin - xform_point(X_wp, c.com(parent));
        v_parent_origin = cross(w_parent, r) + c.body_v(parent);
    }
    vec3 linear_joint_world = xform_vector(X_wpj, v_j.top);
    vec3 angular_joint_world = xform_vector(X_wpj, v_j.bottom);
    vec3 linear_joint_origin;
    if (type == JT_FREE || type == JT_DISTANCE) {
        if (PUBLIC) {
            // com_twist_to_origin_twist (articulation.py:29-33)
            linear_joint_origin = linear_joint_world - cross(angular_joint_world, xform_vector(X_wc, c.com(child)));
        } else {
            spatial v_j_world = fs_transform_twist(X_wpj, v_j);
            linear_joint_origin = cross(v_j_world.bottom, x_child_origin) + v_j_world.top;
            angular_joint_world = v_j_world.bottom;
        }
    } else {
        vec3 child_origin_offset_world = x_child_origin - X_wcj.p;
        linear_joint_origin = linear_joint_world + cross(angular_joint_world, child_origin_offset_world);
    }
    vec3 v_origin = v_parent_origin + linear_joint_origin, w = w_parent + angular_joint_world;
    const vec3 r_com = xform_vector(X_wc, c.com(child));
    c.st_lxf(c.L.bq, nb, child, X_wc);
    c.st_lv3(c.L.bqd, 0, nb, child, cross(w, r_com) + v_origin);
    c.st_lv3(c.L.bqd, 3, nb, child, w);
    // the COM world position fs_fk_item would store for these poses (X_wc * (com, 1)).p: the next substep of a rollout starts
    // from this pass instead of repeating eval_rigid_fk on the same joint_q (fs_substep, fk_is_current)
    if (!PUBLIC) c.st_lv3(f.F.qcom, 0, nb, child, X_wc.p + r_com);
}

// convert_free_distance_joint_qd_internal_to_public (kernels.py:1015-1066) straight into state_out.joint_qd
template <int EPB>
NT_DI void fs_to_public_item(const FsCtx<EPB>& f, int j) {
    const Ctx<EPB>& c = f.c;
    const nt_model& m = c.a.m;
    const int qs = c.T.joint_qd_start[j], type = c.T.joint_type[j];
    const int qe = j + 1 < m.nj ? c.T.joint_qd_start[j + 1] : m.nd;
    auto out = [&](int i) -> float& { return f.f(f.F.qdp, i); };
    if (type != JT_FREE && type != JT_DISTANCE) {
        for (int i = qs; i < qe; ++i) out(i) = f.f(f.F.qdo, i);
        return;
    }
    vec3 r = fs_free_com_offset(f, j);
    vec3 v_int(f.f(f.F.qdo, qs), f.f(f.F.qdo, qs + 1), f.f(f.F.qdo, qs + 2));
    vec3 omega(f.f(f.F.qdo, qs + 3), f.f(f.F.qdo, qs + 4), f.f(f.F.qdo, qs + 5));
    vec3 v_com = v_int + cross(omega, r);
    out(qs) = v_com.x; out(qs + 1) = v_com.y; out(qs + 2) = v_com.z;
    out(qs + 3) = omega.x; out(qs + 4) = omega.y; out(qs + 5) = omega.z;
}

// block-shared tree tables (joint ancestor / depth / articulation, joint of each dof, root-path bit masks)
template <int EPB>
NT_DI void fs_build_tables(const Ctx<EPB>& c, int* extra) {
    const nt_model& m = c.a.m;
    const int nj = m.nj;
    for (int j = threadIdx.x; j < nj; j += blockDim.x) {
        int p = c.T.joint_parent[j], anc = -1;
        if (p >= 0)
            for (int k = 0; k < nj; ++k)
                if (c.T.joint_child[k] == p) anc = k;
        extra[j] = anc;
        int art = 0;
        for (int k = 0; k < m.na; ++k)
            if (m.art_start[k] <= j) art = k;
        extra[2 * nj + j] = art;
    }
    __syncthreads();
    const int words = fs_mask_words(m);
    for (int j = threadIdx.x; j < nj; j += blockDim.x) {
        int d = 0, k = extra[j];
        unsigned* mask = reinterpret_cast<unsigned*>(extra + 3 * nj + m.nd) + j * words;
        for (int w = 0; w < words; ++w) mask[w] = 0u;
        mask[j >> 5] |= 1u << (j & 31);
        while (k >= 0) {
            d += 1;
            mask[k >> 5] |= 1u << (k & 31);
            k = extra[k];
        }
        extra[nj + j] = d;
    }
    for (int d = threadIdx.x; d < m.nd; d += blockDim.x) {
        int j = 0;
        for (int k = 1; k < nj; ++k)
            if (c.T.joint_qd_start[k] <= d) j = k;
        extra[3 * nj + d] = j;
    }
    for (int j = threadIdx.x; j < nj; j += blockDim.x) {
        unsigned* cm = reinterpret_cast<unsigned*>(extra + 3 * nj + m.nd) + nj * words + j * words;
        for (int w = 0; w < words; ++w) cm[w] = 0u;
        for (int k = 0; k < nj; ++k)
            if (extra[k] == j) cm[k >> 5] |= 1u << (k & 31);
    }
    {   // descendant_free_distance_refresh_joint_starts (solver_featherstone.py:243-262) as a per-joint flag
        int* refresh = extra + 3 * nj + m.nd + 2 * nj * words;
        auto desc = [&](int k) {
            const int t = c.T.joint_type[k];
            return (t == JT_FREE || t == JT_DISTANCE) && c.T.joint_parent[k] >= 0 && !(c.T.body_flags[c.T.joint_child[k]] & BODY_KINEMATIC);
        };
        for (int j = threadIdx.x; j <= nj; j += blockDim.x) {
            int flag = 0;
            if (j < nj) {
                int a0 = 0, a1 = nj;
                for (int k = 0; k < m.na; ++k)
                    if (m.art_start[k] <= j) { a0 = m.art_start[k]; a1 = k + 1 < m.na ? m.art_start[k + 1] : nj; }
                for (int k = a0; k < a1 && k <= j; ++k) flag |= desc(k) ? 1 : 0;
            } else {
                for (int k = 0; k < nj; ++k) flag |= desc(k) ? 1 : 0;
            }
            refresh[j] = flag;
        }
    }
    __syncthreads();
    if (fs_tree_ok(m)) {  // the dof tree: a joint's dofs form a chain, its first dof hangs below the last dof of the nearest ancestor
                          // joint that has any
        const int nd = m.nd, W = m.max_art_dofs;
        unsigned* below = reinterpret_cast<unsigned*>(extra + fs_topo_base_ints(m));
        unsigned *above = below + 2 * nd, *lvl = above + 2 * nd, *jbelow = lvl + 2 * (nd + 1);
        int *ddepth = reinterpret_cast<int*>(jbelow + 2 * nj), *rowbase = ddepth + nd, *ent = rowbase + nd + 2;
        int* dpar = ent;  // scratch until the entries are written
        const int* dof_joint = extra + 3 * nj;
        auto qd_end = [&](int k) { return k + 1 < nj ? c.T.joint_qd_start[k + 1] : nd; };
        auto put64 = [](unsigned* p, int i, unsigned long long v) { p[2 * i] = (unsigned)v; p[2 * i + 1] = (unsigned)(v >> 32); };
        auto get64 = [](const unsigned* p, int i) { return (unsigned long long)p[2 * i] | ((unsigned long long)p[2 * i + 1] << 32); };
        for (int d = threadIdx.x; d < nd; d += blockDim.x) {
            const int j = dof_joint[d];
            int p = d - 1;
            if (d == c.T.joint_qd_start[j]) {
                int k = extra[j];
                while (k >= 0 && qd_end(k) == c.T.joint_qd_start[k]) k = extra[k];
                p = k >= 0 ? qd_end(k) - 1 : -1;
            }
            dpar[d] = p;
        }
        __syncthreads();
        for (int d = threadIdx.x; d < nd; d += blockDim.x) {
            unsigned long long up = 0ull;
            int depth = 0;
            for (int k = dpar[d]; k >= 0; k = dpar[k]) {
                depth += 1;
                up |= 1ull << k;
            }
            put64(above, d, up);
            ddepth[d] = depth;
        }
        __syncthreads();
        for (int d = threadIdx.x; d < nd; d += blockDim.x) {
            unsigned long long down = 0ull;
            for (int k = 0; k < nd; ++k)
                if ((get64(above, k) >> d) & 1ull) down |= 1ull << k;
            put64(below, d, down);
            // row d of the articulation's H, packed lower triangle: local row dl starts at dl (dl + 1) / 2 behind the triangles of the
            // articulations in front, minus the articulation's first dof so that rowbase[d] + j addresses column j (an absolute dof)
            {
                const int art = extra[2 * nj + dof_joint[d]];
                int base = 0;
                for (int k = 0; k < art; ++k) {
                    const int j0 = m.art_start[k], j1 = m.art_start[k + 1];
                    const int n_k = (j1 < nj ? c.T.joint_qd_start[j1] : nd) - c.T.joint_qd_start[j0];
                    base += n_k * (n_k + 1) / 2;
                }
                const int d0 = c.T.joint_qd_start[m.art_start[art]], dl = d - d0;
                rowbase[d] = base + dl * (dl + 1) / 2 - d0;
            }
        }
        for (int l = threadIdx.x; l <= nd; l += blockDim.x) {
            unsigned long long at = 0ull;
            for (int k = 0; k < nd; ++k)
                if (ddepth[k] == l) at |= 1ull << k;
            put64(lvl, l, at);
        }
        for (int j = threadIdx.x; j < nj; j += blockDim.x) {
            const unsigned* pm = reinterpret_cast<const unsigned*>(extra + 3 * nj + m.nd);
            const int words = fs_mask_words(m);
            unsigned long long sub = 0ull;
            for (int b = 0; b < nj; ++b)
                if ((pm[b * words + (j >> 5)] >> (j & 31)) & 1u) sub |= 1ull << b;
            put64(jbelow, j, sub);
        }
        __syncthreads();  // (dpar is dead from here on: the entries overwrite it)
        if (threadIdx.x == 0) {
            int mx = 0, cnt = 0;
            for (int k = 0; k < nd; ++k) {
                mx = ddepth[k] > mx ? ddepth[k] : mx;
                cnt += ddepth[k] + 1;
            }
            rowbase[nd] = cnt;
            rowbase[nd + 1] = mx;
        }
        for (int i = threadIdx.x; i < nd; i += blockDim.x) {
            int off = 0;
            for (int k = 0; k < i; ++k) off += ddepth[k] + 1;
            unsigned long long up = get64(above, i);
            while (up) {
                const int j = __ffsll((long long)up) - 1;
                up &= up - 1;
                ent[off++] = i | (j << 8);
            }
            ent[off] = i | (i << 8);
        }
        __syncthreads();
    }
}

// update_mass: rebuild P / H and refactorise (else the factor comes back from nt_featherstone_params.mass_matrix_cache)
template <int EPB>
NT_DI bool fs_update_mass(const Ctx<EPB>& c, int substep) {
    const nt_featherstone_params& p = c.a.fp;
    if (!p.mass_matrix_cache || p.update_mass_matrix_interval <= 1) return true;
    if (p.force_update && substep == 0) return true;
    return ((p.step_index + substep) % p.update_mass_matrix_interval) == 0;
}
// One SolverFeatherstone.step on the state resident in LDS (joint_q in F.jq, public joint_qd in F.qdp).
// ROLLOUT: called by the fused rollout, whose collide phases ran a barrier ago in the same kernel (si_contact_item<FUSED>)
template <int EPB, bool ROLLOUT = false>
NT_DI void fs_substep(const Ctx<EPB>& c, const FsCtx<EPB>& f, const FsLayout& F, int max_depth, bool forces_are_zero,
                      bool publish_fk, float* parent_f_out, const int substep = 0) {
    const KArgs& a = c.a;
    const nt_model& m = a.m;
    const int nj = m.nj, nb = m.nb;
    NT_SKIP_DECL(a);  // timing ablation builds only (-DNT_ABLATION): results are meaningless when set
    NT_TICK(10);
    // eval_rigid_fk: joint transforms for all joints at once, then level by level (a joint's parent body is final one
    // level earlier).  Substeps >= 1 of a rollout skip it: body_q and the COM positions in LDS are the previous substep's FK with
    // velocity conversion of the same joint_q -- the same expressions on the same inputs, bit for bit (rollout == loop stays a
    // bitwise test) -- unless that substep ended with the descendant FREE / DISTANCE pose correction, which rewrites poses.
    const bool fk_is_current = substep > 0 && !publish_fk && !f.any_descendant_free();  // block-uniform
    if (!fk_is_current) {
        if (c.valid && !NT_SKIP(1))
            for (int j = c.slot; j < nj; j += c.nslot) fs_joint_xform_item(f, j);
        __syncthreads();
        for (int lvl = 0; lvl <= max_depth; ++lvl) {
            if (c.valid && !NT_SKIP(1))
                for (int j = c.slot; j < nj; j += c.nslot)
                    if (f.depth[j] == lvl) fs_fk_item(f, j);
            __syncthreads();
        }
    }
    NT_TICK(11);
    // state_in.body_q is refreshed by the reference step (solver_featherstone.py:492-514): publish it when distinct
    if (publish_fk && c.valid && a.s_in.body_q != a.s_out.body_q) unstage_rows(c, c.L.bq, a.s_in.body_q, 7, nb);
    // public -> internal velocities, then eval_rigid_id's per-joint preparation in the same barrier interval: a joint's lane reads
    // the internal speeds of its own dofs only.  With 2 nj slot lanes the spatial-inertia half runs on lanes [nj, 2 nj)
    NT_TICK(12);
    if (c.valid) {
        const bool split = c.nslot >= 2 * nj;
        for (int j = c.slot; j < nj; j += c.nslot) {
            fs_to_internal_item(f, j);
            if (!NT_SKIP(2)) fs_motion_pre_item(f, j, true, !split);
        }
        if (split && c.slot >= nj && c.slot < 2 * nj && !NT_SKIP(2)) fs_motion_pre_item(f, c.slot - nj, false, true);
    }
    __syncthreads();
    for (int lvl = 0; lvl <= max_depth; ++lvl) {
        if (c.valid && !NT_SKIP(2))
            for (int j = c.slot; j < nj; j += c.nslot)
                if (f.depth[j] == lvl) fs_motion_item(f, j);
        __syncthreads();
    }
    if (c.valid && !NT_SKIP(2))
        for (int j = c.slot; j < nj; j += c.nslot) fs_motion_post_item(f, j);
    __syncthreads();
    NT_TICK(13);
    // eval_body_contact on (body_q, body_qd_fk)
    if (a.has_contacts) {
        Ctx<EPB> cc = c;
        cc.L.si_cw.off = F.cw;
        if (c.valid && !NT_SKIP(4))
            for (int s = c.slot; s < m.np * m.cpp; s += c.nslot) si_contact_item<EPB, ROLLOUT>(cc, s);
        __syncthreads();
    }
    if (c.valid)
        for (int b = c.slot; b < nb; b += c.nslot) fs_body_force_item(f, b, forces_are_zero);
    __syncthreads();
    NT_TICK(14);
    // eval_rigid_tau, deepest level first
    for (int lvl = max_depth; lvl >= 0; --lvl) {
        if (c.valid && !NT_SKIP(8))
            for (int j = c.slot; j < nj; j += c.nslot)
                if (f.depth[j] == lvl) fs_tau_item(f, j);
        __syncthreads();
    }
    NT_TICK(15);
    // compute_body_parent_f (kernels.py:2371-2416): f_s of the body's inbound joint (left in `ft` by the pass above),
    // moved from the solve origin to the body COM
    if (parent_f_out && c.valid)
        for (int b = c.slot; b < nb; b += c.nslot) {
            spatial f_s = f.sp6(F.ft, nb, b);
            vec3 r_com = f.v3(F.qcom, 0, nb, b) - f.v3(F.org, 0, nb, b);
            vec3 t = f_s.bottom - cross(r_com, f_s.top);
            parent_f_out[c.g(0, nb, b)] = f_s.top.x; parent_f_out[c.g(1, nb, b)] = f_s.top.y; parent_f_out[c.g(2, nb, b)] = f_s.top.z;
            parent_f_out[c.g(3, nb, b)] = t.x; parent_f_out[c.g(4, nb, b)] = t.y; parent_f_out[c.g(5, nb, b)] = t.z;
        }
    // P = M J (non-zero blocks), H = J^T P (lower triangle), Cholesky, solve
    const int W = m.max_art_dofs;
    const bool update_mass = fs_update_mass(c, substep);
    float* cache = a.fp.mass_matrix_cache;
    const bool tree = a.fp.dense_mass_matrix == 0 && f.tree_ok;  // block-uniform
    // eval_rigid_tau, part 2: one lane per dof, in the barrier interval of the mass-matrix build (tau is first read by the solve)
    if (c.valid && !NT_SKIP(8))
        for (int d = c.slot; d < m.nd; d += c.nslot) fs_tau_dof_item(f, d);
    if (update_mass && tree) {
        if (c.valid && !NT_SKIP(16))
            for (int i = c.slot; i < nj * 6; i += c.nslot) fs_Ic_item(f, i);
        __syncthreads();
        if (c.valid && !NT_SKIP(16))
            for (int i = c.slot; i < m.nd * 6; i += c.nslot) fs_Pd_item(f, i);
        __syncthreads();
        NT_TICK(16);
        if (c.valid && !NT_SKIP(32))
            for (int i = c.slot; i < f.nnz; i += c.nslot) fs_Ht_item(f, i);
    } else if (update_mass) {
        if (c.valid && !NT_SKIP(16))
            for (int i = c.slot; i < nj * W; i += c.nslot) fs_P_item(f, i);
        __syncthreads();
        NT_TICK(16);
        if (c.valid && !NT_SKIP(32))
            for (int i = c.slot; i < m.nd * W; i += c.nslot) fs_H_item(f, i);
    } else if (c.valid) {  // the factor of the last rebuild
        for (int r = c.slot; r < (tree ? fs_tree_nnz_bound(m) : m.nd * W); r += c.nslot) f.f(F.H, r) = cache[(size_t)r * c.ES + c.env];
    }
    __syncthreads();
    NT_TICK(17);
    if (tree) {
        fs_solve_tree(f, update_mass);
    } else {
        const int G = (64 / Ctx<EPB>::N) < c.nslot ? (64 / Ctx<EPB>::N) : c.nslot;
        if (c.valid && !NT_SKIP(64) && c.slot < G)
            for (int k = 0; k < m.na; ++k) fs_solve_coop(f, k, c.slot, G, update_mass);
    }
    __syncthreads();
    if (update_mass && cache && a.fp.update_mass_matrix_interval > 1 && c.valid)
        for (int r = c.slot; r < (tree ? fs_tree_nnz_bound(m) : m.nd * W); r += c.nslot) cache[(size_t)r * c.ES + c.env] = f.f(F.H, r);
    NT_TICK(18);
    // integrate_generalized_joints
    const bool desc_free = f.any_descendant_free();  // block-uniform
    const Fld<7> bq_prev{F.fs};                      // f_b - f_g and the subtree wrenches (12 nb rows) are dead by now
    if (c.valid) {
        for (int j = c.slot; j < nj; j += c.nslot) {
            fs_integrate_item(f, j);
            if (!NT_SKIP(128)) fs_joint_xform_item(f, j);  // the joint transform of the new joint_q (this lane wrote it): no barrier between
        }
        if (desc_free)  // descendant_body_q_prev (solver_featherstone.py:481,514-516): the poses of the start-of-step FK
            for (int r = c.slot; r < 7 * nb; r += c.nslot) c.lds[(bq_prev.off + r) * Ctx<EPB>::N + c.e] = c.lds[(c.L.bq.off + r) * Ctx<EPB>::N + c.e];
    }
    __syncthreads();
    NT_TICK(19);
    // FK with velocity conversion -> public body_q / body_qd; without descendant FREE / DISTANCE joints (their pose correction comes
    // first) a joint's lane converts its velocities to the public convention right behind its FK item: the poses it reads -- its
    // child's and its parent's -- are final then
    for (int lvl = 0; lvl <= max_depth; ++lvl) {
        if (c.valid && !NT_SKIP(128))
            for (int j = c.slot; j < nj; j += c.nslot)
                if (f.depth[j] == lvl) {
                    fs_fk_vel_item<EPB, false>(f, j);
                    if (!desc_free) fs_to_public_item(f, j);
                }
        __syncthreads();
    }
    if (desc_free) {  // solver_featherstone.py:1006-1046
        // correct_free_distance_body_pose_from_world_twist (kernels.py:1897-1929)
        if (c.valid)
            for (int j = c.slot; j < nj; j += c.nslot)
                if (f.descendant_free(j)) {
                    const int child = c.T.joint_child[j];
                    const xform X_wb = c.lxf(bq_prev, 0, nb, child);
                    const vec3 com = c.com(child);
                    const vec3 x_com = xform_point(X_wb, com);
                    const vec3 v_com = c.body_v(child), w = c.body_w(child);
                    const quat drdt = quat(w, 0.0f) * X_wb.q * 0.5f;
                    const quat q_new = normalize(X_wb.q + drdt * c.a.dt);
                    const vec3 x_com_new = x_com + v_com * c.a.dt;
                    c.st_lxf(c.L.bq, nb, child, xform(x_com_new - quat_rotate(q_new, com), q_new));
                }
        __syncthreads();
        // reconstruct_free_distance_joint_q_from_body_pose (kernels.py:978-1012), then that joint's transform again
        if (c.valid)
            for (int j = c.slot; j < nj; j += c.nslot)
                if (f.descendant_free(j)) {
                    const int parent = c.T.joint_parent[j], child = c.T.joint_child[j], cs = c.T.joint_q_start[j];
                    const xform X_wpj = c.body_q(parent) * c.plxf(c.L.jp, 0, nj, j);
                    const xform X_wcj = c.body_q(child) * c.plxf(c.L.jp, 7, nj, j);
                    const vec3 x_err_c = quat_rotate_inv(X_wpj.q, X_wcj.p - X_wpj.p);
                    const quat q_pc = quat_inverse(X_wpj.q) * X_wcj.q;
                    f.f(F.jq, cs) = x_err_c.x; f.f(F.jq, cs + 1) = x_err_c.y; f.f(F.jq, cs + 2) = x_err_c.z;
                    f.f(F.jq, cs + 3) = q_pc.x; f.f(F.jq, cs + 4) = q_pc.y; f.f(F.jq, cs + 5) = q_pc.z; f.f(F.jq, cs + 6) = q_pc.w;
                    fs_joint_xform_item(f, j);
                }
        __syncthreads();
        // eval_fk_with_velocity_conversion_from_joint_starts (kernels.py:2332-2368): from the first such joint of the articulation on
        for (int lvl = 1; lvl <= max_depth; ++lvl) {
            if (c.valid)
                for (int j = c.slot; j < nj; j += c.nslot)
                    if (f.depth[j] == lvl && f.refresh[j]) fs_fk_vel_item<EPB, false>(f, j);
            __syncthreads();
        }
    }
    NT_TICK(20);
    if (desc_free) {
        if (c.valid)
            for (int j = c.slot; j < nj; j += c.nslot) fs_to_public_item(f, j);
        __syncthreads();
    }
    NT_TICK(21);
}

