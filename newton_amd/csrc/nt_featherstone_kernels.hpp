// nt_featherstone_kernels.hpp -- SolverFeatherstone's step / rollout kernels and eval_fk, namespace ieee like the phases of
// nt_featherstone.hpp they run.  (Measured, profiles/r04h_*: the phases compiled in namespace fused -- contraction + fast division --
// pass the whole GPU suite but C3 does not move, 25.6 vs 26.7 M env-steps/s: the solver waits on LDS round trips of H / L and on the
// level-synchronous barriers, not on arithmetic.  The literal operation order stays.)
#pragma once

// the LDS layout of a launch (host twin: fs_launch): persistent block of the XPBD layout without its body-derived tile, per-environment
// parameters or ONE block-shared copy (tile code EPB with NT_UNI), dense or tree-structured mass-matrix region (fs_tree_mode)
template <int EPB>
NT_DI FsLayout fs_kernel_layout(const KArgs& a) {
    return make_fs_layout(a.m, make_layout(a.m, false, false, Ctx<EPB>::UNI, false), fs_tree_mode(a));
}
// workgroup memory: [F.rows x N floats][topology ints][Featherstone ints][uniform-parameter floats]
template <int EPB>
NT_DI int* fs_place_tables(Ctx<EPB>& c, float* lds, const FsLayout& F) {
    int* extra = reinterpret_cast<int*>(lds + (size_t)F.rows * Ctx<EPB>::N) + topo_ints(c.a.m);
    c.up = reinterpret_cast<float*>(extra + fs_topo_ints(c.a.m));  // (Ctx put it behind the topology ints, where `extra` lives here)
    return extra;
}

template <int EPB>
__global__ void __launch_bounds__(256) featherstone_step_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    const nt_model& m = a.m;
    const FsLayout F = fs_kernel_layout<EPB>(a);
    Ctx<EPB> c(a, lds, F.rows);  // topology ints are staged behind the Featherstone rows
    int* extra = fs_place_tables(c, lds, F);
    __syncthreads();
    fs_build_tables(c, extra);
    FsCtx<EPB> f(c, extra);
    int max_depth = 0;
    for (int j = 0; j < m.nj; ++j) max_depth = imax(max_depth, f.depth[j]);
    load_params(c, true);
    if (c.valid) {
        stage_rows(c, F.jq, a.s_in.joint_q, m.nc);
        stage_rows(c, F.qdp, a.s_in.joint_qd, m.nd);
    }
    __syncthreads();
    fs_substep(c, f, F, max_depth, false, true, a.s_out.body_parent_f);
    if (c.valid) {
        unstage_rows(c, F.jq, a.s_out.joint_q, m.nc);
        unstage_rows(c, F.qdp, a.s_out.joint_qd, m.nd);
    }
    store_state(c, a.s_out);
}

// substeps x { clear_forces; CollisionPipeline.collide; SolverFeatherstone.step; swap } in one launch: generalized and
// maximal state, parameters and all Featherstone intermediates stay in LDS; only the contacts touch HBM per substep.
// The result lands in s_in (= s0) for an even number of substeps and in s_out (= s1) for an odd one.
// THREADS: workgroup size (512 for the 16-environment uniform-parameter tile: 32 lanes per environment)
template <int EPB, bool CVX, int THREADS = 256>
__global__ void __launch_bounds__(THREADS) featherstone_rollout_kernel(KArgs a) {
    extern __shared__ __align__(16) float lds[];
    const nt_model& m = a.m;
    const FsLayout F = fs_kernel_layout<EPB>(a);
    Ctx<EPB> c(a, lds, F.rows);
    int* extra = fs_place_tables(c, lds, F);
    __syncthreads();
    fs_build_tables(c, extra);
    FsCtx<EPB> f(c, extra);
    int max_depth = 0;
    for (int j = 0; j < m.nj; ++j) max_depth = imax(max_depth, f.depth[j]);
    load_state(c, a.s_in);
    load_params(c, true);
    if (c.valid) {
        stage_rows(c, F.jq, a.s_in.joint_q, m.nc);
        stage_rows(c, F.qdp, a.s_in.joint_qd, m.nd);
        for (int r = c.slot; r < 6 * m.nb; r += c.nslot) {
            a.s_in.body_f[(size_t)r * c.ES + c.env] = 0.0f;
            a.s_out.body_f[(size_t)r * c.ES + c.env] = 0.0f;
        }
    }
    __syncthreads();
    // the collide phases use the (dead at that point) P / H / contact-wrench union as their scratch
    Ctx<EPB> cc = c;
    place_collide_scratch(cc.L, m, F.cw, false);
    stage_global_world(cc);  // (static shapes: transform + AABB once per launch; the first pair phase is a barrier away)
    cc.gworld_ready = true;
    const nt_state& res = (a.substeps & 1) ? a.s_out : a.s_in;
    for (int s = 0; s < a.substeps; ++s) {
        do_collide<EPB, CVX>(cc, s == a.substeps - 1);
        fs_substep<EPB, true>(c, f, F, max_depth, true, false, s == a.substeps - 1 ? res.body_parent_f : nullptr, s);
    }
    if (c.valid) {
        unstage_rows(c, F.jq, res.joint_q, m.nc);
        unstage_rows(c, F.qdp, res.joint_qd, m.nd);
    }
    store_state(c, res);
}

// newton.eval_fk(model, joint_q, joint_qd, state) (newton/_src/sim/articulation.py:423-573): body_q / body_qd from
// generalized coordinates, all articulations, level by level.
template <int EPB>
__global__ void __launch_bounds__(256) eval_fk_kernel(KArgs a, const float* joint_q, const float* joint_qd) {
    extern __shared__ __align__(16) float lds[];
    const nt_model& m = a.m;
    const int nj = m.nj;
    const FsLayout F = fs_kernel_layout<EPB>(a);
    Ctx<EPB> c(a, lds, F.rows);
    int* extra = fs_place_tables(c, lds, F);
    __syncthreads();
    for (int j = threadIdx.x; j < nj; j += blockDim.x) {
        int p = c.T.joint_parent[j], anc = -1;
        if (p >= 0)
            for (int k = 0; k < nj; ++k)
                if (c.T.joint_child[k] == p) anc = k;
        extra[j] = anc;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nj; j += blockDim.x) {
        int d = 0, k = extra[j];
        while (k >= 0) { d += 1; k = extra[k]; }
        extra[nj + j] = d;
    }
    __syncthreads();
    FsCtx<EPB> f(c, extra);
    int max_depth = 0;
    for (int j = 0; j < nj; ++j) max_depth = imax(max_depth, f.depth[j]);
    load_params(c, false);
    if (c.valid) {
        stage_rows(c, F.jq, joint_q, m.nc);
        stage_rows(c, F.qdo, joint_qd, m.nd);
    }
    __syncthreads();
    if (c.valid)
        for (int j = c.slot; j < nj; j += c.nslot) fs_joint_xform_item(f, j);
    __syncthreads();
    for (int lvl = 0; lvl <= max_depth; ++lvl) {
        if (c.valid)
            for (int j = c.slot; j < nj; j += c.nslot)
                if (f.depth[j] == lvl) fs_fk_vel_item<EPB, true>(f, j);
        __syncthreads();
    }
    store_state(c, a.s_out);
}
