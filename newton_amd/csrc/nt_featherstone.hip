// nt_featherstone.hip -- SolverFeatherstone (step / fused rollout) and eval_fk: launch code + C ABI.  The kernels are
// nt_featherstone_kernels.hpp, their phases nt_featherstone.hpp (namespace ieee); this unit exists so that they are compiled with the
// default scheduler (see nt_step_preamble.hpp).
#include "nt_step_preamble.hpp"

extern "C" {

// shared launch logic of the Featherstone kernels (step / rollout)
static nt_status fs_launch(const nt_model* m, KArgs& a, int32_t envs_per_block, bool rollout, hipStream_t stream) {
    if (m->contact_scratch_in_hbm) return NT_ERR_UNSUPPORTED;  // XPBD / collide only
#ifdef NT_DEV_FAST
    return NT_ERR_UNSUPPORTED;
#else
#ifdef NT_ABLATION
    {
        const char* e = getenv("NT_DEBUG_SKIP");
        a.debug_skip = e ? atoi(e) : 0;
    }
#endif
    const bool tree = fs_tree_mode(a);  // (a.fp is set by the caller: the layout of the launch follows the mass-matrix mode)
    FsLayout F = make_fs_layout(*m, make_layout(*m, false, false, false, false), tree);
    const size_t shared_ints = (size_t)topo_ints(*m) + fs_topo_ints(*m);
    auto fits = [&](int epb) { return (size_t)F.rows * 4 * epb + shared_ints * 4 <= LDS_BYTES_PER_CU; };
    const bool cvx = m->np_analytic < m->np;
    // Uniform-parameter tile of 16 (round 6): the level-synchronous phases of this solver keep a handful of lanes per environment busy
    // and wait on LDS round trips and barriers, so what a CU delivers is the number of environments it holds.  With ONE block-shared
    // parameter copy and the tree-structured solve region (make_fs_layout) an Anymal-class environment needs 9.3 KB instead of 17 KB:
    // 16 per CU in one 512-lane workgroup -- 4 096 environments in ONE round of 256 workgroups instead of two rounds of 1 024 x 4.
    const LdsLayout Lu = make_layout(*m, false, false, true, false);
    const FsLayout Fu = make_fs_layout(*m, Lu, tree);
    // (automatic from 2 049 environments: up to 2 048 the tiles of 4 are ONE round of 512 workgroups already and keep all 64 lanes per
    // environment -- 37.6 vs 32.6 M env-steps/s, profiles/r06H_ab_workloads.txt; envs_per_block = 16 asks for it at any size)
    const bool uni16 = rollout && !cvx && m->params_uniform && (envs_per_block == 16 || (envs_per_block == 0 && m->env_count > 2048)) &&
                       (size_t)Fu.rows * 4 * 16 + shared_ints * 4 + (size_t)Lu.uni_floats * 4 <= LDS_BYTES_PER_CU;
    int epb = 0;
    if (uni16) {
        epb = 16;
    } else if (envs_per_block == 1 || envs_per_block == 4 || envs_per_block == 8 || envs_per_block == 16) {
        epb = fits(envs_per_block) ? envs_per_block : 0;
    } else {
        // measured on MI355X (4096 quadrupeds): 4 envs per workgroup (16 cooperating lanes per env in the Cholesky wave,
        // two resident workgroups per CU) beats 8; 16 rarely fits; articulations too large for 4 (P + H alone are
        // (6 nj + nd) x max_art_dofs floats per environment) run one environment per workgroup, 64 lanes in the Cholesky wave
        const int cands[4] = {4, 8, 16, 1};
        for (int i = 0; i < 4 && !epb; ++i)
            if (fits(cands[i])) epb = cands[i];
    }
    if (!epb) return NT_ERR_UNSUPPORTED;
    if (rollout && cvx && epb == 16) epb = 8;  // the convex rollout is only instantiated for 4 / 8 envs per workgroup
    int want = imax(imax(m->nb, m->nj), imax(m->np * m->cpp, imax(m->nj, m->nd) * m->max_art_dofs));
    const int max_threads = uni16 ? 512 : 256;
    int cap = max_threads / epb;
    a.nslot = want < cap ? want : cap;
    int threads = ((a.nslot * epb + 63) / 64) * 64;
    if (uni16) F = Fu;
    size_t lds_bytes = (size_t)F.rows * 4 * epb + shared_ints * 4 + (uni16 ? (size_t)Lu.uni_floats * 4 : 0);
    int blocks = (m->env_count + epb - 1) / epb;
    auto go = [&](auto kernel) -> nt_status {
        if (lds_bytes > 48 * 1024 &&
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NT_ERR_LAUNCH;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, a);
        return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
    };
    if (uni16) return go(featherstone_rollout_kernel<16 + NT_UNI, false, 512>);
    if (!rollout) {
        if (epb == 16) return go(featherstone_step_kernel<16>);
        if (epb == 8) return go(featherstone_step_kernel<8>);
        if (epb == 4) return go(featherstone_step_kernel<4>);
        return go(featherstone_step_kernel<1>);
    }
    if (cvx) {
        if (epb == 8) return go(featherstone_rollout_kernel<8, true>);
        return epb == 4 ? go(featherstone_rollout_kernel<4, true>) : go(featherstone_rollout_kernel<1, true>);
    }
    if (epb == 16) return go(featherstone_rollout_kernel<16, false>);
    if (epb == 8) return go(featherstone_rollout_kernel<8, false>);
    if (epb == 4) return go(featherstone_rollout_kernel<4, false>);
    return go(featherstone_rollout_kernel<1, false>);
#endif
}

static bool fs_state_ok(const nt_state* s) { return s && s->joint_q && s->joint_qd && s->body_q && s->body_qd; }

nt_status nt_featherstone_step(const nt_model* m, const nt_featherstone_params* p, nt_state* s_in, nt_state* s_out,
                                const nt_control* ctrl, const nt_contacts* c, float dt, int32_t envs_per_block, void* stream) {
    if (!model_ok(m) || !p || !ctrl || !fs_state_ok(s_in) || !fs_state_ok(s_out)) return NT_ERR_INVALID_ARG;
    if (m->nj <= 0 || m->na <= 0 || m->max_art_dofs < 0 || !m->art_start) return NT_ERR_UNSUPPORTED;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s_in;
    a.s_out = *s_out;
    a.c = *ctrl;
    if (c) a.ct = *c;
    a.has_contacts = (c != nullptr && m->np > 0) ? 1 : 0;
    a.sp.friction_smoothing = p->friction_smoothing;
    a.fp = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    return fs_launch(m, a, envs_per_block, false, (hipStream_t)stream);
}

nt_status nt_featherstone_rollout(const nt_model* m, const nt_featherstone_params* p, const nt_collide_params* cp, nt_state* s0,
                                   nt_state* s1, const nt_control* ctrl, nt_contacts* c, float dt, int32_t substeps,
                                   void* stream) {
    if (!model_ok(m) || !p || !ctrl || !c || !fs_state_ok(s0) || !fs_state_ok(s1) || !s0->body_f || !s1->body_f || substeps <= 0)
        return NT_ERR_INVALID_ARG;
    if (m->nj <= 0 || m->na <= 0 || m->max_art_dofs < 0 || !m->art_start) return NT_ERR_UNSUPPORTED;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s0;
    a.s_out = *s1;
    a.c = *ctrl;
    a.ct = *c;
    a.has_contacts = m->np > 0 ? 1 : 0;
    a.sp.friction_smoothing = p->friction_smoothing;
    a.fp = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    a.substeps = substeps;
    return fs_launch(m, a, cp ? cp->envs_per_block : 0, true, (hipStream_t)stream);
}

int32_t nt_featherstone_lds_bytes_per_env(const nt_model* m) {
    if (!m) return -1;
    return make_fs_layout(*m, make_layout(*m, false, false, false, false)).rows * 4;  // (dense mass-matrix region, per-environment parameters: the largest form)
}

nt_status nt_eval_fk(const nt_model* m, const float* joint_q, const float* joint_qd, nt_state* out, void* stream) {
    if (!model_ok(m) || !joint_q || !joint_qd || !out || !out->body_q || !out->body_qd) return NT_ERR_INVALID_ARG;
    if (m->nj <= 0) return NT_ERR_UNSUPPORTED;
#ifdef NT_DEV_FAST
    return NT_ERR_UNSUPPORTED;
#else
    KArgs a = {};
    a.m = *m;
    a.s_out = *out;
    const FsLayout F = make_fs_layout(*m, make_layout(*m, false, false, false, false), fs_tree_mode(a));  // (the kernel's own rule)
    const size_t shared_ints = (size_t)topo_ints(*m) + fs_topo_ints(*m);
    int epb = 0;
    const int cands[4] = {16, 8, 4, 1};
    for (int i = 0; i < 4 && !epb; ++i)
        if ((size_t)F.rows * 4 * cands[i] + shared_ints * 4 <= LDS_BYTES_PER_CU) epb = cands[i];
    if (!epb) return NT_ERR_UNSUPPORTED;
    int want = imax(m->nb, m->nj), cap = 256 / epb;
    a.nslot = want < cap ? want : cap;
    int threads = ((a.nslot * epb + 63) / 64) * 64;
    size_t lds_bytes = (size_t)F.rows * 4 * epb + shared_ints * 4;
    int blocks = (m->env_count + epb - 1) / epb;
    auto go = [&](auto kernel) -> nt_status {
        if (lds_bytes > 48 * 1024 &&
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NT_ERR_LAUNCH;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, a, joint_q, joint_qd);
        return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
    };
    if (epb == 16) return go(eval_fk_kernel<16>);
    if (epb == 8) return go(eval_fk_kernel<8>);
    if (epb == 4) return go(eval_fk_kernel<4>);
    return go(eval_fk_kernel<1>);
#endif
}


#ifdef NT_PHASE_TIMING
// debug build only: read and reset this unit's phase cycle counters (the counters are per translation unit)
int nt_debug_phase_clocks_fs(unsigned long long* out) {
    unsigned long long zero[32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nt_phase_clock), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(nt_phase_clock), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
