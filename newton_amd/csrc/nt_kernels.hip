// nt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of libnewton_hip.so.
//
// Design (DESIGN.md has the long form):
//  * env-major SoA in HBM: base[(comp * nslot + slot) * ES + env]; a wave reads consecutive envs of one
//    (component, slot) = one coalesced request.
//  * one workgroup owns EPB environments for a whole substep (or a whole rollout): thread -> (env, slot) with
//    env = blockIdx * EPB + tid % EPB and slot = tid / EPB.  A slot-thread plays, phase by phase, body `slot`,
//    shape `slot`, candidate pair `slot`, contact slot `slot`, and joint part `slot` ([0,nj) linear rows,
//    [nj,2nj) angular rows), so every constraint row of every environment is solved by its own lane and lanes of a
//    wave run the same code path.
//  * everything an environment needs during a substep is LDS-resident ([row][EPB], conflict-free because lanes of
//    a wave differ in env first): body state, per-body mass properties (the 3x3 inertia tiles), joint frames, dof
//    limits/gains, shape parameters and the control targets.  HBM is touched once per kernel for state/params
//    and once per substep for the Contacts boundary.
//  * constraint threads publish per-joint / per-contact corrections in LDS and the owning body thread sums
//    them in ascending joint / contact order through a CSR incidence list -- no float atomics, deterministic,
//    and the same order a serial ascending-tid Warp-CPU launch produces for wp.atomic_add.
//  * no MFMA: the largest dense object on this path is a 3x3 inertia.
//
// Reference behaviour (file:line under /root/reference) is cited per phase.
#include "nt_step_preamble.hpp"

namespace {

__global__ void clear_forces_kernel(float* body_f, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) body_f[i] = 0.0f;
}

// 4 B/lane coalesced copy with a known byte count: calibrates the FETCH_SIZE / WRITE_SIZE PMC counters for this
// access pattern (MI355X_MICROARCH.md, HBM section)
__global__ void calibration_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

// the same copy in 16 B per lane, eight loads in flight per lane, one trip per lane at the sizes bench.py uses (tools/microbench/
// hbm_copy.hip, profiles/r06H_hbm_copy.jsonl: 5.65 TB/s in this shape, 4.6 TB/s grid-strided over 4 096 workgroups, hipMemcpy 5.5 TB/s,
// read-only 6.1 TB/s): the streaming rate of this box (bench.py prints it as roofline.hbm_peak_measured beside the vendor peak)
__global__ void __launch_bounds__(256) bandwidth_probe_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[i + k * stride] = v[k];
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

// masked env-column copy: dst[row][env] = src[row][env] for every env whose mask byte is non-zero (RL-style world reset)
__global__ void masked_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, const uint8_t* __restrict__ mask,
                                   int rows, int E, int ES) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)rows * ES;
    if (i >= n) return;
    int env = i % ES;
    if (env < E && mask[env]) dst[i] = src[i];
}

// AoS [E*nslot][ncomp] <-> SoA [ncomp][nslot][ES]
__global__ void pack_kernel(const float* __restrict__ aos, float* __restrict__ soa, int ncomp, int nslot, int E, int ES) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)ncomp * nslot * ES;
    if (i >= n) return;
    int env = i % ES;
    int s = (i / ES) % nslot;
    int comp = i / ((size_t)ES * nslot);
    soa[i] = env < E ? aos[((size_t)env * nslot + s) * ncomp + comp] : 0.0f;
}
__global__ void unpack_kernel(const float* __restrict__ soa, float* __restrict__ aos, int ncomp, int nslot, int E, int ES) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)ncomp * nslot * E;
    if (i >= n) return;
    int comp = i % ncomp;
    int s = (i / ncomp) % nslot;
    int env = i / ((size_t)ncomp * nslot);
    aos[i] = soa[((size_t)comp * nslot + s) * ES + env];
}

// contacts export: exclusive scan of per-env counts (single block), then scatter in (env, pair, k) order
// scan_tmp layout ([4*(E+1)] int32): scanA[E+1] | scanC[E+1] | cntA[E+1] | cntC[E+1]
__global__ void contacts_count_kernel(nt_model m, nt_contacts c, int32_t* scan_tmp) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int E = m.env_count;
    if (env >= E) return;
    const int ES = m.env_stride, ncs = m.np * m.cpp, nas = m.np_analytic * m.cpp;
    int na = 0, nc = 0;
    for (int slot = 0; slot < ncs; ++slot) {
        if (c.shape0[(size_t)slot * ES + env] < 0) continue;
        if (slot < nas) na += 1;
        else nc += 1;
    }
    scan_tmp[2 * (E + 1) + env] = na;
    scan_tmp[3 * (E + 1) + env] = nc;
}
__global__ void contacts_scan_kernel(int E, int32_t* scan_tmp, int32_t* out_count) {
    __shared__ int32_t partA[1024], partC[1024];
    const int32_t *cntA = scan_tmp + 2 * (E + 1), *cntC = scan_tmp + 3 * (E + 1);
    int32_t *scanA = scan_tmp, *scanC = scan_tmp + (E + 1);
    int t = threadIdx.x, T = blockDim.x;
    int per = (E + T - 1) / T;
    int beg = t * per, end = beg + per < E ? beg + per : E;
    int sumA = 0, sumC = 0;
    for (int i = beg; i < end; ++i) { sumA += cntA[i]; sumC += cntC[i]; }
    partA[t] = sumA;
    partC[t] = sumC;
    __syncthreads();
    if (t == 0) {
        int accA = 0, accC = 0;
        for (int i = 0; i < T; ++i) {
            int v = partA[i]; partA[i] = accA; accA += v;
            v = partC[i]; partC[i] = accC; accC += v;
        }
        out_count[0] = accA + accC;
        scanA[E] = accA;  // = first index of the convex section
        scanC[E] = accC;
    }
    __syncthreads();
    int accA = partA[t], accC = partC[t];
    for (int i = beg; i < end; ++i) {
        scanA[i] = accA; accA += cntA[i];
        scanC[i] = accC; accC += cntC[i];
    }
}

struct ExportArgs {
    nt_model m;
    nt_contacts c;
    int cap;
    const int32_t* scan;
    int32_t *shape0, *shape1;
    float *point0, *point1, *offset0, *offset1, *normal, *margin0, *margin1;
};
__global__ void contacts_export_kernel(ExportArgs a) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int E = a.m.env_count;
    if (env >= E) return;
    const int ES = a.m.env_stride, ncs = a.m.np * a.m.cpp, nas = a.m.np_analytic * a.m.cpp;
    int idxA = a.scan[env];
    int idxC = a.scan[E] + a.scan[(E + 1) + env];
    for (int slot = 0; slot < ncs; ++slot) {
        size_t gi = (size_t)slot * ES + env;
        int s0 = a.c.shape0[gi];
        if (s0 < 0) continue;
        int idx = slot < nas ? idxA++ : idxC++;
        if (idx < a.cap) {
            a.shape0[idx] = s0;
            a.shape1[idx] = a.c.shape1[gi];
            const float* D = a.c.data;
            auto ld = [&](int comp) { return D[((size_t)comp * ncs + slot) * ES + env]; };
            for (int k = 0; k < 3; ++k) {
                a.point0[3 * idx + k] = ld(CD_POINT0 + k);
                a.point1[3 * idx + k] = ld(CD_POINT1 + k);
                a.offset0[3 * idx + k] = ld(CD_OFFSET0 + k);
                a.offset1[3 * idx + k] = ld(CD_OFFSET1 + k);
                a.normal[3 * idx + k] = ld(CD_NORMAL + k);
            }
            a.margin0[idx] = ld(CD_MARGIN0);
            a.margin1[idx] = ld(CD_MARGIN1);
        }
    }
}

// convert_contact_impulse_to_force in export order; entries beyond the live count are zeroed
__global__ void contacts_export_force_kernel(nt_model m, nt_contacts c, const float* __restrict__ impulse, float inv_dt, int cap,
                                             const int32_t* __restrict__ scan, float* __restrict__ out) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    const int E = m.env_count;
    if (env >= E) return;
    const int ES = m.env_stride, ncs = m.np * m.cpp, nas = m.np_analytic * m.cpp;
    int idxA = scan[env];
    int idxC = scan[E] + scan[(E + 1) + env];
    for (int slot = 0; slot < ncs; ++slot) {
        size_t gi = (size_t)slot * ES + env;
        if (c.shape0[gi] < 0) continue;
        int idx = slot < nas ? idxA++ : idxC++;
        if (idx < cap)
            for (int k = 0; k < 6; ++k) out[6 * (size_t)idx + k] = impulse[((size_t)k * ncs + slot) * ES + env] * inv_dt;
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
inline int max_threads_for(int epb) { return epb <= 8 ? 256 : 512; }

// slot-threads per env: enough for the widest per-env population (contact slots, joint parts, bodies, shapes,
// pairs), capped by the block size; phases with more items than slot-threads loop.
int slots_for(const nt_model& m, int epb, int max_threads) {
    int want = imax(imax(m.nb, 2 * m.nj), imax(imax(m.ns, m.np), m.np * m.cpp + m.nj));
    int cap = max_threads / epb;
    return want < cap ? want : cap;
}

size_t tile_lds_bytes(const nt_model& m, int epb, bool restitution, bool uni) {
    LdsLayout L = make_layout_host(m, restitution, uni);
    return (size_t)L.rows_per_env * 4 * epb + (size_t)topo_ints(m) * 4 + (size_t)L.uni_floats * 4;
}
bool epb_fits(const nt_model& m, int epb, bool restitution = false, bool uni = false) {
    return tile_lds_bytes(m, epb, restitution, uni) <= LDS_BYTES_PER_CU;
}

// pair-heavy tile: does the workgroup of NT_BIG_SCENE_LANES_WIDE lanes (20 rows of polygon scratch each) still fit the CU?
bool big_wide_fits(const nt_model& m, bool restitution) {
    if (!m.contact_scratch_in_hbm || m.np_analytic == m.np) return false;  // (only the convex kernels carry the wide form)
    LdsLayout L = make_layout_host(m, restitution, false, 0, NT_BIG_SCENE_LANES_WIDE);
    return (size_t)L.rows_per_env * 4 + (size_t)topo_ints(m) * 4 <= LDS_BYTES_PER_CU;
}

int pick_epb(const nt_model& m, int requested, bool restitution = false) {
    // pair-heavy scenes (contact records in HBM) only have the one-environment-per-workgroup kernels
    if (m.contact_scratch_in_hbm) return (requested == 0 || requested == 1) && epb_fits(m, 1, restitution) ? 1 : 0;
    if (requested == 1 || requested == 4 || requested == 8 || requested == 16 || requested == 32 || requested == 64)
        return epb_fits(m, requested, restitution) ? requested : 0;
    // auto: the widest tile (best coalescing) that still yields >= 256 workgroups (one per CU); else the narrowest
    const int cands[4] = {64, 32, 16, 8};
    for (int i = 0; i < 4; ++i) {
        int epb = cands[i];
        if (!epb_fits(m, epb, restitution)) continue;
        int blocks = (m.env_count + epb - 1) / epb;
        if (blocks >= 256 || epb == 8) return epb;
    }
    for (int i = 3; i >= 0; --i)
        if (epb_fits(m, cands[i], restitution)) return cands[i];
    // scenes too large for 8 environments per workgroup (> 20 KB of LDS each): one environment per workgroup, 256 lanes
    // on its items, several workgroups resident per CU while their LDS fits
    return epb_fits(m, 1, restitution) ? 1 : 0;
}

// semi: the SolverSemiImplicit kernel (own scratch layout); max_threads: the kernel's THREADS template argument
// a.tile_opts on entry: the NT_TILE_* layout extras the kernel can use (nt_xpbd_rollout asks); granted only while the tile still fits
// the CU, and the kernel sees what was granted
template <typename K>
nt_status launch(K kernel, KArgs a, int epb, hipStream_t stream, int max_threads = 0, bool semi = false, bool uni = false) {
    int tile_opts = a.tile_opts;
    auto tile_bytes = [&](int opts) {
        LdsLayout Lo = make_layout_host(a.m, xpbd_keeps_prestep_state(a.p), uni, opts, NT_BIG_SCENE_LANES);
        return (size_t)Lo.rows_per_env * 4 * epb + (size_t)topo_ints(a.m) * 4 + (size_t)Lo.uni_floats * 4;
    };
    if (semi) tile_opts = 0;
    if ((tile_opts & NT_TILE_LDS_RECORDS) &&
        (a.m.np_analytic != a.m.np || a.m.contact_scratch_in_hbm || xpbd_keeps_prestep_state(a.p) || tile_bytes(tile_opts) > LDS_BYTES_PER_CU))
        tile_opts &= ~NT_TILE_LDS_RECORDS;
    if (tile_bytes(tile_opts) > LDS_BYTES_PER_CU) tile_opts = 0;
    a.tile_opts = tile_opts;
    if (max_threads <= 0) max_threads = max_threads_for(epb);
    int nslot = slots_for(a.m, epb, max_threads);
    // rows of the SDF legs (nt_contacts.flat) are walked by the environment's slot-lanes too: hundreds per environment in a pile,
    // far more than the tile's own populations ask for -- give them every lane the workgroup may have
    if (a.ct.flat.row_start) nslot = max_threads / epb;
    a.nslot = nslot;
#ifdef NT_ABLATION
    {
        static int dbg = -1;
        if (dbg < 0) { const char* e = getenv("NT_DEBUG_SKIP"); dbg = e ? atoi(e) : 0; }
        a.debug_skip = dbg;
    }
#endif
    int threads = ((nslot * epb + 63) / 64) * 64;
    // the pair-heavy tile's wide workgroup (the caller checked the fit): the kernel sizes its per-lane polygon scratch by blockDim.x
    const int big_lanes = a.m.contact_scratch_in_hbm && threads > NT_BIG_SCENE_LANES ? NT_BIG_SCENE_LANES_WIDE : NT_BIG_SCENE_LANES;
    LdsLayout L = make_layout_host(a.m, xpbd_keeps_prestep_state(a.p), uni, tile_opts, big_lanes);
    size_t lds_bytes = (size_t)(semi ? L.rows_semi : L.rows_per_env) * 4 * epb + (size_t)topo_ints(a.m) * 4 + (size_t)L.uni_floats * 4;
    if (lds_bytes > LDS_BYTES_PER_CU) return NT_ERR_UNSUPPORTED;
    int blocks = (a.m.env_count + epb - 1) / epb;
    if (lds_bytes > 48 * 1024) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

// Launch shape of the analytic (non-convex) fused XPBD rollout: environments per workgroup, workgroup size, minimum waves
// per SIMD (register cap), uniform-parameter tile.  NT_XPBD_CFG="epb,threads,minw[,uni]" selects one of the compiled shapes
// for A/B measurements.
struct XpbdCfg { int epb, threads, minw, uni, cvx; };
inline bool xpbd_cfg_override(XpbdCfg& c) {
    const char* e = getenv("NT_XPBD_CFG");
    if (!e) return false;
    c.uni = c.cvx = 0;
    return sscanf(e, "%d,%d,%d,%d,%d", &c.epb, &c.threads, &c.minw, &c.uni, &c.cvx) >= 3;
}
// Shapes the default dispatch can take are always compiled; the A/B shapes (NT_XPBD_CFG experiments, tests/test_uniform_tile.py on
// the emulated library) only with -DNT_ALL_SHAPES: each fused-rollout instantiation costs ~25 s of hipcc time.
#ifdef NT_ALL_SHAPES
#define NT_XPBD_ROLLOUT_SHAPES(X) \
    X(16, 512, 1, 0) X(16, 256, 1, 0) X(8, 256, 2, 0) \
    X(16, 256, 2, 1) X(16, 512, 1, 1) X(16, 512, 2, 1) X(16, 512, 4, 1) X(32, 512, 1, 1) X(8, 128, 4, 1) X(8, 256, 4, 1)
#define NT_XPBD_ROLLOUT_SHAPES_CVX(X) X(8, 256, 2, 1) X(16, 512, 1, 1) X(16, 256, 2, 1)
#elif defined(NT_DEV_FAST)
#define NT_XPBD_ROLLOUT_SHAPES(X) X(16, 512, 1, 0) X(32, 512, 1, 1) X(16, 512, 1, 1)
#ifdef NT_DEV_CVX  // (+ the convex uniform tile of 16: the 8-box stacks and the box-foot quadruped)
#define NT_XPBD_ROLLOUT_SHAPES_CVX(X) X(16, 512, 1, 1)
#else
#define NT_XPBD_ROLLOUT_SHAPES_CVX(X)
#endif
#else
#define NT_XPBD_ROLLOUT_SHAPES(X) X(16, 512, 1, 0) X(32, 512, 1, 1) X(16, 512, 1, 1)
#define NT_XPBD_ROLLOUT_SHAPES_CVX(X) X(8, 256, 2, 1) X(16, 512, 1, 1)
#endif
// convex (MPR / GJK) variants of the uniform-parameter tile: the parameter diet lets two 8-environment workgroups (or one of 16)
// share a CU where the per-environment tile fits only 8 environments (8-box stacks: 13.6 -> 9.6 KB of LDS per environment)
// the shape uniform-parameter models run by default once there are enough environments to give every CU a tile of 32 (measured,
// MI355X, quadruped: 4096 envs 78 vs 92 M env-steps/s for the 16-env per-environment tile -- half the CUs idle; 8192 envs 158
// vs 101 M; 65536 envs 157 vs 99 M.  2 x (16, 256) per CU: 144-149 M)
constexpr XpbdCfg NT_XPBD_UNI_DEFAULT = {32, 512, 1, 1, 0};
// ... and below that size: the uniform tile of 16 (one per CU up to 4 096 environments).  Same kernels as the per-environment tile with
// the parameters read by broadcast from one block-shared copy: 3 % faster at 4 096 environments (profiles/r05c_ab.txt), and the 89 KB of
// LDS it leaves free hold the tile extras of the fused rollout (NT_TILE_*)
constexpr XpbdCfg NT_XPBD_UNI_SMALL = {16, 512, 1, 1, 0};
// the analytic rollout shape of a uniform-parameter model (false: the per-environment tiles)
inline bool pick_uni_shape(const nt_model& m, bool rest, const nt_collide_params* cp, XpbdCfg& c) {
    if (!m.params_uniform || rest || (cp != nullptr && cp->envs_per_block != 0)) return false;
    if (m.env_count >= 256 * NT_XPBD_UNI_DEFAULT.epb && epb_fits(m, NT_XPBD_UNI_DEFAULT.epb, rest, true)) { c = NT_XPBD_UNI_DEFAULT; return true; }
    if (epb_fits(m, NT_XPBD_UNI_SMALL.epb, rest, true) && m.env_count >= NT_XPBD_UNI_SMALL.epb) { c = NT_XPBD_UNI_SMALL; return true; }
    return false;
}
nt_status launch_xpbd_rollout_shape(const KArgs& a, XpbdCfg c, hipStream_t stream) {
#define X(E, T, W, U) \
    if (!c.cvx && c.epb == E && c.threads == T && c.minw == W && c.uni == U) \
        return launch(xpbd_rollout_kernel<E + U * NT_UNI, false, false, T, W>, a, E, stream, T, false, U != 0);
    NT_XPBD_ROLLOUT_SHAPES(X)
#undef X
#define X(E, T, W, U) \
    if (c.cvx && c.epb == E && c.threads == T && c.minw == W && c.uni == U) \
        return launch(xpbd_rollout_kernel<E + U * NT_UNI, true, false, T, W>, a, E, stream, T, false, U != 0);
    NT_XPBD_ROLLOUT_SHAPES_CVX(X)
#undef X
    return NT_ERR_UNSUPPORTED;
}
// convex models with uniform parameters: the widest uniform tile that fits, once every CU gets at least two of the narrow ones
inline bool pick_cvx_uni_shape(const nt_model& m, bool rest, XpbdCfg& c) {
    if (!m.params_uniform || rest || m.contact_scratch_in_hbm) return false;
    if (m.env_count >= 256 * 16 && tile_lds_bytes(m, 16, rest, true) <= LDS_BYTES_PER_CU) { c = {16, 512, 1, 1, 1}; return true; }
    if (m.env_count >= 256 * 16 && 2 * tile_lds_bytes(m, 8, rest, true) <= LDS_BYTES_PER_CU) { c = {8, 256, 2, 1, 1}; return true; }
    return false;
}

// -DNT_DEV_FAST (measurement builds of tools/build_variant.py only, never the product): just the headline's kernels -- the analytic
// XPBD rollout shapes and the 16-environment collide / step kernels -- so that a kernel experiment compiles in a minute instead of six.
// Everything else answers NT_ERR_UNSUPPORTED.
#ifdef NT_DEV_FAST
#define NT_DISPATCH_EPB(KERNEL, args, epb, stream) ((epb) == 16 ? launch(KERNEL<16>, args, 16, stream) : NT_ERR_UNSUPPORTED)
#define NT_DISPATCH_EPB_CVX(KERNEL, m, args, epb, stream) \
    ((m).np_analytic < (m).np || (epb) != 16 ? NT_ERR_UNSUPPORTED : launch(KERNEL<16, false>, args, 16, stream))
#else
#define NT_DISPATCH_EPB(KERNEL, args, epb, stream)                                      \
    ((epb) == 64 ? launch(KERNEL<64>, args, 64, stream)                                 \
     : (epb) == 32 ? launch(KERNEL<32>, args, 32, stream)                               \
     : (epb) == 16 ? launch(KERNEL<16>, args, 16, stream)                               \
     : ((epb) == 8 || (epb) == 4) ? launch(KERNEL<8>, args, 8, stream) : launch(KERNEL<1>, args, 1, stream))

#define NT_DISPATCH_EPB2(KERNEL, B, args, epb, stream)                                  \
    ((epb) == 64 ? launch(KERNEL<64, B>, args, 64, stream)                              \
     : (epb) == 32 ? launch(KERNEL<32, B>, args, 32, stream)                            \
     : (epb) == 16 ? launch(KERNEL<16, B>, args, 16, stream)                            \
     : ((epb) == 8 || (epb) == 4) ? launch(KERNEL<8, B>, args, 8, stream) : launch(KERNEL<1, B>, args, 1, stream))
// kernels that collide are compiled twice: the convex (MPR/GJK) code only exists in the variant used by models
// that have convex-routed pairs, so analytic-only models keep their register budget
// the convex variants are only instantiated for 1 / 8 / 16 envs per workgroup (build time): wider tiles fall back to 16
#define NT_DISPATCH_EPB_CVX(KERNEL, m, args, epb, stream)                                              \
    ((m).np_analytic < (m).np                                                                          \
         ? ((epb) >= 16 ? launch(KERNEL<16, true>, args, 16, stream)                                   \
            : ((epb) == 8 || (epb) == 4) ? launch(KERNEL<8, true>, args, 8, stream) : launch(KERNEL<1, true>, args, 1, stream)) \
         : NT_DISPATCH_EPB2(KERNEL, false, args, epb, stream))
#endif


}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* nt_error_string(nt_status s) {
    switch (s) {
        case NT_OK: return "ok";
        case NT_ERR_INVALID_ARG: return "invalid argument";
        case NT_ERR_LAUNCH: return "kernel launch failed";
        case NT_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown";
    }
}

int32_t nt_lds_bytes_per_env(const nt_model* m) {
    if (!m) return -1;
    return make_layout_host(*m).rows_per_env * 4;
}

int32_t nt_pick_envs_per_block(const nt_model* m, int32_t requested) {
    if (!model_ok(m)) return 0;
    return pick_epb(*m, requested);
}

nt_status nt_clear_forces(const nt_model* m, nt_state* s, void* stream) {
    if (!model_ok(m) || !s || !s->body_f) return NT_ERR_INVALID_ARG;
    size_t n = (size_t)6 * m->nb * m->env_stride;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(clear_forces_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->body_f, n);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_collide(const nt_model* m, const nt_state* s, nt_contacts* c, const nt_collide_params* p, void* stream) {
    if (!model_ok(m) || !s || !c || !s->body_q) return NT_ERR_INVALID_ARG;
    if (m->np == 0 && !c->world_xform) return NT_OK;
    if (c->world_xform && (!c->world_aabb_lower || !c->world_aabb_upper)) return NT_ERR_INVALID_ARG;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s;
    a.ct = *c;
    int epb = pick_epb(*m, p ? p->envs_per_block : 0);
    if (!epb) return NT_ERR_UNSUPPORTED;
    if (m->np == 0) return NT_DISPATCH_EPB(shapes_export_kernel, a, epb, (hipStream_t)stream);  // every pair lives outside the tiles
    // uniform-parameter tile: models whose environments share their parameters stage ONE copy per workgroup instead of 3.8 KB per
    // environment per launch (the per-call API pays the parameter staging in every kernel; the fused rollout once per frame)
    if (m->params_uniform && !m->contact_scratch_in_hbm && m->np_analytic == m->np && epb == 16 && epb_fits(*m, 16, false, true))
        return launch(collide_kernel<16 + NT_UNI, false>, a, 16, (hipStream_t)stream, 0, false, true);
#ifdef NT_DEV_FAST
    if (m->contact_scratch_in_hbm) return NT_ERR_UNSUPPORTED;
#else
    if (m->contact_scratch_in_hbm && big_wide_fits(*m, false))
        return launch(collide_kernel<1, true, true, NT_BIG_SCENE_LANES_WIDE>, a, 1, (hipStream_t)stream, NT_BIG_SCENE_LANES_WIDE);
    if (m->contact_scratch_in_hbm)
        return m->np_analytic < m->np ? launch(collide_kernel<1, true, true>, a, 1, (hipStream_t)stream)
                                      : launch(collide_kernel<1, false, true>, a, 1, (hipStream_t)stream);
#endif
    return NT_DISPATCH_EPB_CVX(collide_kernel, *m, a, epb, (hipStream_t)stream);
}

nt_status nt_xpbd_step(const nt_model* m, const nt_xpbd_params* p, nt_state* s_in, nt_state* s_out, const nt_control* ctrl,
                       const nt_contacts* c, float dt, int32_t envs_per_block, const nt_xpbd_report* report, void* stream) {
    if (!model_ok(m) || !p || !s_in || !s_out || !ctrl) return NT_ERR_INVALID_ARG;
    if (s_out->body_parent_f && m->nj > 0 && !(report && report->joint_impulse)) return NT_ERR_INVALID_ARG;
    // the restitution pass covers the rows of the SDF legs through their own records: refuse to skip them silently
    if (p->enable_restitution && c && c->flat.row_start && !c->flat.restitution) return NT_ERR_INVALID_ARG;
    KArgs a = {};
    if (report) a.rep = *report;
    if (!s_out->body_parent_f || m->nj == 0) a.rep.joint_impulse = nullptr;
    a.m = *m;
    a.s_in = *s_in;
    a.s_out = *s_out;
    a.c = *ctrl;
    if (c) a.ct = *c;
    a.has_contacts = (c != nullptr && (m->np > 0 || c->flat.row_start)) ? 1 : 0;  // fixed slots and / or rows of the SDF legs
    a.p = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    int epb = pick_epb(*m, envs_per_block, xpbd_keeps_prestep_state(*p));
    if (!epb) return NT_ERR_UNSUPPORTED;
#ifdef NT_DEV_FAST
    if (m->contact_scratch_in_hbm) return NT_ERR_UNSUPPORTED;
#else
    if (m->contact_scratch_in_hbm) {
        if (a.has_contacts && m->np > 0 && !a.ct.cw) return NT_ERR_INVALID_ARG;
        return launch(xpbd_step_kernel<1, true>, a, 1, (hipStream_t)stream);
    }
#endif
    if (m->params_uniform && !xpbd_keeps_prestep_state(*p) && epb == 16 && epb_fits(*m, 16, false, true))
        return launch(xpbd_step_kernel<16 + NT_UNI>, a, 16, (hipStream_t)stream, 0, false, true);
    return NT_DISPATCH_EPB(xpbd_step_kernel, a, epb, (hipStream_t)stream);
}

nt_status nt_xpbd_rollout(const nt_model* m, const nt_xpbd_params* p, const nt_collide_params* cp, nt_state* s0, nt_state* s1,
                          const nt_control* ctrl, nt_contacts* c, float dt, int32_t substeps, void* stream) {
    if (!model_ok(m) || !p || !s0 || !s1 || !ctrl || !c || substeps < 1) return NT_ERR_INVALID_ARG;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s0;
    a.s_out = *s1;
    a.c = *ctrl;
    a.ct = *c;
    a.has_contacts = m->np > 0 ? 1 : 0;
    a.p = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    a.substeps = substeps;
    a.tile_opts = NT_TILE_POSE_SNAPSHOT | NT_TILE_LDS_RECORDS;  // (request; launch() grants what the tile has room for)
    const bool rest = xpbd_keeps_prestep_state(*p);
    int epb = pick_epb(*m, cp ? cp->envs_per_block : 0, rest);
    if (!epb) return NT_ERR_UNSUPPORTED;
#ifdef NT_DEV_FAST
    if (m->contact_scratch_in_hbm) return NT_ERR_UNSUPPORTED;
#else
    if (m->contact_scratch_in_hbm) {
        if (a.has_contacts && !a.ct.cw) return NT_ERR_INVALID_ARG;
        if (big_wide_fits(*m, rest))
            return launch(xpbd_rollout_kernel<1, true, true, NT_BIG_SCENE_LANES_WIDE>, a, 1, (hipStream_t)stream, NT_BIG_SCENE_LANES_WIDE);
        return m->np_analytic < m->np ? launch(xpbd_rollout_kernel<1, true, true>, a, 1, (hipStream_t)stream)
                                      : launch(xpbd_rollout_kernel<1, false, true>, a, 1, (hipStream_t)stream);
    }
#endif
    if (m->np_analytic < m->np) {  // convex models: uniform-parameter tiles when the parameters allow
        XpbdCfg c;
        if (xpbd_cfg_override(c) && c.cvx) {
            if ((c.uni && (!m->params_uniform || rest)) || !epb_fits(*m, c.epb, rest, c.uni != 0)) return NT_ERR_UNSUPPORTED;
            return launch_xpbd_rollout_shape(a, c, (hipStream_t)stream);
        }
        if ((cp == nullptr || cp->envs_per_block == 0) && pick_cvx_uni_shape(*m, rest, c))
            return launch_xpbd_rollout_shape(a, c, (hipStream_t)stream);
    }
    if (m->np_analytic == m->np) {  // analytic-only models: the tuned launch shapes
        XpbdCfg c;
        if (xpbd_cfg_override(c) && !c.cvx) {
            if ((c.uni && (!m->params_uniform || rest)) || !epb_fits(*m, c.epb, rest, c.uni != 0)) return NT_ERR_UNSUPPORTED;
            return launch_xpbd_rollout_shape(a, c, (hipStream_t)stream);
        }
        if (pick_uni_shape(*m, rest, cp, c)) return launch_xpbd_rollout_shape(a, c, (hipStream_t)stream);
    }
    return NT_DISPATCH_EPB_CVX(xpbd_rollout_kernel, *m, a, epb, (hipStream_t)stream);
}

nt_status nt_xpbd_rollout_shape(const nt_model* m, const nt_xpbd_params* p, const nt_collide_params* cp, int32_t out[5]) {
    if (!model_ok(m) || !p || !out) return NT_ERR_INVALID_ARG;
    const bool rest = xpbd_keeps_prestep_state(*p);
    int epb = pick_epb(*m, cp ? cp->envs_per_block : 0, rest);
    if (!epb) return NT_ERR_UNSUPPORTED;
    const bool cvx = m->np_analytic < m->np, big = m->contact_scratch_in_hbm != 0;
    XpbdCfg c = {epb, max_threads_for(epb), 1, 0, 0};
    if (big) c = {1, big_wide_fits(*m, rest) ? NT_BIG_SCENE_LANES_WIDE : NT_BIG_SCENE_LANES, 1, 0, 0};
    else if (!cvx) {
        XpbdCfg o;
        if (xpbd_cfg_override(o) && !o.cvx) c = o;
        else if (pick_uni_shape(*m, rest, cp, o)) c = o;
        else if (epb == 4) c.epb = 8;
    } else {
        XpbdCfg o;
        if (xpbd_cfg_override(o) && o.cvx) c = o;
        else if ((cp == nullptr || cp->envs_per_block == 0) && pick_cvx_uni_shape(*m, rest, o)) c = o;
        else {
            c.epb = epb >= 16 ? 16 : (epb >= 4 ? 8 : 1);
            c.threads = max_threads_for(c.epb);
        }
    }
    out[0] = c.epb; out[1] = c.threads; out[2] = c.minw; out[3] = c.uni; out[4] = (cvx ? 1 : 0) | (big ? 2 : 0);
    return NT_OK;
}

nt_status nt_semi_implicit_step(const nt_model* m, const nt_semi_implicit_params* p, nt_state* s_in, nt_state* s_out,
                                const nt_control* ctrl, const nt_contacts* c, float dt, int32_t envs_per_block, void* stream) {
    if (!model_ok(m) || !p || !s_in || !s_out || !ctrl) return NT_ERR_INVALID_ARG;
    KArgs a = {};
    a.m = *m;
    a.s_in = *s_in;
    a.s_out = *s_out;
    a.c = *ctrl;
    if (c) a.ct = *c;
    a.has_contacts = (c != nullptr && m->np > 0) ? 1 : 0;
    a.sp = *p;
    a.angular_damping = p->angular_damping;
    a.dt = dt;
    if (m->contact_scratch_in_hbm) return NT_ERR_UNSUPPORTED;  // XPBD / collide only
#ifdef NT_DEV_FAST
    return NT_ERR_UNSUPPORTED;
#else
    int epb = pick_epb(*m, envs_per_block);
    if (!epb) return NT_ERR_UNSUPPORTED;
    if ((size_t)make_layout_host(*m).rows_semi * 4 * epb + (size_t)topo_ints(*m) * 4 > LDS_BYTES_PER_CU) {
        // the wrench records of SolverSemiImplicit make its tile heavier than the collide / XPBD one: narrow it
        while (epb > 8 && (size_t)make_layout_host(*m).rows_semi * 4 * epb + (size_t)topo_ints(*m) * 4 > LDS_BYTES_PER_CU) epb /= 2;
        if ((size_t)make_layout_host(*m).rows_semi * 4 * epb + (size_t)topo_ints(*m) * 4 > LDS_BYTES_PER_CU) epb = 1;
    }
    return epb == 64 ? launch(semi_implicit_step_kernel<64>, a, 64, (hipStream_t)stream, 0, true)
         : epb == 32 ? launch(semi_implicit_step_kernel<32>, a, 32, (hipStream_t)stream, 0, true)
         : epb == 16 ? launch(semi_implicit_step_kernel<16>, a, 16, (hipStream_t)stream, 0, true)
         : epb == 8 ? launch(semi_implicit_step_kernel<8>, a, 8, (hipStream_t)stream, 0, true)
                    : launch(semi_implicit_step_kernel<1>, a, 1, (hipStream_t)stream, 0, true);
#endif
}

#ifdef NT_PHASE_TIMING
// debug build only: read and reset the phase cycle counters
int nt_debug_phase_clocks(unsigned long long* out) {
    unsigned long long zero[32] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nt_phase_clock), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(nt_phase_clock), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

nt_status nt_state_reset(const nt_model* m, nt_state* dst, const nt_state* src, const uint8_t* world_mask, void* stream) {
    if (!model_ok(m) || !dst || !src || !world_mask) return NT_ERR_INVALID_ARG;
    auto go = [&](float* d, const float* s_, int rows) {
        if (!d || !s_ || rows <= 0) return;
        size_t n = (size_t)rows * m->env_stride;
        hipLaunchKernelGGL(masked_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, s_,
                           world_mask, rows, m->env_count, m->env_stride);
    };
    go(dst->body_q, src->body_q, 7 * m->nb);
    go(dst->body_qd, src->body_qd, 6 * m->nb);
    go(dst->body_f, src->body_f, 6 * m->nb);
    go(dst->joint_q, src->joint_q, m->nc);
    go(dst->joint_qd, src->joint_qd, m->nd);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_calibration_copy(const float* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(calibration_copy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, src, dst, (size_t)n);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_bandwidth_probe(const float* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0 || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return NT_ERR_INVALID_ARG;
#ifdef NT_EMULATED_GRID
    const int blocks = NT_EMULATED_GRID;
#else
    size_t want = ((size_t)n / 4 + 256 * 8 - 1) / (256 * 8);  // one trip of eight 16-byte accesses per lane
    const int blocks = (int)(want < 1 ? 1 : (want > 65536 ? 65536 : want));
#endif
    hipLaunchKernelGGL(bandwidth_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, (size_t)n / 4);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_pack_aos(const float* aos, float* soa, int32_t ncomp, int32_t nslot, int32_t env_count, int32_t env_stride,
                      void* stream) {
    if (!aos || !soa || ncomp <= 0 || nslot <= 0 || env_count <= 0 || env_stride < env_count) return NT_ERR_INVALID_ARG;
    size_t n = (size_t)ncomp * nslot * env_stride;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, aos, soa, ncomp,
                       nslot, env_count, env_stride);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_unpack_aos(const float* soa, float* aos, int32_t ncomp, int32_t nslot, int32_t env_count, int32_t env_stride,
                        void* stream) {
    if (!aos || !soa || ncomp <= 0 || nslot <= 0 || env_count <= 0 || env_stride < env_count) return NT_ERR_INVALID_ARG;
    size_t n = (size_t)ncomp * nslot * env_count;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, soa, aos, ncomp,
                       nslot, env_count, env_stride);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_contacts_export(const nt_model* m, const nt_contacts* c, int32_t cap, int32_t* out_count, int32_t* out_shape0,
                             int32_t* out_shape1, float* out_point0, float* out_point1, float* out_offset0,
                             float* out_offset1, float* out_normal, float* out_margin0, float* out_margin1,
                             int32_t* scan_tmp, void* stream) {
    if (!model_ok(m) || !c || !out_count || !scan_tmp) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(contacts_count_kernel, dim3((m->env_count + 63) / 64), dim3(64), 0, (hipStream_t)stream, *m, *c, scan_tmp);
    hipLaunchKernelGGL(contacts_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, m->env_count, scan_tmp, out_count);
    ExportArgs a;
    a.m = *m;
    a.c = *c;
    a.cap = cap;
    a.scan = scan_tmp;
    a.shape0 = out_shape0; a.shape1 = out_shape1;
    a.point0 = out_point0; a.point1 = out_point1;
    a.offset0 = out_offset0; a.offset1 = out_offset1;
    a.normal = out_normal; a.margin0 = out_margin0; a.margin1 = out_margin1;
    hipLaunchKernelGGL(contacts_export_kernel, dim3((m->env_count + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_contacts_export_force(const nt_model* m, const nt_contacts* c, const float* contact_impulse, float dt, int32_t cap,
                                   float* out_force, int32_t* scan_tmp, void* stream) {
    if (!model_ok(m) || !c || !contact_impulse || !out_force || !scan_tmp || cap < 0 || !(dt > 0.0f)) return NT_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int E = m->env_count;
    if (hipMemsetAsync(out_force, 0, sizeof(float) * 6 * (size_t)cap, st) != hipSuccess) return NT_ERR_LAUNCH;
    hipLaunchKernelGGL(contacts_count_kernel, dim3((E + 63) / 64), dim3(64), 0, st, *m, *c, scan_tmp);
    // the total goes into the spare last entry of the cntA section
    hipLaunchKernelGGL(contacts_scan_kernel, dim3(1), dim3(1024), 0, st, E, scan_tmp, scan_tmp + 2 * (E + 1) + E);
    hipLaunchKernelGGL(contacts_export_force_kernel, dim3((E + 63) / 64), dim3(64), 0, st, *m, *c, contact_impulse, 1.0f / dt, cap,
                       scan_tmp, out_force);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
