// valu_issue.hip -- pins the VALU issue cost on gfx950: cycles per wave-instruction for dependent and
// independent v_fma_f32 / v_pk_fma_f32 / v_mul+v_add / v_rcp_f32 streams and for a dependent ds_read chain, at 1, 2, 4 and 8 waves per
// SIMD.  Measurement tool only (never linked into libnewton_hip.so):
//     hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_issue.hip -o gpurun_out/valu_issue && gpurun_out/valu_issue
// One workgroup per CU (a 96 KB LDS allocation keeps a second one off the CU), THREADS = 256 * waves-per-SIMD.  Every wave reads
// s_memtime around its loop; the table reports the median over waves of cycles / (instructions issued by that wave) and the
// aggregate rate per SIMD = waves per SIMD / that interval.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

enum Mode { FMA = 0, PK_FMA = 1, MUL_ADD = 2, RCP = 3, LDS_CHAIN = 4, FMA_HALF_LANES = 5, PK_MUL = 6, CNDMASK = 7 };

template <int MODE, int ILP>
__global__ void __launch_bounds__(1024) k(unsigned long long* cyc, float* sink, int iters, int lds_floats) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < lds_floats; i += blockDim.x) lds[i] = (float)((i * 17 + 1) % lds_floats) ;
    __syncthreads();
    float a = 1.0000001f, b = 1e-9f;
    float x[8];
    float2v p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = 1.0f + tid * 1e-6f + i; p[i] = float2v{x[i], x[i] + 0.5f}; }
    float2v pa = {a, a}, pb = {b, b};
    int idx = tid % lds_floats;
    if (MODE == FMA_HALF_LANES && (tid & 63) >= 16) { /* keep exec = 16 lanes */ }
    const bool active = MODE != FMA_HALF_LANES || (tid & 63) < 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 64 / ILP; ++r) {
#pragma unroll
                for (int i = 0; i < ILP; ++i) {
                    if (MODE == FMA || MODE == FMA_HALF_LANES) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                    else if (MODE == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
                    else if (MODE == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
                    else if (MODE == MUL_ADD) {
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                    } else if (MODE == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                    else if (MODE == CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
                    else if (MODE == LDS_CHAIN) { idx = (int)lds[idx]; }
                }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    s += (float)idx;
    if ((tid & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE, int ILP>
static void run(const char* name, int wps) {
    const int blocks = 256, threads = 256 * wps, iters = 2000;
    const int lds_floats = 24 * 1024;  // 96 KB: one workgroup per CU
    unsigned long long* cyc;
    float* sink;
    const int nw = blocks * threads / 64;
    hipMalloc(&cyc, nw * sizeof(unsigned long long));
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)k<MODE, ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, ILP>), dim3(blocks), dim3(threads), lds_floats * 4, 0, cyc, sink, iters, lds_floats);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_iter = MODE == MUL_ADD || MODE == CNDMASK ? 128.0 : 64.0;  // wave-instructions per loop iteration
    const double n = per_iter * iters;
    const double med = (double)h[nw / 2] / n, lo = (double)h[0] / n, hi = (double)h[nw - 1] / n;
    // aggregate: wave-instructions per SIMD per cycle from the median interval, and from the wall clock at 2.4 GHz
    const double wall_cycles = ms * 1e-3 * 2.4e9;
    printf("{\"stream\": \"%s\", \"ilp\": %d, \"waves_per_simd\": %d, \"cycles_per_wave_instr_median\": %.3f, \"min\": %.3f, \"max\": %.3f, "
           "\"wave_instr_per_simd_cycle\": %.4f, \"kernel_ms\": %.4f, \"wall_cycles_per_instr_per_simd\": %.3f}\n",
           name, ILP, wps, med, lo, hi, wps / med, ms, wall_cycles / (n * wps));
    hipFree(cyc); hipFree(sink);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("{\"device\": \"%s\", \"gcn\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    for (int wps : {1, 2, 4}) {
        run<FMA, 1>("v_fma_f32 dependent", wps);
        run<FMA, 4>("v_fma_f32 4 chains", wps);
        run<FMA, 8>("v_fma_f32 8 chains", wps);
        run<PK_FMA, 1>("v_pk_fma_f32 dependent", wps);
        run<PK_FMA, 4>("v_pk_fma_f32 4 chains", wps);
        run<PK_FMA, 8>("v_pk_fma_f32 8 chains", wps);
        run<PK_MUL, 8>("v_pk_mul_f32 8 chains", wps);
        run<MUL_ADD, 1>("v_mul_f32+v_add_f32 dependent", wps);
        run<MUL_ADD, 8>("v_mul_f32+v_add_f32 8 chains", wps);
        run<RCP, 1>("v_rcp_f32 dependent", wps);
        run<RCP, 8>("v_rcp_f32 8 chains", wps);
        run<CNDMASK, 8>("v_cmp+v_cndmask 8 chains", wps);
        run<FMA_HALF_LANES, 8>("v_fma_f32 8 chains, 16 of 64 lanes", wps);
        run<LDS_CHAIN, 1>("ds_read_b32 dependent chain (latency)", wps);
    }
    return 0;
}
