// MEASUREMENT TOOL: streaming bandwidth of the box for a few copy / read shapes (which one nt_bandwidth_probe should use).
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/hbm_copy.hip -o gpurun_out/hbm_copy ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int U>
__global__ void __launch_bounds__(256) copy16(const float4* __restrict__ s, float4* __restrict__ d, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i + (U - 1) * st < n4; i += U * st) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = s[i + k * st];
#pragma unroll
        for (int k = 0; k < U; ++k) d[i + k * st] = v[k];
    }
    for (; i < n4; i += st) d[i] = s[i];
}
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void __launch_bounds__(256) copy16_nt(const f4* __restrict__ s, f4* __restrict__ d, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i + (U - 1) * st < n4; i += U * st) {
        f4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(&s[i + k * st]);
#pragma unroll
        for (int k = 0; k < U; ++k) __builtin_nontemporal_store(v[k], &d[i + k * st]);
    }
    for (; i < n4; i += st) d[i] = s[i];
}
template <int U>
__global__ void __launch_bounds__(256) read16(const float4* __restrict__ s, float* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (; i + (U - 1) * st < n4; i += U * st) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = s[i + k * st];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
template <typename F>
double time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    double best = 1e30;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    float4 *s, *d; float* o;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
        double t4 = time_ms([&] { hipLaunchKernelGGL(copy16<4>, dim3(g), dim3(256), 0, 0, s, d, n4); }, 10);
        double t8 = time_ms([&] { hipLaunchKernelGGL(copy16<8>, dim3(g), dim3(256), 0, 0, s, d, n4); }, 10);
        double tn = time_ms([&] { hipLaunchKernelGGL(copy16_nt<4>, dim3(g), dim3(256), 0, 0, (const f4*)s, (f4*)d, n4); }, 10);
        double tr = time_ms([&] { hipLaunchKernelGGL(read16<8>, dim3(g), dim3(256), 0, 0, s, o, n4); }, 10);
        printf("{\"grid\": %d, \"copy_u4_gbps\": %.0f, \"copy_u8_gbps\": %.0f, \"copy_nt_u4_gbps\": %.0f, \"read_u8_gbps\": %.0f}\n", g,
               2.0 * bytes / t4 / 1e6, 2.0 * bytes / t8 / 1e6, 2.0 * bytes / tn / 1e6, 1.0 * bytes / tr / 1e6);
    }
    double tm = time_ms([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, 10);
    printf("{\"hipMemcpyDtoD_gbps\": %.0f}\n", 2.0 * bytes / tm / 1e6);
    return 0;
}
