// hip_emu.h -- TEST INFRASTRUCTURE ONLY: a minimal host stand-in for the HIP runtime pieces the gfx950 kernels use, so that the
// very same kernel sources can be executed on the CPU by the `-m "not gpu"` tests (tests/emu/build.py compiles them with g++
// into tests/emu/_build/libnewton_emu.so).  One OS thread per GPU thread, one workgroup at a time; __syncthreads() is a
// std::barrier.  Nothing under newton_amd/ can load this library: the product path has no CPU fallback.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __align__(x) alignas(x)
#define __restrict__ __restrict
#define __shared__ static  // static LDS arrays: one workgroup runs at a time, so a single copy is the workgroup's copy

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct alignas(16) float4 { float x, y, z, w; };
typedef void* hipStream_t;
typedef void* hipGraph_t;      // (nt_graph.hip: handle types only -- the capture entry points answer NT_ERR_UNSUPPORTED under emulation)
typedef void* hipGraphExec_t;
enum hipError_t { hipSuccess = 0, hipErrorEmu = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace emu {
struct Block {
    std::unique_ptr<std::barrier<>> bar;       // __syncthreads
    std::vector<std::unique_ptr<std::barrier<>>> wave_bar;  // wave-level rendezvous (FS_WAVE_SYNC), sized on first use
    std::vector<float> lds;
    unsigned threads = 0;
};
inline Block*& current() {
    static Block* b = nullptr;
    return b;
}
inline float* dynamic_lds() { return current()->lds.data(); }
}  // namespace emu

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

inline void __syncthreads() {
    std::atomic_thread_fence(std::memory_order_seq_cst);
    emu::current()->bar->arrive_and_wait();
}
inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }
// rendezvous of `participants` lanes of the calling lane's wave (the Featherstone Cholesky wave)
inline void emu_wave_sync(unsigned participants) {
    emu::Block* b = emu::current();
    unsigned wave = threadIdx.x / 64;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    b->wave_bar[wave * 65 + participants]->arrive_and_wait();
}

// ---- wave-level intrinsics (nt_broadphase.hip): the 64 lanes of a wave are 64 OS threads that meet at a wave barrier.  Every
// lane of the wave must execute the call (the kernels only use them in wave-uniform control flow).
namespace emu {
struct WaveSlot {
    std::atomic<unsigned long long> mask{0};
    int values[64];
};
inline WaveSlot* wave_slots() {
    static WaveSlot slots[16];  // up to 1024 threads per workgroup
    return slots;
}
}  // namespace emu
inline unsigned __lane_id() { return threadIdx.x % 64; }
inline unsigned long long __ballot(int pred) {
    emu::WaveSlot& w = emu::wave_slots()[threadIdx.x / 64];
    const unsigned lanes = blockDim.x - (threadIdx.x / 64) * 64 < 64 ? blockDim.x - (threadIdx.x / 64) * 64 : 64;
    if (pred) w.mask.fetch_or(1ull << __lane_id());
    emu_wave_sync(lanes);
    unsigned long long m = w.mask.load();
    emu_wave_sync(lanes);
    if (__lane_id() == 0) w.mask.store(0);
    emu_wave_sync(lanes);
    return m;
}
inline int __any(int pred) { return __ballot(pred) != 0ull; }
inline int __shfl(int var, int src_lane) {
    emu::WaveSlot& w = emu::wave_slots()[threadIdx.x / 64];
    const unsigned lanes = blockDim.x - (threadIdx.x / 64) * 64 < 64 ? blockDim.x - (threadIdx.x / 64) * 64 : 64;
    w.values[__lane_id()] = var;
    emu_wave_sync(lanes);
    int v = w.values[src_lane];
    emu_wave_sync(lanes);
    return v;
}
inline int __popc(unsigned int x) { return __builtin_popcount(x); }
// width-limited forms: the wave is cut into segments of `width` lanes (a power of two), lane indices are segment-relative
inline int __shfl(int var, int src_lane, int width) {
    emu::WaveSlot& w = emu::wave_slots()[threadIdx.x / 64];
    const unsigned lanes = blockDim.x - (threadIdx.x / 64) * 64 < 64 ? blockDim.x - (threadIdx.x / 64) * 64 : 64;
    w.values[__lane_id()] = var;
    emu_wave_sync(lanes);
    int v = w.values[(__lane_id() / width) * width + (src_lane % width)];
    emu_wave_sync(lanes);
    return v;
}
inline int __shfl_up(int var, unsigned delta, int width) {
    emu::WaveSlot& w = emu::wave_slots()[threadIdx.x / 64];
    const unsigned lanes = blockDim.x - (threadIdx.x / 64) * 64 < 64 ? blockDim.x - (threadIdx.x / 64) * 64 : 64;
    w.values[__lane_id()] = var;
    emu_wave_sync(lanes);
    int v = (__lane_id() % width) >= delta ? w.values[__lane_id() - delta] : var;
    emu_wave_sync(lanes);
    return v;
}
inline int __shfl_up(int var, unsigned delta) {
    emu::WaveSlot& w = emu::wave_slots()[threadIdx.x / 64];
    const unsigned lanes = blockDim.x - (threadIdx.x / 64) * 64 < 64 ? blockDim.x - (threadIdx.x / 64) * 64 : 64;
    w.values[__lane_id()] = var;
    emu_wave_sync(lanes);
    int v = __lane_id() >= delta ? w.values[__lane_id() - delta] : var;
    emu_wave_sync(lanes);
    return v;
}
inline float emu_rcpf(float x) { return 1.0f / x; }
inline int emu_uniform(int x) { return x; }  // __builtin_amdgcn_readfirstlane of a wave-uniform value
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline float atomicAdd(float* p, float v) {  // global_atomic_add_f32
    float old = *p, want;
    do { want = old + v; } while (!__atomic_compare_exchange(p, &old, &want, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
    return old;
}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {  // ds_max_u64 / global_atomic_umax_x2
    unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

inline unsigned int atomicMin(unsigned int* p, unsigned int v) {  // ds_min_u32
    unsigned int old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {  // global_atomic_umin_x2
    unsigned long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
inline unsigned int __float_as_uint(float f) { unsigned int u; __builtin_memcpy(&u, &f, 4); return u; }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

template <typename K>
inline hipError_t hipFuncSetAttribute(K, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorEmu; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) { memcpy(dst, src, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return hipSuccess;
}
template <typename T>
inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
template <typename T>
inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy(&sym, src, n); return hipSuccess; }

template <typename K, typename... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t, A... args) {
    emu::Block blk;
    blk.threads = block.x;
    blk.lds.assign(lds_bytes / 4 + 64, 0.0f);
    const unsigned waves = (block.x + 63) / 64;
    blk.wave_bar.resize(waves * 65);
    for (unsigned w = 0; w < waves; ++w)
        for (unsigned p = 1; p <= 64; ++p) blk.wave_bar[w * 65 + p] = std::make_unique<std::barrier<>>(p);
    emu::current() = &blk;
    for (unsigned b = 0; b < grid.x; ++b) {
        blk.bar = std::make_unique<std::barrier<>>(block.x);
        std::vector<std::thread> th;
        th.reserve(block.x);
        for (unsigned t = 0; t < block.x; ++t)
            th.emplace_back([&, t, b] {
                threadIdx = dim3(t);
                blockIdx = dim3(b);
                blockDim = block;
                gridDim = grid;
                kernel(args...);
                blk.bar->arrive_and_drop();  // a lane that returned early must not block the others' barriers
            });
        for (auto& x : th) x.join();
    }
    emu::current() = nullptr;
}
