// Standalone world-aware broad phases on arbitrary AABB arrays for gfx950: the drop-in for
// newton.geometry.BroadPhaseAllPairs / BroadPhaseSAP / BroadPhaseExplicit
// (newton/_src/geometry/broad_phase_nxn.py:29-218,221-535; broad_phase_sap.py:44-848).
//
// Layout: the host groups the colliding shapes by world (precompute_world_map: every world's shapes followed by the
// shared world -1 shapes, plus one trailing segment with only the shared shapes).  One lane owns one position t of that
// map and tests its shape against the LATER positions of the same segment, so every unordered pair of a segment is
// visited once; the 64 lanes of a wave walk the same segment, their j-loads hit the same cache lines (broadcast), and
// the i-side AABB stays in registers.  SAP walks a map sorted by the x interval start inside each segment and stops at the
// first later shape whose interval starts past its own end.  Candidate pairs are appended through one wave-aggregated
// atomic per wave and iteration; the counter keeps counting past capacity like the reference (broad_phase_common.py:204-218).
// Swept mode (the *_swept entry points, include/newton_hip_broadphase.h): with a per-shape displacement the pair test is
// check_aabb_overlap_moving (broad_phase_common.py:41-85) and the SAP interval of a shape is extended by its (capped)
// displacement along the sort axis (_sap_project_aabb, broad_phase_sap.py:44-79) -- the speculative-contact broad phase.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"
#include "../../include/newton_hip_broadphase.h"

#define NT_BP_HD __host__ __device__
#include "nt_broadphase_core.hpp"

namespace {

struct BpArgs {
    BpView v;
    const int32_t* map;         // index map (N x N) or the per-segment x-sorted map (SAP)
    const int32_t* slice_ends;  // [segments]
    int32_t segments, num_regular, map_len;
    int32_t* pairs;  // [cap][2]
    int32_t* count;  // [1]
    int32_t cap;
};

// append `hit` lanes' pairs with one atomic per wave
__device__ inline void bp_append(bool hit, int s1, int s2, int32_t* pairs, int32_t* count, int32_t cap) {
    unsigned long long mask = __ballot(hit);
    if (mask == 0ull) return;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(mask));
    base = __shfl(base, leader);
    if (hit) {
        int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (idx < cap) {
            pairs[2 * idx] = s1;
            pairs[2 * idx + 1] = s2;
        }
    }
}

template <bool SAP>
__global__ void __launch_bounds__(256) broadphase_segment_kernel(BpArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = t < a.map_len;
    int seg_end = 0, si = 0;
    bool dedicated = false;
    float hi_i = 0.0f;
    if (active) {
        int seg = bp_segment_of(a.slice_ends, a.segments, t);
        seg_end = a.slice_ends[seg];
        dedicated = seg >= a.num_regular;
        si = a.map[t];
        if (SAP) hi_i = bp_sap_hi(a.v, si);
    }
    // lanes of a wave iterate in lock step so that the ballot in bp_append sees all of them
    int q = t + 1;
    bool running = active && q < seg_end;
    while (__any(running)) {
        bool hit = false;
        int s1 = 0, s2 = 0;
        if (running) {
            int sj = a.map[q];
            if (SAP && bp_sap_past(bp_sap_lo(a.v, sj), hi_i)) {
                running = false;
            } else {
                hit = bp_candidate(a.v, si, sj, dedicated, s1, s2);
                q += 1;
                if (q >= seg_end) running = false;
            }
        }
        bp_append(hit, s1, s2, a.pairs, a.count, a.cap);
    }
}

// ------------------------------------------------------------------------------------------------
// Device SAP (broad_phase_sap.py:44-79 projection, :787-811 per-world sort, :221-300 range + sweep): one workgroup per world
// segment projects the gap-widened AABBs on the reference's fixed axis normalize(0.5935, 0.7790, 0.1235)
// (_sap_project_aabb: centre +- |d| . (half_size + gap)), sorts (projection_lower, map position) with a bitonic network in
// LDS (padding keys 1e30 like the reference's tile sort) and writes the sorted map plus the projected interval of every
// sorted position; the sweep kernel below then walks forward until a later interval starts past its own end.
// ------------------------------------------------------------------------------------------------
constexpr int SAP_TILE = 4096;  // shapes per world segment that fit the LDS sort (32 KB of keys + positions)

__device__ inline void sap_project(const BpView& v, int s, float limit, float& lo, float& hi) {
    const float dx = 0.5935f, dy = 0.7790f, dz = 0.1235f;
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float ax = dx * inv, ay = dy * inv, az = dz * inv;
    const float* l = v.lower + 3 * s;
    const float* u = v.upper + 3 * s;
    const float g = v.gap ? v.gap[s] : 0.0f;
    const float hx = 0.5f * (u[0] - l[0]) + g, hy = 0.5f * (u[1] - l[1]) + g, hz = 0.5f * (u[2] - l[2]) + g;
    const float radius = fabsf(ax) * hx + fabsf(ay) * hy + fabsf(az) * hz;
    const float center = ax * (0.5f * (l[0] + u[0])) + ay * (0.5f * (l[1] + u[1])) + az * (0.5f * (l[2] + u[2]));
    lo = center - radius;
    hi = center + radius;
    if (v.displacement) {  // swept interval: the shape's own motion along the axis, capped by sort_axis_displacement_limit
        const float* d = v.displacement + 3 * s;
        float pd = ax * d[0] + ay * d[1] + az * d[2];
        if (limit >= 0.0f) pd = fminf(fmaxf(pd, -limit), limit);
        lo += fminf(pd, 0.0f);
        hi += fmaxf(pd, 0.0f);
    }
}

__global__ void __launch_bounds__(256) sap_sort_kernel(BpView v, const int32_t* __restrict__ map, const int32_t* __restrict__ slice_ends,
                                                       int32_t* __restrict__ sorted_map, float* __restrict__ proj /*[2][map_len]*/,
                                                       int map_len, float limit) {
    __shared__ float key[SAP_TILE];
    __shared__ int pos[SAP_TILE];
    const int seg = blockIdx.x;
    const int begin = seg == 0 ? 0 : slice_ends[seg - 1], end = slice_ends[seg];
    const int n = end - begin;
    if (n <= 0) return;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        float lo = 1.0e30f, hi;
        if (i < n) sap_project(v, map[begin + i], limit, lo, hi);
        key[i] = lo;
        pos[i] = i;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const float ki = key[i], kl = key[l];
                    const int pi = pos[i], pl = pos[l];
                    const bool gt = ki > kl || (ki == kl && pi > pl);  // (key, original position): a stable order
                    if (gt == up) {
                        key[i] = kl; key[l] = ki;
                        pos[i] = pl; pos[l] = pi;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int s = map[begin + pos[i]];
        float lo, hi;
        sap_project(v, s, limit, lo, hi);
        sorted_map[begin + i] = s;
        proj[begin + i] = lo;
        proj[map_len + begin + i] = hi;
    }
}

__global__ void __launch_bounds__(256) sap_sweep_kernel(BpArgs a, const float* __restrict__ proj) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = t < a.map_len;
    int seg_end = 0, si = 0;
    bool dedicated = false;
    float hi_i = 0.0f;
    if (active) {
        int seg = bp_segment_of(a.slice_ends, a.segments, t);
        seg_end = a.slice_ends[seg];
        dedicated = seg >= a.num_regular;
        si = a.map[t];
        hi_i = proj[a.map_len + t];
    }
    int q = t + 1;
    bool running = active && q < seg_end;
    while (__any(running)) {
        bool hit = false;
        int s1 = 0, s2 = 0;
        if (running) {
            if (bp_sap_past(proj[q], hi_i)) {
                running = false;
            } else {
                hit = bp_candidate(a.v, si, a.map[q], dedicated, s1, s2);
                q += 1;
                if (q >= seg_end) running = false;
            }
        }
        bp_append(hit, s1, s2, a.pairs, a.count, a.cap);
    }
}

// _nxn_broadphase_precomputed_pairs (broad_phase_nxn.py:29-69): one lane per listed pair
__global__ void __launch_bounds__(256) broadphase_explicit_kernel(BpView v, const int32_t* list, int n_pairs, int32_t* pairs,
                                                                  int32_t* count, int32_t cap) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    int s1 = 0, s2 = 0;
    if (t < n_pairs) {
        s1 = list[2 * t];
        s2 = list[2 * t + 1];
        hit = !bp_immovable_filtered(v, s1, s2) && bp_overlap_moving(v, s1, s2);
    }
    bp_append(hit, s1, s2, pairs, count, cap);
}

bool bp_in_ok(const nt_broadphase_in* in) {
    return in && in->lower && in->upper && in->group && in->world && in->num_filter_pairs >= 0 &&
           (in->num_filter_pairs == 0 || in->filter_pairs);
}

BpView make_view(const nt_broadphase_in* in, const nt_broadphase_motion* motion = nullptr) {
    BpView v;
    v.displacement = motion ? motion->displacement : nullptr;
    v.lower = in->lower; v.upper = in->upper; v.gap = in->gap; v.group = in->group; v.world = in->world;
    v.filter_pairs = in->filter_pairs; v.num_filter_pairs = in->num_filter_pairs;
    v.shape_body = in->shape_body; v.body_flags = in->body_flags;
    v.include_static_kinematic_pairs = in->include_static_kinematic_pairs;
    return v;
}

template <bool SAP>
nt_status launch_segments(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* map, const int32_t* slice_ends, int32_t segments,
                          int32_t num_regular, int32_t map_len, int32_t* pairs, int32_t* count, int32_t cap, void* stream) {
    if (!bp_in_ok(in) || !count || cap < 0 || (cap > 0 && !pairs) || segments < 0 || map_len < 0) return NT_ERR_INVALID_ARG;
    if (map_len == 0 || segments == 0) return NT_OK;
    if (!map || !slice_ends) return NT_ERR_INVALID_ARG;
    BpArgs a;
    a.v = make_view(in, motion);
    a.map = map; a.slice_ends = slice_ends; a.segments = segments; a.num_regular = num_regular; a.map_len = map_len;
    a.pairs = pairs; a.count = count; a.cap = cap;
    hipLaunchKernelGGL(broadphase_segment_kernel<SAP>, dim3((map_len + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status sap_device(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* index_map,
                     const int32_t* slice_ends, int32_t segments, int32_t num_regular_worlds, int32_t map_len, int32_t max_segment,
                     int32_t* sorted_map, float* projections, int32_t* pairs, int32_t* count, int32_t cap, void* stream) {
    if (!bp_in_ok(in) || !count || cap < 0 || (cap > 0 && !pairs) || segments < 0 || map_len < 0) return NT_ERR_INVALID_ARG;
    if (map_len == 0 || segments == 0) return NT_OK;
    if (!index_map || !slice_ends || !sorted_map || !projections) return NT_ERR_INVALID_ARG;
    if (max_segment > SAP_TILE) return NT_ERR_UNSUPPORTED;  // a world with more shapes than the LDS sort tile
    BpArgs a;
    a.v = make_view(in, motion);
    a.map = sorted_map; a.slice_ends = slice_ends; a.segments = segments; a.num_regular = num_regular_worlds; a.map_len = map_len;
    a.pairs = pairs; a.count = count; a.cap = cap;
    const float limit = motion ? motion->sort_axis_displacement_limit : -1.0f;
    hipLaunchKernelGGL(sap_sort_kernel, dim3(segments), dim3(256), 0, (hipStream_t)stream, a.v, index_map, slice_ends, sorted_map,
                       projections, map_len, limit);
    hipLaunchKernelGGL(sap_sweep_kernel, dim3((map_len + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, projections);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status explicit_pairs(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* pair_list, int32_t n_pairs,
                         int32_t* pairs, int32_t* count, int32_t cap, void* stream) {
    if (!in || !in->lower || !in->upper || !count || n_pairs < 0 || cap < 0 || (cap > 0 && !pairs)) return NT_ERR_INVALID_ARG;
    if (n_pairs == 0) return NT_OK;
    if (!pair_list) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(broadphase_explicit_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       make_view(in, motion), pair_list, n_pairs, pairs, count, cap);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

bool motion_ok(const nt_broadphase_motion* m) { return m && m->displacement; }

}  // namespace

extern "C" {

nt_status nt_broadphase_nxn(const nt_broadphase_in* in, const int32_t* index_map, const int32_t* slice_ends, int32_t segments,
                            int32_t num_regular_worlds, int32_t map_len, int32_t* pairs, int32_t* count, int32_t cap,
                            void* stream) {
    return launch_segments<false>(in, nullptr, index_map, slice_ends, segments, num_regular_worlds, map_len, pairs, count, cap, stream);
}

nt_status nt_broadphase_sap(const nt_broadphase_in* in, const int32_t* sorted_map, const int32_t* slice_ends, int32_t segments,
                            int32_t num_regular_worlds, int32_t map_len, int32_t* pairs, int32_t* count, int32_t cap,
                            void* stream) {
    return launch_segments<true>(in, nullptr, sorted_map, slice_ends, segments, num_regular_worlds, map_len, pairs, count, cap, stream);
}

nt_status nt_broadphase_sap_device(const nt_broadphase_in* in, const int32_t* index_map, const int32_t* slice_ends,
                                   int32_t segments, int32_t num_regular_worlds, int32_t map_len, int32_t max_segment,
                                   int32_t* sorted_map, float* projections, int32_t* pairs, int32_t* count, int32_t cap,
                                   void* stream) {
    return sap_device(in, nullptr, index_map, slice_ends, segments, num_regular_worlds, map_len, max_segment, sorted_map,
                      projections, pairs, count, cap, stream);
}

nt_status nt_broadphase_explicit(const nt_broadphase_in* in, const int32_t* pair_list, int32_t n_pairs, int32_t* pairs,
                                 int32_t* count, int32_t cap, void* stream) {
    return explicit_pairs(in, nullptr, pair_list, n_pairs, pairs, count, cap, stream);
}

// ---- swept variants (include/newton_hip_broadphase.h) ----
nt_status nt_broadphase_nxn_swept(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* index_map,
                                  const int32_t* slice_ends, int32_t segments, int32_t num_regular_worlds, int32_t map_len,
                                  int32_t* pairs, int32_t* count, int32_t cap, void* stream) {
    if (!motion_ok(motion)) return NT_ERR_INVALID_ARG;
    return launch_segments<false>(in, motion, index_map, slice_ends, segments, num_regular_worlds, map_len, pairs, count, cap, stream);
}

nt_status nt_broadphase_sap_device_swept(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* index_map,
                                         const int32_t* slice_ends, int32_t segments, int32_t num_regular_worlds, int32_t map_len,
                                         int32_t max_segment, int32_t* sorted_map, float* projections, int32_t* pairs,
                                         int32_t* count, int32_t cap, void* stream) {
    if (!motion_ok(motion)) return NT_ERR_INVALID_ARG;
    return sap_device(in, motion, index_map, slice_ends, segments, num_regular_worlds, map_len, max_segment, sorted_map,
                      projections, pairs, count, cap, stream);
}

nt_status nt_broadphase_explicit_swept(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* pair_list,
                                       int32_t n_pairs, int32_t* pairs, int32_t* count, int32_t cap, void* stream) {
    if (!motion_ok(motion)) return NT_ERR_INVALID_ARG;
    return explicit_pairs(in, motion, pair_list, n_pairs, pairs, count, cap, stream);
}

}  // extern "C"
