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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"

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

// _nxn_broadphase_precomputed_pairs (broad_phase_nxn.py:29-69): one lane per listed pair
__global__ void __launch_bounds__(256) broadphase_explicit_kernel(BpView v, const int32_t* list, int n_pairs, int32_t* pairs,
                                                                  int32_t* count, int32_t cap) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    int s1 = 0, s2 = 0;
    if (t < n_pairs) {
        s1 = list[2 * t];
        s2 = list[2 * t + 1];
        hit = !bp_immovable_filtered(v, s1, s2) && bp_overlap(v, s1, s2);
    }
    bp_append(hit, s1, s2, pairs, count, cap);
}

bool bp_in_ok(const nt_broadphase_in* in) {
    return in && in->lower && in->upper && in->group && in->world && in->num_filter_pairs >= 0 &&
           (in->num_filter_pairs == 0 || in->filter_pairs);
}

BpView make_view(const nt_broadphase_in* in) {
    BpView v;
    v.lower = in->lower; v.upper = in->upper; v.gap = in->gap; v.group = in->group; v.world = in->world;
    v.filter_pairs = in->filter_pairs; v.num_filter_pairs = in->num_filter_pairs;
    v.shape_body = in->shape_body; v.body_flags = in->body_flags;
    v.include_static_kinematic_pairs = in->include_static_kinematic_pairs;
    return v;
}

template <bool SAP>
nt_status launch_segments(const nt_broadphase_in* in, const int32_t* map, const int32_t* slice_ends, int32_t segments,
                          int32_t num_regular, int32_t map_len, int32_t* pairs, int32_t* count, int32_t cap, void* stream) {
    if (!bp_in_ok(in) || !count || cap < 0 || (cap > 0 && !pairs) || segments < 0 || map_len < 0) return NT_ERR_INVALID_ARG;
    if (map_len == 0 || segments == 0) return NT_OK;
    if (!map || !slice_ends) return NT_ERR_INVALID_ARG;
    BpArgs a;
    a.v = make_view(in);
    a.map = map; a.slice_ends = slice_ends; a.segments = segments; a.num_regular = num_regular; a.map_len = map_len;
    a.pairs = pairs; a.count = count; a.cap = cap;
    hipLaunchKernelGGL(broadphase_segment_kernel<SAP>, dim3((map_len + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // namespace

extern "C" {

nt_status nt_broadphase_nxn(const nt_broadphase_in* in, const int32_t* index_map, const int32_t* slice_ends, int32_t segments,
                            int32_t num_regular_worlds, int32_t map_len, int32_t* pairs, int32_t* count, int32_t cap,
                            void* stream) {
    return launch_segments<false>(in, index_map, slice_ends, segments, num_regular_worlds, map_len, pairs, count, cap, stream);
}

nt_status nt_broadphase_sap(const nt_broadphase_in* in, const int32_t* sorted_map, const int32_t* slice_ends, int32_t segments,
                            int32_t num_regular_worlds, int32_t map_len, int32_t* pairs, int32_t* count, int32_t cap,
                            void* stream) {
    return launch_segments<true>(in, sorted_map, slice_ends, segments, num_regular_worlds, map_len, pairs, count, cap, stream);
}

nt_status nt_broadphase_explicit(const nt_broadphase_in* in, const int32_t* pair_list, int32_t n_pairs, int32_t* pairs,
                                 int32_t* count, int32_t cap, void* stream) {
    if (!in || !in->lower || !in->upper || !count || n_pairs < 0 || cap < 0 || (cap > 0 && !pairs)) return NT_ERR_INVALID_ARG;
    if (n_pairs == 0) return NT_OK;
    if (!pair_list) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(broadphase_explicit_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, make_view(in),
                       pair_list, n_pairs, pairs, count, cap);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
