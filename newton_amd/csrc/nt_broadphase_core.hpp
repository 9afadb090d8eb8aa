// Pair predicate shared by the standalone broad-phase kernels (nt_broadphase.hip).  Plain C++ so that the host self-check
// (tools/broadphase_host_check.cpp) can run the very same lane program on the CPU, without a GPU.
//   test_group_pair / test_world_and_group_pair   newton/_src/geometry/broad_phase_common.py:220-268
//   check_aabb_overlap                            broad_phase_common.py:20-38
//   check_aabb_overlap_moving                     broad_phase_common.py:41-85 (swept AABBs of the speculative-contact mode)
//   is_pair_excluded                              broad_phase_common.py:132-162
//   is_shape_pair_immovable_filtered              broad_phase_common.py:165-201
#pragma once
#include <stdint.h>

#ifndef NT_BP_HD
#define NT_BP_HD
#endif

struct BpView {
    const float* lower;  // [n][3]
    const float* upper;  // [n][3]
    const float* gap;    // [n] or nullptr (AABBs pre-expanded)
    const int32_t* group;
    const int32_t* world;
    const int32_t* filter_pairs;  // [nf][2] sorted lexicographically, canonical (min, max)
    int32_t num_filter_pairs;
    const int32_t* shape_body;  // nullptr: no immovable filtering
    const int32_t* body_flags;  // nullptr: static-static only
    int32_t include_static_kinematic_pairs;
    const float* displacement;  // [n][3] world-space motion over the collision-update interval, or nullptr (static test)
};

NT_BP_HD inline bool bp_group_pair(int a, int b) {
    if (a == 0 || b == 0) return false;
    if (a > 0) return a == b || b < 0;
    return a != b;
}

NT_BP_HD inline bool bp_world_and_group_pair(int wa, int wb, int ga, int gb) {
    if (wa != -1 && wb != -1 && wa != wb) return false;
    return bp_group_pair(ga, gb);
}

NT_BP_HD inline bool bp_excluded(const BpView& v, int s1, int s2) {
    int low = 0, high = v.num_filter_pairs - 1;
    while (low <= high) {
        int mid = (low + high) >> 1;
        int a = v.filter_pairs[2 * mid], b = v.filter_pairs[2 * mid + 1];
        if (a == s1 && b == s2) return true;
        if (s1 < a || (s1 == a && s2 < b)) high = mid - 1;
        else low = mid + 1;
    }
    return false;
}

NT_BP_HD inline bool bp_immovable_filtered(const BpView& v, int s1, int s2) {
    if (v.include_static_kinematic_pairs || !v.shape_body) return false;
    int ba = v.shape_body[s1], bb = v.shape_body[s2];
    bool static_a = ba < 0, static_b = bb < 0;
    if (static_a && static_b) return true;
    if (!v.body_flags) return false;
    bool kin_a = !static_a && (v.body_flags[ba] & 2) != 0;  // BodyFlags.KINEMATIC
    bool kin_b = !static_b && (v.body_flags[bb] & 2) != 0;
    return (static_a || kin_a) && (static_b || kin_b);
}

// AABB test with the operand order of the reference (box1 = the smaller shape index)
NT_BP_HD inline bool bp_overlap(const BpView& v, int s1, int s2) {
    float c = 0.0f;
    if (v.gap) c = v.gap[s1] + v.gap[s2];
    const float *l1 = v.lower + 3 * s1, *u1 = v.upper + 3 * s1, *l2 = v.lower + 3 * s2, *u2 = v.upper + 3 * s2;
    return l1[0] <= u2[0] + c && u1[0] >= l2[0] - c && l1[1] <= u2[1] + c && u1[1] >= l2[1] - c && l1[2] <= u2[2] + c &&
           u1[2] >= l2[2] - c;
}

// Swept test: do the two gap-widened boxes overlap at ONE time of the unit interval while each translates by its own
// displacement?  Slab clipping of the relative motion per axis, in the reference's operand order (box1 = the smaller index).
NT_BP_HD inline bool bp_overlap_moving(const BpView& v, int s1, int s2) {
    if (!v.displacement) return bp_overlap(v, s1, s2);
    float c = 0.0f;
    if (v.gap) c = v.gap[s1] + v.gap[s2];
    const float *l1 = v.lower + 3 * s1, *u1 = v.upper + 3 * s1, *l2 = v.lower + 3 * s2, *u2 = v.upper + 3 * s2;
    const float *d1 = v.displacement + 3 * s1, *d2 = v.displacement + 3 * s2;
    float enter = 0.0f, exit_time = 1.0f;
    for (int axis = 0; axis < 3; ++axis) {
        const float lower1 = l1[axis], upper1 = u1[axis];
        const float lower2 = l2[axis] - c, upper2 = u2[axis] + c;
        const float delta = d1[axis] - d2[axis];
        if (delta == 0.0f) {
            if (lower1 > upper2 || upper1 < lower2) return false;
        } else {
            float axis_enter = (lower2 - upper1) / delta, axis_exit = (upper2 - lower1) / delta;
            if (axis_enter > axis_exit) {
                const float tmp = axis_enter;
                axis_enter = axis_exit;
                axis_exit = tmp;
            }
            enter = enter > axis_enter ? enter : axis_enter;  // wp.max / wp.min
            exit_time = exit_time < axis_exit ? exit_time : axis_exit;
            if (enter > exit_time) return false;
        }
    }
    return true;
}

// full candidate test of the N x N / SAP kernels for two shapes of one world segment (broad_phase_nxn.py:172-218);
// on success (s1, s2) is the canonical pair
NT_BP_HD inline bool bp_candidate(const BpView& v, int sa, int sb, bool dedicated_global_segment, int& s1, int& s2) {
    s1 = sa < sb ? sa : sb;
    s2 = sa < sb ? sb : sa;
    int w1 = v.world[s1], w2 = v.world[s2];
    if (w1 == -1 && w2 == -1 && !dedicated_global_segment) return false;
    if (!bp_world_and_group_pair(w1, w2, v.group[s1], v.group[s2])) return false;
    if (bp_immovable_filtered(v, s1, s2)) return false;
    if (!bp_overlap_moving(v, s1, s2)) return false;
    if (v.num_filter_pairs > 0 && bp_excluded(v, s1, s2)) return false;
    return true;
}

// segment of map position t: first index whose slice end is > t
NT_BP_HD inline int bp_segment_of(const int32_t* slice_ends, int segments, int t) {
    int low = 0, high = segments - 1;
    while (low < high) {
        int mid = (low + high) >> 1;
        if (slice_ends[mid] > t) high = mid;
        else low = mid + 1;
    }
    return low;
}

// sweep key of the SAP variant: AABB interval on x widened by the shape's own gap
NT_BP_HD inline float bp_sap_lo(const BpView& v, int s) { return v.lower[3 * s] - (v.gap ? v.gap[s] : 0.0f); }
NT_BP_HD inline float bp_sap_hi(const BpView& v, int s) { return v.upper[3 * s] + (v.gap ? v.gap[s] : 0.0f); }
// conservative early-out of the sweep: later keys only grow, and the slack covers the different rounding of
// (l2 - (g1 + g2)) in the exact test vs (l2 - g2) in the key
NT_BP_HD inline bool bp_sap_past(float key_j, float hi_i) {
    float slack = 1e-5f * (1.0f + (hi_i < 0.0f ? -hi_i : hi_i) + (key_j < 0.0f ? -key_j : key_j));
    return key_j > hi_i + slack;
}
