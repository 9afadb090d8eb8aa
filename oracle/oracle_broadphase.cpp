// CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Standalone broad phases on arbitrary AABB arrays, restating
//   precompute_world_map                      newton/_src/geometry/broad_phase_common.py:271-388
//   _nxn_broadphase_kernel                    newton/_src/geometry/broad_phase_nxn.py:132-218
//   _nxn_broadphase_precomputed_pairs         newton/_src/geometry/broad_phase_nxn.py:29-69
//   check_aabb_overlap / test_world_and_group_pair / is_pair_excluded / is_shape_pair_immovable_filtered / write_pair
//                                             newton/_src/geometry/broad_phase_common.py:20-38,132-268
// The reference's sort-and-sweep kernels (broad_phase_sap.py:44-848) emit the same pair SET as N x N (its tests compare both
// with one numpy brute force, newton/tests/test_broad_phase.py:399-2272); the SAP entry point below is a plain
// sort + sweep along x with the same pair predicate.  Parity unpinned at bit level (Warp builtins), pinned as exact integer
// sets by the restated reference tests in tests/test_broad_phase_standalone.py.
// Swept mode (o_broadphase_*_swept): check_aabb_overlap_moving (broad_phase_common.py:41-85) as the pair test, and for
// sort-and-sweep the reference's own criterion restated in full -- _sap_project_aabb (broad_phase_sap.py:44-79) on the fixed axis
// normalize(0.5935, 0.7790, 0.1235) with the capped displacement, sort by projected lower bound, binary_search_segment for the
// first later lower bound >= the shape's upper bound (:82-111,221-270) -- because with a capped displacement the projected
// intervals decide which pairs are tested at all.  Pinned by the executed reference classes
// (tests/golden/make_broadphase_reference_vectors.py, cases "swept/*").
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

struct View {
    const float *lower, *upper, *gap;
    const int32_t *group, *world, *filter_pairs;
    int nf;
    const int32_t *shape_body, *body_flags;
    int include_static_kinematic_pairs;
    const float* shape_displacement = nullptr;  // [n][3] or nullptr (the reference's empty array)
};

bool check_aabb_overlap(const float* l1, const float* u1, float c1, const float* l2, const float* u2, float c2) {
    float cutoff_combined = c1 + c2;
    return l1[0] <= u2[0] + cutoff_combined && u1[0] >= l2[0] - cutoff_combined && l1[1] <= u2[1] + cutoff_combined &&
           u1[1] >= l2[1] - cutoff_combined && l1[2] <= u2[2] + cutoff_combined && u1[2] >= l2[2] - cutoff_combined;
}

bool check_aabb_overlap_moving(const View& v, int shape1, int shape2, float cutoff1, float cutoff2) {
    const float* box_lower = v.lower;
    const float* box_upper = v.upper;
    if (!v.shape_displacement)
        return check_aabb_overlap(box_lower + 3 * shape1, box_upper + 3 * shape1, cutoff1, box_lower + 3 * shape2,
                                  box_upper + 3 * shape2, cutoff2);
    float cutoff_combined = cutoff1 + cutoff2;
    float enter = 0.0f, exit_time = 1.0f;
    for (int axis = 0; axis < 3; ++axis) {
        float relative_displacement = v.shape_displacement[3 * shape1 + axis] - v.shape_displacement[3 * shape2 + axis];
        float lower1 = box_lower[3 * shape1 + axis], upper1 = box_upper[3 * shape1 + axis];
        float lower2 = box_lower[3 * shape2 + axis] - cutoff_combined, upper2 = box_upper[3 * shape2 + axis] + cutoff_combined;
        float delta = relative_displacement;
        if (delta == 0.0f) {
            if (lower1 > upper2 || upper1 < lower2) return false;
        } else {
            float axis_enter = (lower2 - upper1) / delta, axis_exit = (upper2 - lower1) / delta;
            if (axis_enter > axis_exit) std::swap(axis_enter, axis_exit);
            enter = std::max(enter, axis_enter);
            exit_time = std::min(exit_time, axis_exit);
            if (enter > exit_time) return false;
        }
    }
    return true;
}

bool test_group_pair(int a, int b) {
    if (a == 0 || b == 0) return false;
    if (a > 0) return a == b || b < 0;
    return a != b;
}

bool test_world_and_group_pair(int wa, int wb, int ga, int gb) {
    if (wa != -1 && wb != -1 && wa != wb) return false;
    return test_group_pair(ga, gb);
}

bool is_pair_excluded(const View& v, int s1, int s2) {
    int low = 0, high = v.nf - 1;
    while (low <= high) {
        int mid = (low + high) >> 1;
        int a = v.filter_pairs[2 * mid], b = v.filter_pairs[2 * mid + 1];
        if (a == s1 && b == s2) return true;
        if (s1 < a || (s1 == a && s2 < b)) high = mid - 1;
        else low = mid + 1;
    }
    return false;
}

bool is_shape_pair_immovable_filtered(const View& v, int a, int b) {
    if (v.include_static_kinematic_pairs || !v.shape_body) return false;
    int body_a = v.shape_body[a], body_b = v.shape_body[b];
    bool static_a = body_a < 0, static_b = body_b < 0;
    if (static_a && static_b) return true;
    if (!v.body_flags) return false;
    bool kinematic_a = !static_a && (v.body_flags[body_a] & 2) != 0;
    bool kinematic_b = !static_b && (v.body_flags[body_b] & 2) != 0;
    return (static_a || kinematic_a) && (static_b || kinematic_b);
}

struct Writer {  // write_pair: the counter keeps counting past capacity
    int32_t* out;
    int cap, count = 0;
    void push(int a, int b) {
        int id = count++;
        if (id >= cap) return;
        out[2 * id] = a;
        out[2 * id + 1] = b;
    }
};

void test_and_write(const View& v, int sa, int sb, bool dedicated, Writer& w) {
    int shape1 = std::min(sa, sb), shape2 = std::max(sa, sb);
    int world1 = v.world[shape1], world2 = v.world[shape2];
    if (world1 == -1 && world2 == -1 && !dedicated) return;
    if (!test_world_and_group_pair(world1, world2, v.group[shape1], v.group[shape2])) return;
    if (is_shape_pair_immovable_filtered(v, shape1, shape2)) return;
    float gap1 = v.gap ? v.gap[shape1] : 0.0f, gap2 = v.gap ? v.gap[shape2] : 0.0f;
    if (!check_aabb_overlap_moving(v, shape1, shape2, gap1, gap2)) return;
    if (v.nf > 0 && is_pair_excluded(v, shape1, shape2)) return;
    w.push(shape1, shape2);
}

// _sap_project_aabb (broad_phase_sap.py:44-79)
void sap_project_aabb(const View& v, int elementid, const float direction[3], float sort_axis_displacement_limit, float& projection_lower,
                      float& projection_upper) {
    const float* lower = v.lower + 3 * elementid;
    const float* upper = v.upper + 3 * elementid;
    float gap = v.gap ? v.gap[elementid] : 0.0f;
    float half_size[3];
    for (int k = 0; k < 3; ++k) half_size[k] = 0.5f * (upper[k] - lower[k]) + gap;
    float radius = std::fabs(direction[0]) * half_size[0] + std::fabs(direction[1]) * half_size[1] + std::fabs(direction[2]) * half_size[2];
    float center = direction[0] * (0.5f * (lower[0] + upper[0])) + direction[1] * (0.5f * (lower[1] + upper[1])) +
                   direction[2] * (0.5f * (lower[2] + upper[2]));
    projection_lower = center - radius;
    projection_upper = center + radius;
    if (v.shape_displacement) {
        const float* d = v.shape_displacement + 3 * elementid;
        float projected_displacement = direction[0] * d[0] + direction[1] * d[1] + direction[2] * d[2];
        if (sort_axis_displacement_limit >= 0.0f)
            projected_displacement = std::min(std::max(projected_displacement, -sort_axis_displacement_limit), sort_axis_displacement_limit);
        projection_lower += std::min(projected_displacement, 0.0f);
        projection_upper += std::max(projected_displacement, 0.0f);
    }
}

}  // namespace

extern "C" {

// returns the number of candidate pairs found (may exceed cap); pairs in the order a serial launch appends them
int o_broadphase_nxn(const float* lower, const float* upper, const float* gap, const int32_t* group, const int32_t* world,
                     const int32_t* index_map, const int32_t* slice_ends, int segments, int num_regular_worlds,
                     const int32_t* filter_pairs, int num_filter_pairs, const int32_t* shape_body, const int32_t* body_flags,
                     int include_static_kinematic_pairs, int32_t* out_pairs, int cap) {
    View v{lower, upper, gap, group, world, filter_pairs, num_filter_pairs, shape_body, body_flags, include_static_kinematic_pairs};
    Writer w{out_pairs, cap};
    int start = 0;
    for (int seg = 0; seg < segments; ++seg) {
        int end = slice_ends[seg];
        for (int r = start; r < end; ++r)  // local lower-triangular enumeration == (r, c) lexicographic
            for (int c = r + 1; c < end; ++c) test_and_write(v, index_map[r], index_map[c], seg >= num_regular_worlds, w);
        start = end;
    }
    return w.count;
}

int o_broadphase_sap(const float* lower, const float* upper, const float* gap, const int32_t* group, const int32_t* world,
                     const int32_t* index_map, const int32_t* slice_ends, int segments, int num_regular_worlds,
                     const int32_t* filter_pairs, int num_filter_pairs, const int32_t* shape_body, const int32_t* body_flags,
                     int include_static_kinematic_pairs, int32_t* out_pairs, int cap) {
    View v{lower, upper, gap, group, world, filter_pairs, num_filter_pairs, shape_body, body_flags, include_static_kinematic_pairs};
    Writer w{out_pairs, cap};
    int start = 0;
    for (int seg = 0; seg < segments; ++seg) {
        int end = slice_ends[seg];
        std::vector<int> order(index_map + start, index_map + end);
        auto lo = [&](int s) { return double(lower[3 * s]) - double(gap ? gap[s] : 0.0f); };
        auto hi = [&](int s) { return double(upper[3 * s]) + double(gap ? gap[s] : 0.0f); };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lo(a) < lo(b); });
        for (size_t i = 0; i < order.size(); ++i)
            for (size_t j = i + 1; j < order.size(); ++j) {
                if (lo(order[j]) > hi(order[i]) + 1e-4 * (1.0 + std::abs(hi(order[i])))) break;  // conservative sweep end
                test_and_write(v, order[i], order[j], seg >= num_regular_worlds, w);
            }
        start = end;
    }
    return w.count;
}

// ---- swept variants ----
int o_broadphase_nxn_swept(const float* lower, const float* upper, const float* gap, const int32_t* group, const int32_t* world,
                           const int32_t* index_map, const int32_t* slice_ends, int segments, int num_regular_worlds,
                           const int32_t* filter_pairs, int num_filter_pairs, const int32_t* shape_body, const int32_t* body_flags,
                           int include_static_kinematic_pairs, const float* shape_displacement, int32_t* out_pairs, int cap) {
    View v{lower, upper, gap, group, world, filter_pairs, num_filter_pairs, shape_body, body_flags, include_static_kinematic_pairs,
           shape_displacement};
    Writer w{out_pairs, cap};
    int start = 0;
    for (int seg = 0; seg < segments; ++seg) {
        int end = slice_ends[seg];
        for (int r = start; r < end; ++r)
            for (int c = r + 1; c < end; ++c) test_and_write(v, index_map[r], index_map[c], seg >= num_regular_worlds, w);
        start = end;
    }
    return w.count;
}

// BroadPhaseSAP.launch with the reference's projection criterion (see the header comment); pairs in sweep order
int o_broadphase_sap_swept(const float* lower, const float* upper, const float* gap, const int32_t* group, const int32_t* world,
                           const int32_t* index_map, const int32_t* slice_ends, int segments, int num_regular_worlds,
                           const int32_t* filter_pairs, int num_filter_pairs, const int32_t* shape_body, const int32_t* body_flags,
                           int include_static_kinematic_pairs, const float* shape_displacement, float sort_axis_displacement_limit,
                           int32_t* out_pairs, int cap) {
    View v{lower, upper, gap, group, world, filter_pairs, num_filter_pairs, shape_body, body_flags, include_static_kinematic_pairs,
           shape_displacement};
    Writer w{out_pairs, cap};
    const float raw[3] = {0.5935f, 0.7790f, 0.1235f};  // broad_phase_sap.py:700-703
    const float len = std::sqrt(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2]);
    const float direction[3] = {raw[0] / len, raw[1] / len, raw[2] / len};
    int start = 0;
    for (int seg = 0; seg < segments; ++seg) {
        int end = slice_ends[seg], n = end - start;
        std::vector<float> plo(n), phi(n);
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) {
            sap_project_aabb(v, index_map[start + i], direction, sort_axis_displacement_limit, plo[i], phi[i]);
            order[i] = i;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return plo[a] < plo[b]; });
        for (int i = 0; i < n; ++i) {
            float upper_i = phi[order[i]];
            for (int j = i + 1; j < n && plo[order[j]] < upper_i; ++j)  // binary_search_segment: first lower >= upper ends the range
                test_and_write(v, index_map[start + order[i]], index_map[start + order[j]], seg >= num_regular_worlds, w);
        }
        start = end;
    }
    return w.count;
}

int o_broadphase_explicit_swept(const float* lower, const float* upper, const float* gap, const int32_t* pair_list, int n_pairs,
                                const int32_t* shape_body, const int32_t* body_flags, int include_static_kinematic_pairs,
                                const float* shape_displacement, int32_t* out_pairs, int cap) {
    View v{lower, upper, gap, nullptr, nullptr, nullptr, 0, shape_body, body_flags, include_static_kinematic_pairs, shape_displacement};
    Writer w{out_pairs, cap};
    for (int e = 0; e < n_pairs; ++e) {
        int shape1 = pair_list[2 * e], shape2 = pair_list[2 * e + 1];
        if (is_shape_pair_immovable_filtered(v, shape1, shape2)) continue;
        float gap1 = gap ? gap[shape1] : 0.0f, gap2 = gap ? gap[shape2] : 0.0f;
        if (check_aabb_overlap_moving(v, shape1, shape2, gap1, gap2)) w.push(shape1, shape2);
    }
    return w.count;
}

int o_broadphase_explicit(const float* lower, const float* upper, const float* gap, const int32_t* pair_list, int n_pairs,
                          const int32_t* shape_body, const int32_t* body_flags, int include_static_kinematic_pairs,
                          int32_t* out_pairs, int cap) {
    View v{lower, upper, gap, nullptr, nullptr, nullptr, 0, shape_body, body_flags, include_static_kinematic_pairs};
    Writer w{out_pairs, cap};
    for (int e = 0; e < n_pairs; ++e) {
        int shape1 = pair_list[2 * e], shape2 = pair_list[2 * e + 1];
        if (is_shape_pair_immovable_filtered(v, shape1, shape2)) continue;
        float gap1 = gap ? gap[shape1] : 0.0f, gap2 = gap ? gap[shape2] : 0.0f;
        if (check_aabb_overlap(lower + 3 * shape1, upper + 3 * shape1, gap1, lower + 3 * shape2, upper + 3 * shape2, gap2))
            w.push(shape1, shape2);
    }
    return w.count;
}

}  // extern "C"
