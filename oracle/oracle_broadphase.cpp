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
};

bool check_aabb_overlap(const float* l1, const float* u1, float c1, const float* l2, const float* u2, float c2) {
    float cutoff_combined = c1 + c2;
    return l1[0] <= u2[0] + cutoff_combined && u1[0] >= l2[0] - cutoff_combined && l1[1] <= u2[1] + cutoff_combined &&
           u1[1] >= l2[1] - cutoff_combined && l1[2] <= u2[2] + cutoff_combined && u1[2] >= l2[2] - cutoff_combined;
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
    if (!check_aabb_overlap(v.lower + 3 * shape1, v.upper + 3 * shape1, gap1, v.lower + 3 * shape2, v.upper + 3 * shape2, gap2))
        return;
    if (v.nf > 0 && is_pair_excluded(v, shape1, shape2)) return;
    w.push(shape1, shape2);
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
