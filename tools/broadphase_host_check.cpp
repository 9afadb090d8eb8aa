// Host self-check of the standalone broad-phase kernels' logic (no GPU needed): runs the SAME pair predicate, segment lookup
// and sweep early-out as csrc/nt_broadphase.hip (shared header nt_broadphase_core.hpp), one "lane" after the other, and
// prints the resulting pair list so that tests/test_broad_phase_standalone.py can compare it with the oracle.
//   stdin : n nf segments num_regular map_len mode(0 nxn, 1 sap) has_gap | lower[3n] upper[3n] gap[n]? group[n] world[n]
//           filter[2nf] map[map_len] slice_ends[segments]
//   stdout: count, then the pairs
#include <cstdio>
#include <vector>

#include "../newton_amd/csrc/nt_broadphase_core.hpp"

int main() {
    int n, nf, segments, num_regular, map_len, mode, has_gap;
    if (scanf("%d %d %d %d %d %d %d", &n, &nf, &segments, &num_regular, &map_len, &mode, &has_gap) != 7) return 1;
    std::vector<float> lower(3 * n), upper(3 * n), gap(n);
    std::vector<int32_t> group(n), world(n), filter(2 * nf + 2), map(map_len + 1), ends(segments + 1);
    for (auto& x : lower) if (scanf("%f", &x) != 1) return 1;
    for (auto& x : upper) if (scanf("%f", &x) != 1) return 1;
    if (has_gap) for (auto& x : gap) if (scanf("%f", &x) != 1) return 1;
    for (auto& x : group) if (scanf("%d", &x) != 1) return 1;
    for (auto& x : world) if (scanf("%d", &x) != 1) return 1;
    for (int i = 0; i < 2 * nf; ++i) if (scanf("%d", &filter[i]) != 1) return 1;
    for (int i = 0; i < map_len; ++i) if (scanf("%d", &map[i]) != 1) return 1;
    for (int i = 0; i < segments; ++i) if (scanf("%d", &ends[i]) != 1) return 1;
    BpView v{lower.data(), upper.data(), has_gap ? gap.data() : nullptr, group.data(), world.data(), filter.data(), nf,
             nullptr, nullptr, 1};
    std::vector<int> out;
    for (int t = 0; t < map_len; ++t) {  // one iteration per lane of broadphase_segment_kernel
        int seg = bp_segment_of(ends.data(), segments, t);
        int seg_end = ends[seg];
        bool dedicated = seg >= num_regular;
        int si = map[t];
        float hi_i = mode ? bp_sap_hi(v, si) : 0.0f;
        for (int q = t + 1; q < seg_end; ++q) {
            int sj = map[q], s1, s2;
            if (mode && bp_sap_past(bp_sap_lo(v, sj), hi_i)) break;
            if (bp_candidate(v, si, sj, dedicated, s1, s2)) { out.push_back(s1); out.push_back(s2); }
        }
    }
    printf("%zu\n", out.size() / 2);
    for (size_t i = 0; i < out.size(); i += 2) printf("%d %d\n", out[i], out[i + 1]);
    return 0;
}
