// nt_match.hip -- frame-to-frame contact matching (newton/_src/geometry/contact_match.py:266-391,442-480) on the fixed-slot
// contact layout.
//
// The reference sorts the flat contact list by (shape0, shape1, sub-key), binary-searches the previous frame's sorted keys for
// the pair's range and lets the new contacts of a pair race for their closest previous contact with a packed atomic_min.  Here
// a pair's contacts of one environment always live in the same `cpp` slots, so the pair range IS the slot group: one lane per
// (env, slot) scans the <= 5 saved midpoints of its own pair, and the race is resolved without atomics -- every lane re-derives
// the claims of its (<= 4) siblings and the winner is the reference's: smallest distance, ties by the smaller sub-contact index
// (the low bits of the sort key, contact_data.py:60-90).  Results per slot: index of the matched previous SLOT, -1 (pair had no
// contacts last frame), -2 (no candidate within the thresholds, or lost the race).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"
#include "nt_math.hpp"

using namespace nt;

namespace {

constexpr int MATCH_NOT_FOUND = -1, MATCH_BROKEN = -2;

struct MatchArgs {
    nt_model m;
    nt_state s;
    nt_contacts c;
    nt_contact_history h;
    float pos_threshold_sq, normal_dot_threshold;
    const uint8_t* reset_world_mask;
    int32_t* match_index;
};

__device__ inline vec3 ld3(const float* base, int comp0, int n, int slot, int ES, int env) {
    return vec3(base[((size_t)(comp0 + 0) * n + slot) * ES + env], base[((size_t)(comp0 + 1) * n + slot) * ES + env],
                base[((size_t)(comp0 + 2) * n + slot) * ES + env]);
}
__device__ inline xform body_xform(const nt_state& s, int nb, int b, int ES, int env) {
    const float* q = s.body_q;
    auto g = [&](int comp) { return q[((size_t)comp * nb + b) * ES + env]; };
    return xform(vec3(g(0), g(1), g(2)), quat(g(3), g(4), g(5), g(6)));
}
// world-space midpoint of the two contact points of a slot (the quantity the reference persists and compares)
__device__ inline vec3 midpoint(const MatchArgs& a, int slot, int env, int p) {
    const int ES = a.m.env_stride, ncs = a.m.np * a.m.cpp;
    int sa = a.m.pair_a[p], sb = a.m.pair_b[p];
    if (a.m.shape_type[sa] > a.m.shape_type[sb]) { int t = sa; sa = sb; sb = t; }  // contacts are written type-sorted
    const int ba = a.m.shape_body[sa], bb = a.m.shape_body[sb];
    vec3 p0 = ld3(a.c.data, 0, ncs, slot, ES, env), p1 = ld3(a.c.data, 3, ncs, slot, ES, env);
    if (ba >= 0) p0 = xform_point(body_xform(a.s, a.m.nb, ba, ES, env), p0);
    if (bb >= 0) p1 = xform_point(body_xform(a.s, a.m.nb, bb, ES, env), p1);
    return 0.5f * (p0 + p1);
}

// best previous slot of new contact (p, k): closest saved midpoint within the position threshold whose normal agrees
__device__ inline int best_candidate(const MatchArgs& a, int p, int k, int env, float& best_dist_sq, bool& any_prev) {
    const int ES = a.m.env_stride, cpp = a.m.cpp, ncs = a.m.np * cpp;
    const int slot = p * cpp + k;
    const vec3 pos = midpoint(a, slot, env, p);
    const vec3 n = ld3(a.c.data, 12, ncs, slot, ES, env);
    int best = -1;
    best_dist_sq = a.pos_threshold_sq;
    any_prev = false;
    for (int j = 0; j < cpp; ++j) {
        const int ps = p * cpp + j;
        if (!a.h.prev_live[(size_t)ps * ES + env]) continue;
        any_prev = true;
        const vec3 d = pos - ld3(a.h.prev_pos_world, 0, ncs, ps, ES, env);
        const float dist_sq = dot(d, d);
        if (dist_sq <= best_dist_sq) {
            if (dot(n, ld3(a.h.prev_normal, 0, ncs, ps, ES, env)) >= a.normal_dot_threshold) {
                best_dist_sq = dist_sq;
                best = ps;
            }
        }
    }
    return best;
}

__global__ void __launch_bounds__(256) contacts_match_kernel(MatchArgs a) {
    const int ES = a.m.env_stride, cpp = a.m.cpp, ncs = a.m.np * cpp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ncs * ES) return;
    const int env = (int)(i % ES), slot = (int)(i / ES);
    if (env >= a.m.env_count) return;
    const size_t gi = (size_t)slot * ES + env;
    if (a.c.shape0[gi] < 0 || a.c.shape0[gi] == a.c.shape1[gi]) {  // dead slot: not a contact of this frame
        a.match_index[gi] = MATCH_NOT_FOUND;
        return;
    }
    if (a.reset_world_mask && a.reset_world_mask[env]) {  // contacts of a world that was just reset never match
        a.match_index[gi] = MATCH_NOT_FOUND;
        return;
    }
    const int p = slot / cpp, k = slot - p * cpp;
    float my_dist;
    bool any_prev;
    const int cand = best_candidate(a, p, k, env, my_dist, any_prev);
    if (!any_prev) { a.match_index[gi] = MATCH_NOT_FOUND; return; }
    if (cand < 0) { a.match_index[gi] = MATCH_BROKEN; return; }
    // the race for prev[cand]: a sibling wins with a smaller distance, or the same distance and a smaller sub-contact index
    bool lost = false;
    for (int k2 = 0; k2 < cpp && !lost; ++k2) {
        if (k2 == k) continue;
        const size_t g2 = (size_t)(p * cpp + k2) * ES + env;
        if (a.c.shape0[g2] < 0 || a.c.shape0[g2] == a.c.shape1[g2]) continue;
        float d2;
        bool ap;
        if (best_candidate(a, p, k2, env, d2, ap) != cand) continue;
        if (d2 < my_dist || (d2 == my_dist && k2 < k)) lost = true;
    }
    a.match_index[gi] = lost ? MATCH_BROKEN : cand;
}

// _save_sorted_state_kernel: persist this frame's midpoints, normals and live flags for the next match
__global__ void __launch_bounds__(256) contacts_save_kernel(MatchArgs a) {
    const int ES = a.m.env_stride, cpp = a.m.cpp, ncs = a.m.np * cpp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ncs * ES) return;
    const int env = (int)(i % ES), slot = (int)(i / ES);
    const size_t gi = (size_t)slot * ES + env;
    const bool live = env < a.m.env_count && a.c.shape0[gi] >= 0 && a.c.shape0[gi] != a.c.shape1[gi];
    a.h.prev_live[gi] = live ? 1 : 0;
    if (!live) return;
    const vec3 pos = midpoint(a, slot, env, slot / cpp);
    const vec3 n = ld3(a.c.data, 12, ncs, slot, ES, env);
    if (a.h.prev_body_frame) {  // sticky mode: the body-frame points and offsets of the record actually used this frame
        float* B = a.h.prev_body_frame;
        for (int comp = 0; comp < 12; ++comp) B[((size_t)comp * ncs + slot) * ES + env] = a.c.data[((size_t)comp * ncs + slot) * ES + env];
    }
    float* P = a.h.prev_pos_world;
    float* N = a.h.prev_normal;
    P[((size_t)0 * ncs + slot) * ES + env] = pos.x; P[((size_t)1 * ncs + slot) * ES + env] = pos.y; P[((size_t)2 * ncs + slot) * ES + env] = pos.z;
    N[((size_t)0 * ncs + slot) * ES + env] = n.x; N[((size_t)1 * ncs + slot) * ES + env] = n.y; N[((size_t)2 * ncs + slot) * ES + env] = n.z;
}

// _replay_matched_kernel (contact_match.py:530-562): a matched contact that still touches (fresh gap <= 0) keeps last
// frame's body-frame points / offsets and normal; everything else of the row is key-derived or a per-shape constant
__global__ void __launch_bounds__(256) contacts_replay_kernel(MatchArgs a) {
    const int ES = a.m.env_stride, cpp = a.m.cpp, ncs = a.m.np * cpp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ncs * ES) return;
    const int env = (int)(i % ES), slot = (int)(i / ES);
    if (env >= a.m.env_count) return;
    const size_t gi = (size_t)slot * ES + env;
    if (a.c.shape0[gi] < 0 || a.c.shape0[gi] == a.c.shape1[gi]) return;
    const int idx = a.match_index[gi];
    if (idx < 0) return;  // MATCH_NOT_FOUND or MATCH_BROKEN: keep the new frame's data
    const int p = slot / cpp;
    int sa = a.m.pair_a[p], sb = a.m.pair_b[p];
    if (a.m.shape_type[sa] > a.m.shape_type[sb]) { int t = sa; sa = sb; sb = t; }
    const int ba = a.m.shape_body[sa], bb = a.m.shape_body[sb];
    float* D = a.c.data;
    vec3 p0 = ld3(D, 0, ncs, slot, ES, env), p1 = ld3(D, 3, ncs, slot, ES, env);
    if (ba >= 0) p0 = xform_point(body_xform(a.s, a.m.nb, ba, ES, env), p0);
    if (bb >= 0) p1 = xform_point(body_xform(a.s, a.m.nb, bb, ES, env), p1);
    const vec3 n = ld3(D, 12, ncs, slot, ES, env);
    const float margins = D[((size_t)15 * ncs + slot) * ES + env] + D[((size_t)16 * ncs + slot) * ES + env];
    const float fresh_gap = dot(p1 - p0, n) - margins;
    if (fresh_gap > 0.0f) return;
    const float* B = a.h.prev_body_frame;
    for (int comp = 0; comp < 12; ++comp) D[((size_t)comp * ncs + slot) * ES + env] = B[((size_t)comp * ncs + idx) * ES + env];
    for (int comp = 0; comp < 3; ++comp) D[((size_t)(12 + comp) * ncs + slot) * ES + env] = a.h.prev_normal[((size_t)comp * ncs + idx) * ES + env];
}

bool args_ok(const nt_model* m, const nt_state* s, const nt_contacts* c, const nt_contact_history* h) {
    return m && s && c && h && m->env_count > 0 && m->np > 0 && s->body_q && c->shape0 && c->shape1 && c->data && h->prev_pos_world &&
           h->prev_normal && h->prev_live;
}

}  // namespace

extern "C" {

nt_status nt_contacts_match(const nt_model* m, const nt_state* s, const nt_contacts* c, const nt_contact_history* h,
                            float pos_threshold, float normal_dot_threshold, const uint8_t* reset_world_mask, int32_t* match_index,
                            void* stream) {
    if (!args_ok(m, s, c, h) || !match_index || !(pos_threshold >= 0.0f)) return NT_ERR_INVALID_ARG;
    MatchArgs a = {*m, *s, *c, *h, pos_threshold * pos_threshold, normal_dot_threshold, reset_world_mask, match_index};
    const size_t n = (size_t)m->np * m->cpp * m->env_stride;
    hipLaunchKernelGGL(contacts_match_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_contacts_replay_matched(const nt_model* m, const nt_state* s, nt_contacts* c, const nt_contact_history* h,
                                     const int32_t* match_index, void* stream) {
    if (!args_ok(m, s, c, h) || !match_index || !h->prev_body_frame) return NT_ERR_INVALID_ARG;
    MatchArgs a = {*m, *s, *c, *h, 0.0f, 0.0f, nullptr, const_cast<int32_t*>(match_index)};
    const size_t n = (size_t)m->np * m->cpp * m->env_stride;
    hipLaunchKernelGGL(contacts_replay_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_contacts_save_history(const nt_model* m, const nt_state* s, const nt_contacts* c, nt_contact_history* h, void* stream) {
    if (!args_ok(m, s, c, h)) return NT_ERR_INVALID_ARG;
    MatchArgs a = {*m, *s, *c, *h, 0.0f, 0.0f, nullptr, nullptr};
    const size_t n = (size_t)m->np * m->cpp * m->env_stride;
    hipLaunchKernelGGL(contacts_save_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
