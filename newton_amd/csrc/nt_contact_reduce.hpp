// nt_contact_reduce.hpp -- the global contact reduction's building blocks in LDS (contact_reduction_global.py), shared by the
// translation units that own a shape pair per workgroup: nt_sdf.hip (mesh-SDF edge contacts: the centred two-depth variant,
// red_offer) and nt_mesh_plane.hip (mesh vertices vs infinite planes: the buffered variant, reduce_contact_in_hashtable).
// Include it INSIDE the including unit's anonymous namespace, after nt_math.hpp (`using namespace nt`).  The design comment
// lives at the head of the reduction section of nt_sdf.hip.
#pragma once

constexpr int RED_BINS = 20, RED_DIRS = 6, RED_VALUES = 7, RED_VOXELS = 100;
constexpr int RED_ENTRIES = RED_BINS + (RED_VOXELS + RED_VALUES - 1) / RED_VALUES;  // 35
constexpr int RED_SLOTS = RED_ENTRIES * RED_VALUES;                                 // 245
constexpr unsigned long long RED_FP_MASK = (1ull << 22) - 1;

__device__ const float RED_FACE[RED_BINS][3] = {  // contact_reduction.py:170-191 (icosahedron)
    {0.49112338f, 0.79465455f, 0.35682216f},   {-0.18759243f, 0.79465450f, 0.57735026f},  {-0.60706190f, 0.79465450f, 0.0f},
    {-0.18759237f, 0.79465450f, -0.57735026f}, {0.49112340f, 0.79465455f, -0.35682210f},  {0.98224690f, -0.18759257f, 0.0f},
    {0.79465440f, 0.18759239f, -0.57735030f},  {0.30353096f, -0.18759252f, 0.93417233f},  {0.79465440f, 0.18759243f, 0.57735030f},
    {-0.79465450f, -0.18759249f, 0.57735030f}, {-0.30353105f, 0.18759243f, 0.93417240f},  {-0.79465440f, -0.18759240f, -0.57735030f},
    {-0.98224690f, 0.18759254f, 0.0f},         {0.30353096f, -0.18759250f, -0.93417233f}, {-0.30353084f, 0.18759246f, -0.93417240f},
    {0.18759249f, -0.79465440f, 0.57735026f},  {-0.49112338f, -0.79465450f, 0.35682213f}, {-0.49112338f, -0.79465455f, -0.35682213f},
    {0.18759243f, -0.79465440f, -0.57735026f}, {0.60706200f, -0.79465440f, 0.0f}};
// get_spatial_direction_2d (:402-412): cos / sin of float(i) * (2 pi / 6), the float32 values as literals
__device__ const float RED_DIR[RED_DIRS][2] = {{0x1p+0f, 0x0p+0f},
                                               {0x1.fffffep-2f, 0x1.bb67bp-1f},
                                               {-0x1.000002p-1f, 0x1.bb67aep-1f},
                                               {-0x1p+0f, -0x1.777a5cp-24f},
                                               {-0x1.fffffap-2f, -0x1.bb67bp-1f},
                                               {0x1.fffffap-2f, -0x1.bb67bp-1f}};

NT_DI vec3 red_face(int b) { return vec3(RED_FACE[b][0], RED_FACE[b][1], RED_FACE[b][2]); }
NT_DI uint32_t red_float_flip(float f) {  // contact_reduction.py:99-107
    uint32_t i;
    __builtin_memcpy(&i, &f, 4);
    return i ^ ((uint32_t)(-(int32_t)(i >> 31)) | 0x80000000u);
}
NT_DI unsigned long long red_value_depth(float score, int fp) {  // _make_contact_value_det, fingerprint in the id's place
    return ((unsigned long long)(red_float_flip(score) >> 10) << 22) | ((unsigned long long)fp & RED_FP_MASK);
}
NT_DI unsigned long long red_value_spatial(float score, bool inner, int fp) {  // _make_spatial_contact_value_det
    return ((unsigned long long)(inner ? 1 : 0) << 43) | ((unsigned long long)(red_float_flip(score) >> 11) << 22) |
           ((unsigned long long)fp & RED_FP_MASK);
}
NT_DI int red_get_slot(vec3 n) {  // get_slot, icosahedron: pruned scan over the top cap / belt / bottom cap
    int lo, hi;
    if (n.y > 0.65f) { lo = 0; hi = 5; }
    else if (n.y < -0.65f) { lo = 15; hi = 20; }
    else if (n.y >= 0.0f) { lo = 0; hi = 15; }
    else { lo = 5; hi = 20; }
    int best = lo;
    float best_dot = dot(n, red_face(lo));
    for (int i = lo + 1; i < hi; ++i) {
        const float d = dot(n, red_face(i));
        if (d > best_dot) { best_dot = d; best = i; }
    }
    return best;
}
NT_DI void red_face_frame(int b, vec3& u, vec3& v) {  // project_point_to_plane's basis
    const vec3 fn = red_face(b);
    const vec3 ref = fabsf(fn.y) < 0.9f ? vec3(0.0f, 1.0f, 0.0f) : vec3(1.0f, 0.0f, 0.0f);
    u = normalize(ref - dot(ref, fn) * fn);
    v = cross(fn, u);
}
NT_DI int red_voxel_index(vec3 p, const float* lo, const float* hi, const int* res) {  // compute_voxel_index :432-466
    int vi[3];
    for (int k = 0; k < 3; ++k) {
        const float size = hi[k] - lo[k];
        float rel = 0.0f;
        if (size > 1e-6f) rel = (vget(p, k) - lo[k]) / size;
        int q = (int)(rel * (float)res[k]);
        vi[k] = q < 0 ? 0 : (q > res[k] - 1 ? res[k] - 1 : q);
    }
    return vi[0] + vi[1] * res[0] + vi[2] * res[0] * res[1];
}
NT_DI void red_encode_oct(vec3 n, float& ex, float& ey) {  // :631-658
    const float l1 = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    if (l1 < 1.0e-20f) { ex = 0.0f; ey = 0.0f; return; }
    const float inv = 1.0f / l1;
    float ox = n.x * inv, oy = n.y * inv;
    const float oz = n.z * inv;
    if (oz < 0.0f) {
        const float sx = ox < 0.0f ? -1.0f : 1.0f, sy = oy < 0.0f ? -1.0f : 1.0f;
        const float nx = (1.0f - fabsf(oy)) * sx, ny = (1.0f - fabsf(ox)) * sy;
        ox = nx; oy = ny;
    }
    ex = ox; ey = oy;
}
NT_DI vec3 red_decode_oct(float ex, float ey) {  // :661-683
    const float nz = 1.0f - fabsf(ex) - fabsf(ey);
    float nx = ex, ny = ey;
    if (nz < 0.0f) {
        const float sx = nx < 0.0f ? -1.0f : 1.0f, sy = ny < 0.0f ? -1.0f : 1.0f;
        const float tx = (1.0f - fabsf(ny)) * sx, ty = (1.0f - fabsf(nx)) * sy;
        nx = tx; ny = ty;
    }
    return normalize(vec3(nx, ny, nz));
}
NT_DI bool red_near_ulps(float a, float b) {  // _floats_are_near_ulps :141-150
    if (fabsf(a - b) > 1.0e-8f) return false;
    const uint32_t x = red_float_flip(a), y = red_float_flip(b);
    return (x > y ? x - y : y - x) <= 16u;
}
// One contact offered to the pair's table (export_and_reduce_contact_centered_two_spatial_depths without the races).
NT_DI void red_offer(unsigned long long* tbl, vec3 normal, vec3 centered, float depth, float inner_depth, float outer_depth,
                     vec3 local, const float* lo, const float* hi, const int* res, int fp) {
    if (!(depth < outer_depth)) return;
    const bool use_inner = depth < inner_depth;
    const int b = red_get_slot(normal);
    vec3 u, v;
    red_face_frame(b, u, v);
    const float px = dot(centered, u), py = dot(centered, v);
    for (int d = 0; d < RED_DIRS; ++d) {
        const float score = px * RED_DIR[d][0] + py * RED_DIR[d][1];
        atomicMax(&tbl[b * RED_VALUES + d], red_value_spatial(score, use_inner, fp));
    }
    if (use_inner) {
        const unsigned long long dv = red_value_depth(-depth, fp);
        atomicMax(&tbl[b * RED_VALUES + RED_DIRS], dv);
        int vox = red_voxel_index(local, lo, hi, res);
        vox = vox < 0 ? 0 : (vox > RED_VOXELS - 1 ? RED_VOXELS - 1 : vox);
        atomicMax(&tbl[(RED_BINS + vox / RED_VALUES) * RED_VALUES + vox % RED_VALUES], dv);
    }
}
struct RedLds {
    unsigned long long tbl[RED_SLOTS];
    float pos[RED_SLOTS][4];  // what the reference's buffer holds of a winner: position, depth ...
    float oct[RED_SLOTS][2];  // ... and the octahedral code of its normal
    int fp[RED_SLOTS];        // fingerprint of the slot's winner, -1 = empty
    int src[RED_SLOTS];       // list variant: index of the winner in the caller's list
    int keep[RED_SLOTS];      // survives the roundoff-twin pass of its entry
    int first[RED_SLOTS];     // first kept slot holding this fingerprint (exported_flags: a contact leaves once)
    int srcidx[RED_SLOTS];    // list variant: the winner's list index, kept across red_finish (which compacts into src)
    int base, total;
};

// After the winners' records are in LDS: twins, de-duplication, rank by fingerprint.  Leaves L.first[k] (slot k exports a
// contact), L.keep[k] = its rank among the pair's survivors, L.total; every lane of the workgroup must call it (any size).
// The quadratic steps (first occurrence of a fingerprint, rank among the survivors) only walk the OCCUPIED slots: a pair keeps a
// few dozen winners at most, usually a handful, and a pair without any winner skips them altogether.
struct RedLdsRec {  // records of RedLds
    const RedLds& L;
    NT_DI void operator()(int k, float* o) const {
        o[0] = L.pos[k][0]; o[1] = L.pos[k][1]; o[2] = L.pos[k][2]; o[3] = L.pos[k][3];
        o[4] = L.oct[k][0]; o[5] = L.oct[k][1];
    }
};
// `rec(slot, out[6])` hands out the winner's record (position, depth, octahedral normal code): LDS arrays (RedLds) or the survivor
// list in HBM (RedLdsIdx: the staged reduction keeps only an index per slot, which more than doubles the pairs in flight per CU).
template <class LDS, class REC>
NT_DI void red_finish(LDS& L, REC rec) {
    const int t = threadIdx.x, nt_ = blockDim.x;
    if (t == 0) L.total = 0;
    for (int en = t; en < RED_ENTRIES; en += nt_) {  // _roundoff_duplicate_bit_for_slot_pair over the 21 slot pairs of the entry
        int suppressed = 0, filled = 0;
        const int e0 = en * RED_VALUES;
        for (int sl = 0; sl < RED_VALUES; ++sl) filled += L.fp[e0 + sl] >= 0 ? 1 : 0;
        if (filled >= 2) {
            float rc[RED_VALUES][6];  // the entry's records first (one batch of loads), then the comparisons in registers
#pragma unroll
            for (int sl = 0; sl < RED_VALUES; ++sl)
                if (L.fp[e0 + sl] >= 0) rec(e0 + sl, rc[sl]);
#pragma unroll
            for (int sb = 1; sb < RED_VALUES; ++sb)
#pragma unroll
                for (int sa = 0; sa < sb; ++sa) {
                    const int fa = L.fp[e0 + sa], fb = L.fp[e0 + sb];
                    if (fa < 0 || fb < 0 || fa == fb) continue;
                    const float* pa = rc[sa];
                    const float* pb = rc[sb];
                    const bool same = red_near_ulps(pa[0], pb[0]) && red_near_ulps(pa[1], pb[1]) && red_near_ulps(pa[2], pb[2]) &&
                                      red_near_ulps(pa[3], pb[3]) && red_near_ulps(pa[4], pb[4]) && red_near_ulps(pa[5], pb[5]);
                    if (same) suppressed |= fb < fa ? (1 << sa) : (1 << sb);
                }
        }
        for (int sl = 0; sl < RED_VALUES; ++sl) {
            L.keep[e0 + sl] = L.fp[e0 + sl] >= 0 && !((suppressed >> sl) & 1);
            L.first[e0 + sl] = 0;
        }
    }
    __syncthreads();
    // the kept slots, compacted in ascending slot order into L.src (free here: the list variant has consumed it)
    if (t < 64) {  // the first wave: ballot prefix over rounds of 64 slots
        const unsigned long long below = (1ull << t) - 1ull;
        int n = 0;
        for (int k0 = 0; k0 < RED_SLOTS; k0 += 64) {
            const int k = k0 + t;
            const bool kept = k < RED_SLOTS && L.keep[k];
            const unsigned long long m = __ballot(kept);
            if (kept) L.src[n + __popcll(m & below)] = k;
            n += __popcll(m);
        }
        if (t == 0) L.base = n;
    }
    __syncthreads();
    const int n_occ = L.base;
    for (int i = t; i < n_occ; i += nt_) {
        const int k = L.src[i];
        int first = 1;
        for (int j = 0; j < i && first; ++j)
            if (L.fp[L.src[j]] == L.fp[k]) first = 0;
        L.first[k] = first;
    }
    __syncthreads();
    for (int k = t; k < RED_SLOTS; k += nt_) L.keep[k] = -1;
    __syncthreads();
    for (int i = t; i < n_occ; i += nt_) {
        const int k = L.src[i];
        if (!L.first[k]) continue;
        int rank = 0;
        for (int j = 0; j < n_occ; ++j) {
            const int kj = L.src[j];
            rank += (L.first[kj] && L.fp[kj] < L.fp[k]) ? 1 : 0;
        }
        L.keep[k] = rank;
        atomicAdd(&L.total, 1);
    }
    __syncthreads();
}
