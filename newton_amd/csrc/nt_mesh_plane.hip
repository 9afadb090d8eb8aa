// nt_mesh_plane.hip -- MESH vs infinite plane for gfx950: the vertex leg of CollisionPipeline.collide (include/newton_hip_mesh.h).
//
// Reference behaviour (paths under /root/reference/newton/_src/geometry):
//   routing      narrow_phase.py:618-631     (infinite plane, mesh) pairs -> shape_pairs_mesh_plane, stored (mesh, plane)
//   contacts     narrow_phase.py:1866-1990   one lane per vertex: world point, projection through the plane's frame, distance,
//                                            admission distance < gap sum + margin sum, centre = midpoint, normal = -n
//   reduction    contact_reduction_global.py:2059-2096 (write_contact_to_reducer: position, depth, octahedral normal code),
//                :1246-1346 (reduce_contact_in_hashtable, beta = 1e-4), :2098-2290 (export: roundoff twins, every contact once)
//
// MI355X design.  The reference spreads a pair's vertices over several blocks, buffers every admitted contact in global memory,
// registers the buffer in a device-wide hashtable in a second launch and exports in a third.  Here one workgroup owns a pair: its
// 256 lanes stride over the vertices (12 B each, shared by every world of a replicated scene -> L2 hits after the first world),
// an admitted contact goes straight into the pair's reduction table in LDS (245 x ds_max_u64, nt_contact_reduce.hpp), and the
// <= 245 winners recompute their record from the vertex index after the barrier (same instructions, same bits) -- no contact
// buffer, no hashtable, one launch.  The packed value carries the fingerprint (vertex index, < 2^22), which is unique inside a
// pair.  Rows leave as one contiguous block per pair in ascending vertex order, the order `deterministic=True` sorts into.
// HBM-bound integer / float streaming: vertices in, <= a few dozen 44-byte rows out per pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"
#include "../../include/newton_hip_mesh.h"
#include "nt_math.hpp"

using namespace nt;

namespace {

#include "nt_contact_reduce.hpp"

constexpr int GEO_PLANE = 1;
constexpr float RED_BETA = 0.0001f;  // contact_reduction_global.py:89 BETA_THRESHOLD

NT_DI xform ld_xform(const float* p) { return xform(vec3(p[0], p[1], p[2]), quat(p[3], p[4], p[5], p[6])); }

NT_DI int mp_live_pairs(const nt_mesh_plane_args& a) { return a.pair_world_prefix ? a.pair_world_prefix[a.worlds] : a.pair_count; }
NT_DI int mp_pair_slot(const nt_mesh_plane_args& a, int f) {  // flat live index -> position w * pairs_per_world + k
    if (!a.pair_world_prefix) return f;
    int lo = 0, hi = a.worlds;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.pair_world_prefix[mid] <= f) lo = mid;
        else hi = mid;
    }
    return lo * a.pairs_per_world + (f - a.pair_world_prefix[lo]);
}

struct PairCtx {  // what every vertex of the pair shares
    xform X_mesh, X_plane, X_plane_sw;
    vec3 plane_normal, scale;
    float threshold, margin_mesh, margin_plane;
    int mesh, plane, v0, nv;
};

NT_DI void pair_setup(const nt_mesh_plane_args& a, int s0, int s1, PairCtx& c) {
    const bool plane_first = a.shape_type[s0] == GEO_PLANE;
    c.mesh = plane_first ? s1 : s0;
    c.plane = plane_first ? s0 : s1;
    c.X_mesh = ld_xform(a.shape_transform + 7 * (size_t)c.mesh);
    c.X_plane = ld_xform(a.shape_transform + 7 * (size_t)c.plane);
    c.X_plane_sw = xform_inverse(c.X_plane);
    c.plane_normal = xform_vector(c.X_plane, vec3(0.0f, 0.0f, 1.0f));
    const float* dm = a.shape_data + 4 * (size_t)c.mesh;
    c.scale = vec3(dm[0], dm[1], dm[2]);
    c.margin_mesh = dm[3];
    c.margin_plane = a.shape_data[4 * (size_t)c.plane + 3];
    const float gap_sum = a.shape_gap[c.mesh] + a.shape_gap[c.plane];
    c.threshold = gap_sum + (c.margin_mesh + c.margin_plane);
    c.v0 = a.shape_vertex_range[2 * (size_t)c.mesh];
    c.nv = a.shape_vertex_range[2 * (size_t)c.mesh + 1];
}

// the contact of vertex vi, if it is within margin + gap of the plane
NT_DI bool vertex_contact(const nt_mesh_plane_args& a, const PairCtx& c, int vi, vec3& centre, float& distance) {
    const float* p = a.vertices + 3 * (size_t)(c.v0 + vi);
    const vec3 local(p[0] * c.scale.x, p[1] * c.scale.y, p[2] * c.scale.z);  // wp.cw_mul
    const vec3 world = xform_point(c.X_mesh, local);
    const vec3 in_plane = xform_point(c.X_plane_sw, world);
    const vec3 on_plane = xform_point(c.X_plane, vec3(in_plane.x, in_plane.y, 0.0f));
    distance = dot(world - on_plane, c.plane_normal);
    if (!(distance < c.threshold)) return false;
    centre = (world + on_plane) * 0.5f;
    return true;
}

// reduce_contact_in_hashtable for one buffered contact (position, octahedral-coded normal, depth) of the pair
NT_DI void red_offer_buffered(unsigned long long* tbl, vec3 normal_decoded, vec3 position, float depth, const xform& X_a_inv,
                              const float* lo, const float* hi, const int* res, int fp) {
    const int b = red_get_slot(normal_decoded);
    vec3 u, v;
    red_face_frame(b, u, v);
    const float px = dot(position, u), py = dot(position, v);
    const vec3 diag(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    if (depth < RED_BETA * length(diag)) {
        for (int d = 0; d < RED_DIRS; ++d) {
            const float score = px * RED_DIR[d][0] + py * RED_DIR[d][1];
            atomicMax(&tbl[b * RED_VALUES + d], red_value_depth(score, fp));
        }
    }
    const unsigned long long dv = red_value_depth(-depth, fp);
    atomicMax(&tbl[b * RED_VALUES + RED_DIRS], dv);
    int vox = red_voxel_index(xform_point(X_a_inv, position), lo, hi, res);
    vox = vox < 0 ? 0 : (vox > RED_VOXELS - 1 ? RED_VOXELS - 1 : vox);
    atomicMax(&tbl[(RED_BINS + vox / RED_VALUES) * RED_VALUES + vox % RED_VALUES], dv);
}

NT_DI void write_row(const nt_mesh_plane_args& a, const PairCtx& c, int slot, int pair_idx, int vi, vec3 centre, vec3 normal, float dist) {
    a.out_pair[slot] = pair_idx;
    a.out_key[slot] = vi;
    float* o = a.out_data + 9 * (size_t)slot;
    o[0] = centre.x; o[1] = centre.y; o[2] = centre.z;
    o[3] = normal.x; o[4] = normal.y; o[5] = normal.z;
    o[6] = dist;
    o[7] = c.margin_mesh;
    o[8] = c.margin_plane;
}

__global__ void __launch_bounds__(256) mesh_plane_pairs_kernel(nt_mesh_plane_args a) {
    __shared__ RedLds L;
    __shared__ int wave_hits[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int live = mp_live_pairs(a);
    for (int f = blockIdx.x; f < live; f += gridDim.x) {
        const int pair_idx = mp_pair_slot(a, f);
        if (a.pair_kind && a.pair_kind[pair_idx] != NT_PAIR_KIND_MESH_PLANE) continue;  // another leg's pair (uniform)
        const int s0 = a.pairs[2 * (size_t)pair_idx], s1 = a.pairs[2 * (size_t)pair_idx + 1];
        PairCtx c;
        pair_setup(a, s0, s1, c);
        __syncthreads();  // every lane has read the pair before it is rewritten as (mesh, plane)
        if (t == 0) { a.pairs[2 * (size_t)pair_idx] = c.mesh; a.pairs[2 * (size_t)pair_idx + 1] = c.plane; }
        const vec3 normal = -c.plane_normal;
        if (!a.reduce) {
            // every admitted vertex is a row, ascending vertex order: rounds of 256 vertices, ballot-compacted
            if (t == 0) L.total = 0;
            __syncthreads();
            int counted = 0;
            for (int pass = 0; pass < 2; ++pass) {  // pass 0 counts, pass 1 writes behind the pair's base
                int run = 0;
                for (int v0 = 0; v0 < c.nv; v0 += 256) {
                    const int vi = v0 + t;
                    vec3 centre;
                    float dist = 0.0f;
                    const bool hit = vi < c.nv && vertex_contact(a, c, vi, centre, dist);
                    const unsigned long long m = __ballot(hit);
                    if (lane == 0) wave_hits[wave] = __popcll(m);
                    __syncthreads();
                    int off = run;
                    for (int k = 0; k < wave; ++k) off += wave_hits[k];
                    if (pass == 1 && hit) {
                        const int slot = L.base + off + __popcll(m & ((1ull << lane) - 1ull));
                        if (slot < a.capacity) write_row(a, c, slot, pair_idx, vi, centre, normal, dist);
                    }
                    run += wave_hits[0] + wave_hits[1] + wave_hits[2] + wave_hits[3];
                    __syncthreads();
                }
                if (pass == 0) {
                    counted = run;
                    if (t == 0) {
                        L.base = counted > 0 ? atomicAdd(a.out_count, counted) : 0;
                        const int room = a.capacity - L.base;
                        a.out_blk[2 * (size_t)pair_idx] = L.base;
                        a.out_blk[2 * (size_t)pair_idx + 1] = counted < room ? counted : (room > 0 ? room : 0);
                    }
                    __syncthreads();
                    if (counted == 0) break;
                }
            }
            continue;
        }
        for (int k = t; k < RED_SLOTS; k += blockDim.x) { L.tbl[k] = 0ull; L.fp[k] = -1; L.keep[k] = 0; }
        __syncthreads();
        // what the buffer would hold of the normal: its octahedral code, decoded again (every contact of the pair shares it)
        float ox, oy;
        red_encode_oct(normal, ox, oy);
        const vec3 normal_buffered = red_decode_oct(ox, oy);
        const xform X_mesh_inv = xform_inverse(c.X_mesh);
        const float* lo = a.shape_aabb_lower + 3 * (size_t)c.mesh;
        const float* hi = a.shape_aabb_upper + 3 * (size_t)c.mesh;
        const int* res = a.shape_voxel_res + 3 * (size_t)c.mesh;
        for (int vi = t; vi < c.nv; vi += blockDim.x) {
            vec3 centre;
            float dist;
            if (vertex_contact(a, c, vi, centre, dist)) red_offer_buffered(L.tbl, normal_buffered, centre, dist, X_mesh_inv, lo, hi, res, vi);
        }
        __syncthreads();
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {  // the winner of slot k: its record recomputed from the vertex index
            if (L.tbl[k] == 0ull) continue;
            const int vi = (int)(L.tbl[k] & RED_FP_MASK);
            vec3 centre;
            float dist;
            vertex_contact(a, c, vi, centre, dist);
            L.pos[k][0] = centre.x; L.pos[k][1] = centre.y; L.pos[k][2] = centre.z; L.pos[k][3] = dist;
            L.oct[k][0] = ox; L.oct[k][1] = oy;
            L.fp[k] = vi;
        }
        __syncthreads();
        red_finish(L, RedLdsRec{L});
        if (t == 0) {
            L.base = L.total > 0 ? atomicAdd(a.out_count, L.total) : 0;
            const int room = a.capacity - L.base;
            a.out_blk[2 * (size_t)pair_idx] = L.base;
            a.out_blk[2 * (size_t)pair_idx + 1] = L.total < room ? L.total : (room > 0 ? room : 0);
        }
        __syncthreads();
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {
            const int slot = L.base + L.keep[k];
            if (L.keep[k] < 0 || slot >= a.capacity) continue;
            write_row(a, c, slot, pair_idx, L.fp[k], vec3(L.pos[k][0], L.pos[k][1], L.pos[k][2]), normal_buffered, L.pos[k][3]);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" nt_status nt_mesh_plane_pairs(const nt_mesh_plane_args* a, void* stream) {
    if (!a || !a->pairs || !a->shape_type || !a->shape_transform || !a->shape_data || !a->shape_gap || !a->shape_vertex_range ||
        !a->vertices || !a->out_count || !a->out_pair || !a->out_key || !a->out_data || !a->out_blk || a->capacity < 0)
        return NT_ERR_INVALID_ARG;
    if (a->reduce && (!a->shape_aabb_lower || !a->shape_aabb_upper || !a->shape_voxel_res)) return NT_ERR_INVALID_ARG;
    if (a->pair_world_prefix ? (a->worlds <= 0 || a->pairs_per_world <= 0) : a->pair_count < 0) return NT_ERR_INVALID_ARG;
    long long blocks = a->pair_world_prefix ? (long long)a->worlds * a->pairs_per_world : (long long)a->pair_count;
    if (blocks == 0) return NT_OK;
#ifdef NT_EMULATED_GRID
    const long long grid_cap = NT_EMULATED_GRID;
#else
    const long long grid_cap = 8192;
#endif
    if (blocks > grid_cap) blocks = grid_cap;
    hipLaunchKernelGGL(mesh_plane_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}
