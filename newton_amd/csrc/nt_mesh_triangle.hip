// nt_mesh_triangle.hip -- MESH (no SDF route) and HEIGHTFIELD vs convex primitive / convex hull for gfx950: the triangle leg of CollisionPipeline.collide
// (include/newton_hip_mesh.h).
//
// Reference behaviour (paths under /root/reference/newton/_src/geometry):
//   routing      narrow_phase.py:633-638        a MESH against a non-mesh shape that took none of the earlier routes -> shape_pairs_mesh
//   midphase     narrow_phase.py:1455-1568, collision_core.py:996-1180   the convex shape's support-function AABB in the mesh frame
//                (compute_tight_aabb_from_support :452-548), unscaled (aabb_to_unscaled :924-956), widened by (margin + gap) / |scale|,
//                mesh query, front-face test against the convex shape's origin -> (mesh, convex, triangle) triples
//   contacts     contact_reduction_global.py:2299-2403 (reduce_contacts=True) / narrow_phase.py:1571-1665: the triangle in world space
//                (collision_core.py:1218-1276; mirror parity swaps two vertices), back-face culling, GJK / MPR + manifold
//                (compute_gjk_mpr_contacts with the TRIANGLE support map, support_function.py:174-191, and the triangle's Minkowski
//                seed :467-538), fingerprint = (((triangle << 1) | 1) << 3) | manifold index
//   reduction    contact_reduction_global.py:2059-2096 (write_contact_to_reducer: every contact is buffered, no gap test),
//                :1246-1346 (reduce_contact_in_hashtable, beta = 1e-4), :2098-2290 (export: roundoff twins, every contact once)
//
// MI355X design.  The reference runs four launches over device-wide buffers (midphase -> triangle list, contacts -> contact buffer,
// hashtable registration, export).  Here one workgroup -- ONE WAVE -- owns a pair.  Its lanes scan the mesh's triangles 64 at a time -- 36 B
// of indices + vertices per triangle, shared by every world of a replicated scene (L2 hits after the first world); Warp's BVH is
// replaced by that scan: a tree walk is a dependent-load chain per lane, the scan is a coalesced stream, and the set it returns is
// the set the BVH query returns (every triangle whose bounds touch the query box); with the optional bounds of every 64 consecutive
// triangles (a one-level hierarchy over the index order, built once on the host) the scan first drops the blocks that miss the
// box -- same set, 32 rounds -> 2 on an 8 192-triangle terrain.  Survivors are ballot-compacted into an LDS list
// in ascending triangle order; once a wave's worth is waiting (or the scan ends) every lane takes one triangle through MPR / GJK and the
// manifold (nt_convex.hpp, the code of the convex tiles) and offers its contacts to the pair's 245-slot reduction table in LDS
// (ds_max_u64, nt_contact_reduce.hpp).  The <= 245 winners take their record from the LDS copy of the pair's (usually only) batch,
// or recompute it from (triangle, manifold index) when the pair had several -- same instructions, same bits -- so there is no
// contact buffer in HBM, no hashtable and no atomics on shared counters except one row allocation per pair.  Rows leave as one contiguous block per pair in ascending fingerprint order (what deterministic=True sorts
// into).  reduce = 0: every generated contact is a row (counted in a first pass, written in a second).
// Bound: the dependent MPR / GJK iterations of the survivors (one lane per triangle), not the scan.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/newton_hip.h"
#include "../../include/newton_hip_mesh.h"
// This unit compiles its OWN variant of nt_convex.hpp (triangle support, unfiltered contacts).  On the device every translation unit
// is its own code object; the host build of the emulator (tests/emu) links all units into one library, where the plain __device__
// functions of the header would collide with the stepping units' copies -- so this unit's math + convex code lives in namespace nt_tri.
#define NT_CONVEX_WITH_TRIANGLES 1
#define nt nt_tri
#include "nt_math.hpp"
#include "nt_convex.hpp"

using namespace nt;

namespace {

#include "nt_contact_reduce.hpp"

constexpr float RED_BETA = 0.0001f;  // contact_reduction_global.py:89 BETA_THRESHOLD
// Lanes per workgroup (= per pair).  A pair of a pile or of a body on a terrain has a few dozen candidate triangles: with 256 lanes one
// wave of four runs MPR while three idle, and 251 VGPRs leave room for two such workgroups per CU.  One WAVE per pair keeps eight
// pairs per CU in flight (26 KB of LDS each).  256 lanes remain as a launch shape for measurements (NT_MESH_TRIANGLE_THREADS=256).
constexpr int MT_DEFAULT_THREADS = 64;

NT_DI xform ld_xform(const float* p) { return xform(vec3(p[0], p[1], p[2]), quat(p[3], p[4], p[5], p[6])); }
NT_DI vec3 ld_vec3(const float* p) { return vec3(p[0], p[1], p[2]); }
NT_DI int imax(int a, int b) { return a > b ? a : b; }

NT_DI int mt_live_pairs(const nt_mesh_triangle_args& a) { return a.pair_world_prefix ? a.pair_world_prefix[a.worlds] : a.pair_count; }
NT_DI int mt_pair_slot(const nt_mesh_triangle_args& a, int f) {  // flat live index -> position w * pairs_per_world + k
    if (!a.pair_world_prefix) return f;
    int lo = 0, hi = a.worlds;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.pair_world_prefix[mid] <= f) lo = mid;
        else hi = mid;
    }
    return lo * a.pairs_per_world + (f - a.pair_world_prefix[lo]);
}

struct PairSetup {  // what every triangle of the pair shares
    xform X_mesh, X_convex;
    vec3 mesh_scale, q_lo, q_hi, center_in_bvh;
    Geom gb;  // the convex shape as extract_shape_data hands it over
    float margin_mesh, margin_convex, gap_sum, radius_b;
    int mesh, convex, v0, t0, nt_;
    bool mirrored;
    // heightfield pairs (`mesh` is the heightfield shape): the cell range of the midphase, the field's grid
    bool hfield;
    int row_min, col_min, n_cols, hf_cols, hf_ncol, hf_offset;  // cells [row_min ..][col_min .. col_min + n_cols), hf_cols = ncol - 1
    float hx, hy, dx, dy, min_z, z_range;
};

NT_DI void pair_setup(const nt_mesh_triangle_args& a, int s0, int s1, PairSetup& c) {
    c.hfield = a.shape_heightfield_index != nullptr && (a.shape_type[s0] == GEO_HFIELD || a.shape_type[s1] == GEO_HFIELD);
    const bool mesh_first = c.hfield ? a.shape_type[s0] == GEO_HFIELD : a.shape_type[s0] == GEO_MESH;
    c.mesh = mesh_first ? s0 : s1;
    c.convex = mesh_first ? s1 : s0;
    c.X_mesh = ld_xform(a.shape_transform + 7 * (size_t)c.mesh);
    c.X_convex = ld_xform(a.shape_transform + 7 * (size_t)c.convex);
    const float* dm = a.shape_data + 4 * (size_t)c.mesh;
    const float* dc = a.shape_data + 4 * (size_t)c.convex;
    c.mesh_scale = vec3(dm[0], dm[1], dm[2]);
    c.margin_mesh = dm[3];
    c.margin_convex = dc[3];
    c.gap_sum = a.shape_gap[c.convex] + a.shape_gap[c.mesh];
    const float contact_threshold = c.gap_sum + c.margin_convex + c.margin_mesh;
    c.gb = Geom();
    c.gb.type = a.shape_type[c.convex];
    c.gb.scale = vec3(dc[0], dc[1], dc[2]);
    c.gb.center = vec3(0.0f);
    c.gb.aux = vec3(0.0f);
    c.radius_b = (c.gb.type == GEO_SPHERE || c.gb.type == GEO_CAPSULE) ? dc[0] : 0.0f;  // compute_effective_radius of the export
    if (c.gb.type == GEO_CONVEX_MESH) {  // extract_shape_data: the hull's vertex table; _shape_center (support_function.py:448-464):
        c.gb.points = a.hull_points + 3 * (size_t)a.shape_hull_range[2 * (size_t)c.convex];  // the centre of its scaled bounds seeds MPR / GJK
        c.gb.count = a.shape_hull_range[2 * (size_t)c.convex + 1];
        const vec3 first = cw_mul(ld_vec3(c.gb.points), c.gb.scale);
        vec3 lower = first, upper = first;
        for (int i = 1; i < c.gb.count; ++i) {
            const vec3 point = cw_mul(ld_vec3(c.gb.points + 3 * (size_t)i), c.gb.scale);
            lower = vmin(lower, point);
            upper = vmax(upper, point);
        }
        c.gb.center = 0.5f * (lower + upper);
    }
    c.mirrored = false;
    c.v0 = c.t0 = 0;
    if (c.hfield) {
        // heightfield_vs_convex_midphase (utils/heightfield.py:366-462): the partner's LOCAL AABB (Model.shape_collision_aabb_*) as an
        // oriented box in the heightfield frame, its axis-aligned hull widened by margin + gap, mapped to a cell range
        const nt_heightfield hd = a.heightfields[a.shape_heightfield_index[c.mesh]];
        const xform X_other_in_hfield = xform_inverse(c.X_mesh) * c.X_convex;
        const vec3 other_pos = X_other_in_hfield.p;
        const quat other_rot = X_other_in_hfield.q;
        const vec3 local_lo = ld_vec3(a.shape_aabb_lower + 3 * (size_t)c.convex), local_hi = ld_vec3(a.shape_aabb_upper + 3 * (size_t)c.convex);
        const vec3 local_center = 0.5f * (local_lo + local_hi), local_half = 0.5f * (local_hi - local_lo);
        const vec3 center_in_hfield = quat_rotate(other_rot, local_center) + other_pos;
        const vec3 r0 = quat_rotate(other_rot, vec3(1.0f, 0.0f, 0.0f)), r1 = quat_rotate(other_rot, vec3(0.0f, 1.0f, 0.0f)),
                   r2 = quat_rotate(other_rot, vec3(0.0f, 0.0f, 1.0f));
        const vec3 half_in_hfield(fabsf(r0.x) * local_half.x + fabsf(r1.x) * local_half.y + fabsf(r2.x) * local_half.z,
                                  fabsf(r0.y) * local_half.x + fabsf(r1.y) * local_half.y + fabsf(r2.y) * local_half.z,
                                  fabsf(r0.z) * local_half.x + fabsf(r1.z) * local_half.y + fabsf(r2.z) * local_half.z);
        const float margin_sum = c.margin_mesh + c.margin_convex;
        const float threshold = (a.shape_gap[c.mesh] + a.shape_gap[c.convex]) + margin_sum;
        const vec3 threshold_vec(threshold, threshold, threshold);
        const vec3 q_lo = center_in_hfield - half_in_hfield - threshold_vec, q_hi = center_in_hfield + half_in_hfield + threshold_vec;
        c.hx = hd.hx; c.hy = hd.hy;
        c.dx = 2.0f * hd.hx / (float)(hd.ncol - 1);
        c.dy = 2.0f * hd.hy / (float)(hd.nrow - 1);
        const int col_min = imax((int)floorf((q_lo.x + hd.hx) / c.dx), 0), col_max = imin((int)floorf((q_hi.x + hd.hx) / c.dx), hd.ncol - 2);
        const int row_min = imax((int)floorf((q_lo.y + hd.hy) / c.dy), 0), row_max = imin((int)floorf((q_hi.y + hd.hy) / c.dy), hd.nrow - 2);
        c.row_min = row_min; c.col_min = col_min;
        c.n_cols = col_max >= col_min ? col_max - col_min + 1 : 0;
        const int n_rows = row_max >= row_min ? row_max - row_min + 1 : 0;
        c.nt_ = 2 * n_rows * c.n_cols;  // candidates: both triangles of every cell of the range (no bounds test)
        c.hf_cols = hd.ncol - 1; c.hf_ncol = hd.ncol; c.hf_offset = hd.data_offset;
        c.min_z = hd.min_z; c.z_range = hd.max_z - hd.min_z;
        c.gap_sum = a.shape_gap[c.mesh] + a.shape_gap[c.convex];
        c.mesh_scale = vec3(1.0f, 1.0f, 1.0f);
        return;
    }
    // _compute_mesh_vs_convex_query_aabb (collision_core.py:996-1040)
    const xform X_mesh_shape = xform_inverse(c.X_mesh) * c.X_convex;
    const vec3 pos_in_mesh = X_mesh_shape.p;
    const mat33 rt = transpose(quat_to_matrix(X_mesh_shape.q));
    const vec3 local_x(rt.m00, rt.m10, rt.m20), local_y(rt.m01, rt.m11, rt.m21), local_z(rt.m02, rt.m12, rt.m22);
    float max_x, max_y, max_z, min_x, min_y, min_z;
    if (c.gb.type == GEO_CONVEX_MESH) {  // collision_core.py:491-523: one pass over the hull's vertices, axes pre-scaled
        const vec3 scaled_x = cw_mul(local_x, c.gb.scale), scaled_y = cw_mul(local_y, c.gb.scale), scaled_z = cw_mul(local_z, c.gb.scale);
        min_x = min_y = min_z = 1.0e10f;
        max_x = max_y = max_z = -1.0e10f;
        for (int i = 0; i < c.gb.count; ++i) {
            const vec3 p = ld_vec3(c.gb.points + 3 * (size_t)i);
            const float vx = dot(p, scaled_x), vy = dot(p, scaled_y), vz = dot(p, scaled_z);
            min_x = fminw(min_x, vx); max_x = fmaxw(max_x, vx);
            min_y = fminw(min_y, vy); max_y = fmaxw(max_y, vy);
            min_z = fminw(min_z, vz); max_z = fmaxw(max_z, vz);
        }
    } else {
        max_x = dot(local_x, support_map(c.gb, local_x));
        max_y = dot(local_y, support_map(c.gb, local_y));
        max_z = dot(local_z, support_map(c.gb, local_z));
        min_x = dot(local_x, support_map(c.gb, -local_x));
        min_y = dot(local_y, support_map(c.gb, -local_y));
        min_z = dot(local_z, support_map(c.gb, -local_z));
    }
    const vec3 aabb_lower = vec3(min_x, min_y, min_z) + pos_in_mesh, aabb_upper = vec3(max_x, max_y, max_z) + pos_in_mesh;
    const float eps = 1.0e-12f;
    auto guarded = [&](float s) { return fabsf(s) > eps ? s : (s >= 0.0f ? eps : -eps); };
    const vec3 inv_scale(1.0f / guarded(c.mesh_scale.x), 1.0f / guarded(c.mesh_scale.y), 1.0f / guarded(c.mesh_scale.z));
    const vec3 l0 = cw_mul(aabb_lower, inv_scale), l1 = cw_mul(aabb_upper, inv_scale);
    const vec3 margin_vec(contact_threshold / fmaxw(fabsf(c.mesh_scale.x), 1.0e-12f), contact_threshold / fmaxw(fabsf(c.mesh_scale.y), 1.0e-12f),
                          contact_threshold / fmaxw(fabsf(c.mesh_scale.z), 1.0e-12f));
    c.q_lo = vmin(l0, l1) - margin_vec;
    c.q_hi = vmax(l0, l1) + margin_vec;
    c.center_in_bvh = cw_mul(pos_in_mesh, inv_scale);
    c.v0 = a.shape_vertex_range[2 * (size_t)c.mesh];
    c.t0 = a.shape_triangle_range[2 * (size_t)c.mesh];
    c.nt_ = a.shape_triangle_range[2 * (size_t)c.mesh + 1];
    c.mirrored = c.mesh_scale.x * c.mesh_scale.y * c.mesh_scale.z < 0.0f;
}
// candidate j of a heightfield pair -> packed triangle index (row * (ncol - 1) + col) * 2 + tri_sub, ascending in j
NT_DI int hfield_candidate(const PairSetup& c, int j) {
    const int cell = j >> 1, r = c.row_min + cell / c.n_cols, col = c.col_min + cell % c.n_cols;
    return (r * c.hf_cols + col) * 2 + (j & 1);
}

// the midphase's verdict on triangle ti: its bounds touch the query box and it faces the convex shape's origin
NT_DI bool triangle_candidate(const nt_mesh_triangle_args& a, const PairSetup& c, int ti) {
    const int* idx = a.indices + 3 * (size_t)(c.t0 + ti);
    const vec3 v0 = ld_vec3(a.vertices + 3 * (size_t)(c.v0 + idx[0])), v1 = ld_vec3(a.vertices + 3 * (size_t)(c.v0 + idx[1])),
               v2 = ld_vec3(a.vertices + 3 * (size_t)(c.v0 + idx[2]));
    const vec3 tlo = vmin(v0, vmin(v1, v2)), thi = vmax(v0, vmax(v1, v2));
    if (tlo.x > c.q_hi.x || tlo.y > c.q_hi.y || tlo.z > c.q_hi.z || thi.x < c.q_lo.x || thi.y < c.q_lo.y || thi.z < c.q_lo.z) return false;
    const vec3 face_normal = cross(v1 - v0, v2 - v0);  // _mesh_triangle_is_front_facing_local: unscaled frame, stored winding
    return !(dot(face_normal, c.center_in_bvh - v0) < 0.0f);
}

// the contacts of triangle ti (emission order, unfiltered); false: culled as a back face in world space
NT_DI bool triangle_contacts(const nt_mesh_triangle_args& a, const PairSetup& c, int ti, PolyRef poly, ConvexContacts& out) {
    if (c.hfield) {
        // get_triangle_shape_from_heightfield (utils/heightfield.py:280-363): cell (row, col), tri_sub 0 = (p00, p10, p11), 1 = (p00, p11, p01);
        // a TRIANGLE_PRISM with its edges in the heightfield frame, MPR / GJK run in that frame (rotation = the heightfield's)
        const int cell_idx = ti / 2, tri_sub = ti - cell_idx * 2, row = cell_idx / c.hf_cols, col = cell_idx - row * c.hf_cols;
        const float x0 = -c.hx + (float)col * c.dx, x1 = x0 + c.dx, y0 = -c.hy + (float)row * c.dy, y1 = y0 + c.dy;
        const float* e = a.elevations + c.hf_offset;
        const float h00 = e[row * c.hf_ncol + col], h10 = e[row * c.hf_ncol + (col + 1)], h01 = e[(row + 1) * c.hf_ncol + col],
                    h11 = e[(row + 1) * c.hf_ncol + (col + 1)];
        const float z00 = c.min_z + h00 * c.z_range, z10 = c.min_z + h10 * c.z_range, z01 = c.min_z + h01 * c.z_range, z11 = c.min_z + h11 * c.z_range;
        const vec3 p00(x0, y0, z00), p10(x1, y0, z10), p01(x0, y1, z01), p11(x1, y1, z11);
        const vec3 v0_local = p00, v1_local = tri_sub == 0 ? p10 : p11, v2_local = tri_sub == 0 ? p11 : p01;
        triangle_pair(GEO_TRIANGLE_PRISM, v1_local - v0_local, v2_local - v0_local, xform_point(c.X_mesh, v0_local), c.X_mesh.q, c.gb, c.X_convex,
                      c.margin_mesh, c.margin_convex, c.gap_sum, poly, out);
        return true;
    }
    const int* idx = a.indices + 3 * (size_t)(c.t0 + ti);
    const int i0 = idx[0], i1 = c.mirrored ? idx[2] : idx[1], i2 = c.mirrored ? idx[1] : idx[2];
    // get_triangle_shape_from_mesh (collision_core.py:1218-1276)
    const vec3 v0_world = xform_point(c.X_mesh, cw_mul(ld_vec3(a.vertices + 3 * (size_t)(c.v0 + i0)), c.mesh_scale));
    const vec3 v1_world = xform_point(c.X_mesh, cw_mul(ld_vec3(a.vertices + 3 * (size_t)(c.v0 + i1)), c.mesh_scale));
    const vec3 v2_world = xform_point(c.X_mesh, cw_mul(ld_vec3(a.vertices + 3 * (size_t)(c.v0 + i2)), c.mesh_scale));
    const vec3 ab = v1_world - v0_world, ac = v2_world - v0_world;
    out.count = 0;
    // back-face culling (contact_reduction_global.py:2368-2375)
    if (dot(cross(ab, ac), c.X_convex.p - v0_world) < 0.0f) return false;
    triangle_pair(GEO_TRIANGLE, ab, ac, v0_world, quat(0.0f, 0.0f, 0.0f, 1.0f), c.gb, c.X_convex, c.margin_mesh, c.margin_convex, c.gap_sum, poly, out);
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

NT_DI void write_row(const nt_mesh_triangle_args& a, const PairSetup& c, int slot, int pair_idx, int fp, vec3 centre, vec3 normal, float dist) {
    a.out_pair[slot] = pair_idx;
    a.out_key[slot] = fp;
    float* o = a.out_data + 9 * (size_t)slot;
    o[0] = centre.x; o[1] = centre.y; o[2] = centre.z;
    o[3] = normal.x; o[4] = normal.y; o[5] = normal.z;
    o[6] = dist;
    o[7] = c.margin_mesh;
    o[8] = c.margin_convex;
    if (a.out_radius) {
        a.out_radius[2 * (size_t)slot] = 0.0f;
        a.out_radius[2 * (size_t)slot + 1] = c.radius_b;
    }
}

template <int MT_THREADS>
struct MtLds {
    RedLds red;
    int list[2 * MT_THREADS];  // candidate triangles waiting for a batch, ascending
    unsigned short hblk[(1 << 18) / NT_MESH_TRIANGLE_BLOCK];  // blocks of the mesh that touch the query box, ascending
    int n_hblk;
    float poly[20 * MT_THREADS];  // manifold clipper scratch: 10 x vec2 per lane, lane-strided
    float rec[30 * MT_THREADS];   // the first batch's contacts, lane-strided: 5 x (centre, distance, octahedral normal code)
    int batch_tri[MT_THREADS];    // ... and its triangles (ascending)
    int wave_hits[MT_THREADS / 64];
    int waiting;        // candidates in `list`
    int rows;           // reduce = 0: contacts generated so far (pass 0: count; pass 1: rank of the next batch)
};

template <int MT_THREADS>
__global__ void __launch_bounds__(MT_THREADS) mesh_triangle_pairs_kernel(nt_mesh_triangle_args a) {
    constexpr int MT_WAVES = MT_THREADS / 64;
    __shared__ MtLds<MT_THREADS> S;
    RedLds& L = S.red;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const PolyRef poly{S.poly + t, MT_THREADS};
    const int live = mt_live_pairs(a);
    for (int f = blockIdx.x; f < live; f += gridDim.x) {
        const int pair_idx = mt_pair_slot(a, f);
        if (a.pair_kind && a.pair_kind[pair_idx] != NT_PAIR_KIND_MESH_TRIANGLE) continue;  // another leg's pair (uniform)
        const int s0 = a.pairs[2 * (size_t)pair_idx], s1 = a.pairs[2 * (size_t)pair_idx + 1];
        if (!a.hull_points && (a.shape_type[s0] == GEO_CONVEX_MESH || a.shape_type[s1] == GEO_CONVEX_MESH)) continue;  // (uniform) no hull table
        {   // (uniform) a heightfield pair without the heightfield tables / a mesh pair without the mesh tables / two mesh-like shapes
            const int ty0 = a.shape_type[s0], ty1 = a.shape_type[s1];
            const bool hf = ty0 == GEO_HFIELD || ty1 == GEO_HFIELD, ms = ty0 == GEO_MESH || ty1 == GEO_MESH;
            if ((hf && (!a.shape_heightfield_index || ms || ty0 == ty1)) || (!hf && (!ms || !a.indices || ty0 == ty1))) continue;
            if (hf && a.shape_heightfield_index[ty0 == GEO_HFIELD ? s0 : s1] < 0) continue;
        }
        PairSetup c;
        pair_setup(a, s0, s1, c);
        __syncthreads();  // every lane has read the pair before it is rewritten as (mesh, convex)
        if (t == 0) { a.pairs[2 * (size_t)pair_idx] = c.mesh; a.pairs[2 * (size_t)pair_idx + 1] = c.convex; }
        const xform X_mesh_inv = xform_inverse(c.X_mesh);
        const float* lo = a.shape_aabb_lower ? a.shape_aabb_lower + 3 * (size_t)c.mesh : nullptr;
        const float* hi = a.shape_aabb_upper ? a.shape_aabb_upper + 3 * (size_t)c.mesh : nullptr;
        const int* res = a.shape_voxel_res ? a.shape_voxel_res + 3 * (size_t)c.mesh : nullptr;
        if (a.reduce)
            for (int k = t; k < RED_SLOTS; k += blockDim.x) { L.tbl[k] = 0ull; L.fp[k] = -1; L.keep[k] = 0; }
        // ---- block bounds (optional): the blocks of 64 consecutive triangles whose bounds touch the query box, ascending.  A
        // triangle that touches the box lies in a block that does, so the candidate set is the full scan's; spatially coherent index
        // orders (grids, most exported meshes) leave a handful of blocks of a large mesh
        const bool use_blocks = a.block_bounds != nullptr && a.shape_block_start != nullptr && !c.hfield;
        int n_hblk = 0;
        if (use_blocks) {
            const int nblk = (c.nt_ + NT_MESH_TRIANGLE_BLOCK - 1) / NT_MESH_TRIANGLE_BLOCK;
            const float* bb = a.block_bounds + 6 * (size_t)a.shape_block_start[c.mesh];
            if (t == 0) S.n_hblk = 0;
            __syncthreads();
            for (int b0 = 0; b0 < nblk; b0 += MT_THREADS) {
                const int b = b0 + t;
                bool hit = false;
                if (b < nblk) {
                    const float* o = bb + 6 * (size_t)b;
                    hit = !(o[0] > c.q_hi.x || o[1] > c.q_hi.y || o[2] > c.q_hi.z || o[3] < c.q_lo.x || o[4] < c.q_lo.y || o[5] < c.q_lo.z);
                }
                const unsigned long long m = __ballot(hit);
                if (lane == 0) S.wave_hits[wave] = __popcll(m);
                __syncthreads();
                int off = S.n_hblk;
                for (int k = 0; k < wave; ++k) off += S.wave_hits[k];
                if (hit) S.hblk[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)b;
                __syncthreads();
                if (t == 0) {
                    int n = S.n_hblk;
                    for (int k = 0; k < MT_THREADS / 64; ++k) n += S.wave_hits[k];
                    S.n_hblk = n;
                }
                __syncthreads();
            }
            n_hblk = S.n_hblk;
        }
        const int n_rounds = use_blocks ? (n_hblk + MT_WAVES - 1) / MT_WAVES : (c.nt_ + MT_THREADS - 1) / MT_THREADS;
        const int passes = a.reduce ? 1 : 2;  // reduce = 0: pass 0 counts the pair's contacts, pass 1 writes them behind its base
        int batches = 0, batch_n = 0;  // (uniform) batches of this pair so far, triangles of the first one
        for (int pass = 0; pass < passes; ++pass) {
            if (t == 0) { S.waiting = 0; S.rows = 0; }
            __syncthreads();
            for (int r = 0; r < n_rounds || S.waiting > 0; ++r) {  // (uniform: S.waiting is read between barriers)
                // ---- scan: the next triangles (a workgroup's worth of the mesh, or one surviving block per wave) against the query box, survivors
                // appended in ascending order
                if (r < n_rounds) {
                    int ti = r * MT_THREADS + t;
                    if (use_blocks) ti = MT_WAVES * r + wave < n_hblk ? (int)S.hblk[MT_WAVES * r + wave] * NT_MESH_TRIANGLE_BLOCK + lane : c.nt_;
                    bool hit = ti < c.nt_;
                    if (hit && c.hfield) ti = hfield_candidate(c, ti);  // (every cell of the range is a candidate)
                    else hit = hit && triangle_candidate(a, c, ti);
                    const unsigned long long m = __ballot(hit);
                    if (lane == 0) S.wave_hits[wave] = __popcll(m);
                    __syncthreads();
                    int off = S.waiting;
                    for (int k = 0; k < wave; ++k) off += S.wave_hits[k];
                    if (hit) S.list[off + __popcll(m & ((1ull << lane) - 1ull))] = ti;
                    __syncthreads();
                    if (t == 0) {
                        int n = S.waiting;
                        for (int k = 0; k < MT_THREADS / 64; ++k) n += S.wave_hits[k];
                        S.waiting = n;
                    }
                    __syncthreads();
                }
                const int waiting = S.waiting;
                const bool last = r + 1 >= n_rounds;
                if (waiting < MT_THREADS && !(last && waiting > 0)) continue;  // (uniform) keep scanning until a full batch waits
                // ---- batch: one lane per waiting triangle
                const int nb = waiting < MT_THREADS ? waiting : MT_THREADS;
                ConvexContacts cc;
                cc.count = 0;
                int ti = -1;
                if (t < nb) {
                    ti = S.list[t];
                    triangle_contacts(a, c, ti, poly, cc);
                }
                if (a.reduce) {
                    for (int i = 0; i < cc.count; ++i) {
                        float ox, oy;
                        red_encode_oct(cc.normal_of(i), ox, oy);  // (the buffer holds the normal as its octahedral code)
                        red_offer_buffered(L.tbl, red_decode_oct(ox, oy), cc.center(i), cc.distance(i), X_mesh_inv, lo, hi, res, (ti << 4) | 8 | i);
                    }
                    // the FIRST batch's contacts stay in LDS: a pair that needs no second batch (the usual case) hands its winners
                    // their records from here instead of running MPR / GJK for them again
                    if (batches == 0 && t < nb) {
                        S.batch_tri[t] = ti;
                        float* rcd = S.rec + t;
                        for (int i = 0; i < cc.count; ++i) {
                            const vec3 ctr = cc.center(i);
                            float ox, oy;
                            red_encode_oct(cc.normal_of(i), ox, oy);
                            rcd[(6 * i) * MT_THREADS] = ctr.x; rcd[(6 * i + 1) * MT_THREADS] = ctr.y; rcd[(6 * i + 2) * MT_THREADS] = ctr.z;
                            rcd[(6 * i + 3) * MT_THREADS] = cc.distance(i); rcd[(6 * i + 4) * MT_THREADS] = ox; rcd[(6 * i + 5) * MT_THREADS] = oy;
                        }
                    }
                    if (batches == 0) batch_n = nb;
                    batches += 1;
                } else {
                    // exclusive prefix of the lanes' contact counts over the workgroup, in lane (= triangle) order
                    int x = cc.count;
                    for (int d = 1; d < 64; d <<= 1) {
                        const int y = __shfl_up(x, d);
                        if (lane >= d) x += y;
                    }
                    if (lane == 63) S.wave_hits[wave] = x;
                    __syncthreads();
                    int before = S.rows + x - cc.count;
                    for (int k = 0; k < wave; ++k) before += S.wave_hits[k];
                    if (pass == 1)
                        for (int i = 0; i < cc.count; ++i) {
                            const int slot = L.base + before + i;
                            if (slot < a.capacity) write_row(a, c, slot, pair_idx, (ti << 4) | 8 | i, cc.center(i), cc.normal_of(i), cc.distance(i));
                        }
                    __syncthreads();
                    if (t == 0) {
                        int n = S.rows;
                        for (int k = 0; k < MT_THREADS / 64; ++k) n += S.wave_hits[k];
                        S.rows = n;
                    }
                }
                __syncthreads();
                // the triangles still waiting move to the front of the list
                const int left = waiting - nb;
                int moved = -1;
                if (t < left) moved = S.list[nb + t];
                __syncthreads();
                if (t < left) S.list[t] = moved;
                if (t == 0) S.waiting = left;
                __syncthreads();
                if (last && left == 0) break;
            }
            __syncthreads();
            if (!a.reduce && pass == 0) {
                const int counted = S.rows;
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
        if (!a.reduce) {
            __syncthreads();
            continue;
        }
        __syncthreads();
        // ---- the winner of slot k: its record from the first batch's LDS copy (the pair had one batch: found by its triangle in the
        // batch's ascending list), else recomputed from (triangle, manifold index) -- the same instructions, the same bits
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {
            if (L.tbl[k] == 0ull) continue;
            const int fp = (int)(L.tbl[k] & RED_FP_MASK);
            if (batches == 1) {
                const int tri = fp >> 4, i = fp & 7;
                int lo_ = 0, hi_ = batch_n - 1;
                while (lo_ < hi_) {
                    const int mid = (lo_ + hi_) >> 1;
                    if (S.batch_tri[mid] < tri) lo_ = mid + 1;
                    else hi_ = mid;
                }
                const float* rcd = S.rec + lo_;
                L.pos[k][0] = rcd[(6 * i) * MT_THREADS]; L.pos[k][1] = rcd[(6 * i + 1) * MT_THREADS];
                L.pos[k][2] = rcd[(6 * i + 2) * MT_THREADS]; L.pos[k][3] = rcd[(6 * i + 3) * MT_THREADS];
                L.oct[k][0] = rcd[(6 * i + 4) * MT_THREADS]; L.oct[k][1] = rcd[(6 * i + 5) * MT_THREADS];
                L.fp[k] = fp;
                continue;
            }
            ConvexContacts cc;
            triangle_contacts(a, c, fp >> 4, poly, cc);
            const int i = fp & 7;
            const vec3 centre = cc.center(i);
            float ox, oy;
            red_encode_oct(cc.normal_of(i), ox, oy);
            L.pos[k][0] = centre.x; L.pos[k][1] = centre.y; L.pos[k][2] = centre.z; L.pos[k][3] = cc.distance(i);
            L.oct[k][0] = ox; L.oct[k][1] = oy;
            L.fp[k] = fp;
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
            write_row(a, c, slot, pair_idx, L.fp[k], vec3(L.pos[k][0], L.pos[k][1], L.pos[k][2]), red_decode_oct(L.oct[k][0], L.oct[k][1]),
                      L.pos[k][3]);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" nt_status nt_mesh_triangle_pairs(const nt_mesh_triangle_args* a, void* stream) {
    if (!a || !a->pairs || !a->shape_type || !a->shape_transform || !a->shape_data || !a->shape_gap ||
        (!a->shape_heightfield_index && (!a->shape_vertex_range || !a->shape_triangle_range || !a->vertices || !a->indices)) || !a->out_count || !a->out_pair || !a->out_key || !a->out_data ||
        !a->out_blk || a->capacity < 0)
        return NT_ERR_INVALID_ARG;
    if (a->reduce && (!a->shape_aabb_lower || !a->shape_aabb_upper || !a->shape_voxel_res)) return NT_ERR_INVALID_ARG;
    if (a->pair_world_prefix ? (a->worlds <= 0 || a->pairs_per_world <= 0) : a->pair_count < 0) return NT_ERR_INVALID_ARG;
    if ((a->block_bounds == nullptr) != (a->shape_block_start == nullptr)) return NT_ERR_INVALID_ARG;
    if ((a->hull_points == nullptr) != (a->shape_hull_range == nullptr)) return NT_ERR_INVALID_ARG;
    if (a->shape_heightfield_index && (!a->heightfields || !a->elevations || !a->shape_aabb_lower || !a->shape_aabb_upper)) return NT_ERR_INVALID_ARG;
    long long blocks = a->pair_world_prefix ? (long long)a->worlds * a->pairs_per_world : (long long)a->pair_count;
    if (blocks == 0) return NT_OK;
#ifdef NT_EMULATED_GRID
    const long long grid_cap = NT_EMULATED_GRID;
#else
    const long long grid_cap = 32768;
#endif
    if (blocks > grid_cap) blocks = grid_cap;
    static const int threads = [] {  // (read once: a launch shape for measurements, not an API)
        const char* e = getenv("NT_MESH_TRIANGLE_THREADS");
        return e && atoi(e) == 256 ? 256 : MT_DEFAULT_THREADS;
    }();
    if (threads == 256) hipLaunchKernelGGL(mesh_triangle_pairs_kernel<256>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(mesh_triangle_pairs_kernel<64>, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}
