// nt_sdf_pipeline.hip -- the mesh-SDF leg of CollisionPipeline.collide as device stages for gfx950: candidate pairs of the SDF
// shapes per world, deterministic row placement, the contact writer on the env-major state, the per-body block lists the solvers
// sum through, and the penalty forces of SolverSemiImplicit / SolverFeatherstone on those rows.
//
// Reference behaviour (paths under /root/reference/newton/_src):
//   pair routing     geometry/narrow_phase.py:620-655 (both shapes carry a texture SDF and collision edges, not box-box -> the
//                    mesh-mesh SDF kernel), broad phase rules geometry/broad_phase_common.py:20-38,220-268
//   contact rows     sim/collide.py:166-254 (write_contact: world -> body frames, gap admission)
//   penalty forces   solvers/semi_implicit/kernels_contact.py:381-556 (eval_body_contact)
//
// MI355X design.  The reference appends candidate pairs and contacts through device-wide atomic counters, so its row order (and
// with it every float-atomic sum over rows) changes from run to run; `deterministic=True` sorts afterwards.  Here every stage keeps
// a world's data in one contiguous, ordered range:
//   * one workgroup per world walks the world's (static) list of SDF shape pairs in ascending (shape0, shape1) order, tests the
//     gap-widened world AABBs the tile collide kernel exported, and compacts the hits with wave ballots -- the candidate list of a
//     world is sorted by construction, a scan over the worlds makes the flat pair index;
//   * the narrow phase (nt_sdf.hip, one workgroup per pair) allocates a pair's reduced rows as one block (the only atomic; the
//     block's position is arbitrary) and records (block offset, row count) per pair;
//   * two small scans give every pair its final row range -- world-major, pairs ascending, rows in fingerprint order -- and the
//     writer moves each raw row to its final place while converting it to Newton's body-frame contact record;
//   * per world, each body gets the list of row blocks that touch it (a pair's rows all touch the same two bodies), in ascending row
//     order: the XPBD apply phase and the penalty-force gather sum through it, ordered, without float atomics.
// Everything is HBM / L2 streaming work on small integer tables; no LDS tiling beyond the per-world scratch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"
#include "nt_math.hpp"

using namespace nt;

namespace {

NT_DI vec3 ld3(const float* p) { return vec3(p[0], p[1], p[2]); }
NT_DI void st3(float* p, vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// Newton shape id of template shape l (local block first, then the global shapes) in world w
NT_DI int global_shape_id(const nt_sdf_scene& sc, int l, int w) {
    return l < sc.ns ? sc.shape_local0 + w * sc.ns + l : sc.gshape_id[l - sc.ns];
}
// env-local body of Newton shape id `gid` seen from world w (-1: static / global shape)
NT_DI int shape_body_of(const nt_sdf_scene& sc, int gid, int w) {
    const int l = gid - sc.shape_local0 - w * sc.ns;
    return (l >= 0 && l < sc.ns) ? sc.shape_body[l] : -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// candidate pairs: one workgroup per world, ordered compaction of the template pair list
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sdf_pairs_kernel(nt_sdf_scene sc, const float* __restrict__ lower,
                                                        const float* __restrict__ upper, int32_t* __restrict__ world_pairs,
                                                        int32_t* __restrict__ pair_count) {
    __shared__ int wave_hits[4];
    __shared__ int base;
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int p0 = 0; p0 < sc.template_pairs; p0 += 256) {
        const int p = p0 + t;
        bool hit = false;
        int s1 = 0, s2 = 0;
        if (p < sc.template_pairs) {
            s1 = global_shape_id(sc, sc.template_pair[2 * p], w);
            s2 = global_shape_id(sc, sc.template_pair[2 * p + 1], w);
            if (s1 > s2) { const int x = s1; s1 = s2; s2 = x; }
            // check_aabb_overlap, cutoff 0 (the AABBs carry margin + gap), box1 = the smaller shape index
            const float *l1 = lower + 3 * (size_t)s1, *u1 = upper + 3 * (size_t)s1, *l2 = lower + 3 * (size_t)s2, *u2 = upper + 3 * (size_t)s2;
            hit = l1[0] <= u2[0] && u1[0] >= l2[0] && l1[1] <= u2[1] && u1[1] >= l2[1] && l1[2] <= u2[2] && u1[2] >= l2[2];
        }
        const unsigned long long mask = __ballot(hit);
        if (lane == 0) wave_hits[wave] = __popcll(mask);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wave; ++k) off += wave_hits[k];
        if (hit) {
            const int idx = off + __popcll(mask & ((1ull << lane) - 1ull));
            if (idx < sc.pairs_per_world) {
                world_pairs[2 * ((size_t)w * sc.pairs_per_world + idx)] = s1;
                world_pairs[2 * ((size_t)w * sc.pairs_per_world + idx) + 1] = s2;
                if (sc.template_kind) sc.world_pair_kind[(size_t)w * sc.pairs_per_world + idx] = sc.template_kind[p];
            }
        }
        __syncthreads();
        if (t == 0) base += wave_hits[0] + wave_hits[1] + wave_hits[2] + wave_hits[3];
        __syncthreads();
    }
    if (t == 0) pair_count[w] = base;  // keeps counting past the capacity (overflow is visible to the host), consumers clamp
}

// exclusive scan of min(in[i], clamp) over n <= a few 10^5 entries by ONE workgroup (per-world counts: n = world_count);
// out[n] = total.  1024 lanes, tiles of 1024 with a running carry.  Every output is limited to `cap` (the capacity of the array the
// prefix indexes): a consumer that walks [out[i], out[i + 1]) stays inside the array whatever the counts say; the unclamped
// counts stay in `in` for the host's overflow report.
__global__ void __launch_bounds__(1024) scan_worlds_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n, int clamp,
                                                           int cap) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + t;
        int v = 0;
        if (i < n) { v = in[i]; v = v < clamp ? v : clamp; v = v < 0 ? 0 : v; }
        int x = v;  // inclusive scan inside the wave
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int off = carry;
        for (int k = 0; k < wave; ++k) off += wsum[k];
        if (i < n) out[i] = (off + x - v) < cap ? (off + x - v) : cap;
        __syncthreads();
        if (t == 1023) carry = off + x;
        __syncthreads();
    }
    if (t == 0) out[n] = carry < cap ? carry : cap;
}

// per world: local exclusive scan of its pairs' row counts -> pair_row[w * PPW + k] (row offset inside the world), world_rows[w]
__global__ void __launch_bounds__(256) sdf_world_rows_kernel(nt_sdf_scene sc, const int32_t* __restrict__ pair_count,
                                                             const int32_t* __restrict__ blk, int32_t* __restrict__ pair_row,
                                                             int32_t* __restrict__ world_rows) {
    __shared__ int wsum[4];
    __shared__ int carry;
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int np = pair_count[w];
    np = np < sc.pairs_per_world ? np : sc.pairs_per_world;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int k0 = 0; k0 < np; k0 += 256) {
        const int k = k0 + t;
        const int v = k < np ? blk[2 * ((size_t)w * sc.pairs_per_world + k) + 1] : 0;
        int x = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int off = carry;
        for (int j = 0; j < wave; ++j) off += wsum[j];
        if (k < np) pair_row[(size_t)w * sc.pairs_per_world + k] = off + x - v;
        __syncthreads();
        if (t == 255) carry = off + x;
        __syncthreads();
    }
    if (t == 0) world_rows[w] = carry;
}

NT_DI xform body_xform(const float* body_q, int nb, int ES, int b, int env) {  // env-major SoA [7][nb][ES]
    auto at = [&](int c) { return body_q[((size_t)c * nb + b) * ES + env]; };
    return xform(vec3(at(0), at(1), at(2)), quat(at(3), at(4), at(5), at(6)));
}

// write_contact (collide.py:166-254) of one raw ContactData row i of pair position idx (world w) at its final, deterministic position
NT_DI void write_row(const nt_sdf_scene& sc, const nt_sdf_rows_io& io, const float* __restrict__ body_q, int idx, int w, int i,
                     int rank, bool hydro) {
    if (rank >= io.blk[2 * (size_t)idx + 1]) return;
    const int dst = io.row_start[w] + io.pair_row[idx] + rank;
    if (dst >= io.row_capacity) return;
    const int shape_a = io.world_pairs[2 * (size_t)idx], shape_b = io.world_pairs[2 * (size_t)idx + 1];
    const float* d = io.raw_data + 9 * (size_t)i;
    const float dist = d[6], margin_a = d[7], margin_b = d[8];
    // SDF / mesh shapes have no effective radius (compute_effective_radius); the triangle leg's partner can be a sphere / capsule
    const bool tri = io.raw_radius && sc.template_kind && sc.world_pair_kind[idx] == 3;
    const float ra = tri ? io.raw_radius[2 * (size_t)i] : 0.0f, rb = tri ? io.raw_radius[2 * (size_t)i + 1] : 0.0f;
    const float total = ra + rb + margin_a + margin_b;
    const vec3 nab = normalize(ld3(d + 3));
    const vec3 center = ld3(d);
    const vec3 aw = center - nab * (0.5f * dist + ra);
    const vec3 bw = center + nab * (0.5f * dist + rb);
    const float sep = dot(bw - aw, nab) - total;
    int sa = -1, sb = -1;
    vec3 p0, p1, o0, o1, nrm;
    float m0 = 0.0f, m1 = 0.0f;
    // decode_contacts_kernel hands its rows to the writer with a reserved index: no gap test (collide.py:246-252)
    if (hydro || !(sep > sc.shape_gap[shape_a] + sc.shape_gap[shape_b])) {
        sa = shape_a;
        sb = shape_b;
        const int ba = shape_body_of(sc, sa, w), bb = shape_body_of(sc, sb, w);
        const xform Xa = ba < 0 ? xform() : xform_inverse(body_xform(body_q, sc.nb, sc.env_stride, ba, w));
        const xform Xb = bb < 0 ? xform() : xform_inverse(body_xform(body_q, sc.nb, sc.env_stride, bb, w));
        m0 = ra + margin_a;
        m1 = rb + margin_b;
        p0 = xform_point(Xa, aw);
        p1 = xform_point(Xb, bw);
        o0 = xform_vector(Xa, m0 * nab);
        o1 = xform_vector(Xb, -m1 * nab);
        nrm = nab;
    }
    io.shape0[dst] = sa;
    io.shape1[dst] = sb;
    st3(io.point0 + 3 * (size_t)dst, p0);
    st3(io.point1 + 3 * (size_t)dst, p1);
    st3(io.offset0 + 3 * (size_t)dst, o0);
    st3(io.offset1 + 3 * (size_t)dst, o1);
    st3(io.normal + 3 * (size_t)dst, nrm);
    io.margin0[dst] = m0;
    io.margin1[dst] = m1;
    if (io.key) io.key[dst] = io.raw_key[i];
    if (io.stiffness) {
        io.stiffness[dst] = hydro && io.raw_stiffness ? io.raw_stiffness[i] : 0.0f;
        io.damping[dst] = 0.0f;
        io.friction_scale[dst] = hydro && io.raw_friction ? io.raw_friction[i] : 0.0f;
    }
}
// Raw rows come in two regions: [0, raw_base) holds the blocks the staged narrow phase wrote at the start of each pair's survivor
// block (gaps in between: walked pair by pair, eight lanes per candidate position); [raw_base, *raw_count) holds rows appended
// through the counter (single-kernel narrow phase, hydroelastic leg: ranked), one lane per row.
__global__ void __launch_bounds__(256) sdf_rows_write_kernel(nt_sdf_scene sc, nt_sdf_rows_io io, const float* __restrict__ body_q) {
    if (io.raw_base > 0) {
        const long long groups = (long long)sc.env_count * sc.pairs_per_world;
        const int sub = threadIdx.x & 7;
        for (long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; g < groups; g += ((long long)gridDim.x * blockDim.x) >> 3) {
            const int idx = (int)g;  // world * PPW + k
            const int w = idx / sc.pairs_per_world;
            int live = io.pair_count[w];
            live = live < sc.pairs_per_world ? live : sc.pairs_per_world;
            if (idx - w * sc.pairs_per_world >= live) continue;
            if (sc.template_kind && sc.world_pair_kind[idx] == 1) continue;  // appended rows
            const int off = io.blk[2 * (size_t)idx], cnt = io.blk[2 * (size_t)idx + 1];
            for (int i = off + sub; i < off + cnt && i < io.raw_base; i += 8) write_row(sc, io, body_q, idx, w, i, i - off, false);
        }
    }
    int n = *io.raw_count;
    n = n < io.raw_capacity ? n : io.raw_capacity;
    for (int i = io.raw_base + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int idx = io.raw_pair[i];  // world * PPW + k
        const int w = idx / sc.pairs_per_world;
        const bool hydro = sc.template_kind && sc.world_pair_kind[idx] == 1;  // rows of nt_hydro_pairs: ranked, pre-admitted
        write_row(sc, io, body_q, idx, w, i, hydro ? io.raw_rank[i] : i - io.blk[2 * (size_t)idx], hydro);
    }
}

// per world: the row blocks of every body, ascending.  Lane = body; the world's pair list is staged in LDS.
constexpr int BLK_PAIRS_LDS = 2560;  // (config C5 at Newton's default gap: all 2 336 pairs of a world are candidates)
__global__ void __launch_bounds__(256) sdf_body_blocks_kernel(nt_sdf_scene sc, nt_sdf_rows_io io, int32_t* __restrict__ body_blk_start,
                                                             int32_t* __restrict__ body_blk_list) {
    __shared__ int pa[BLK_PAIRS_LDS], pb[BLK_PAIRS_LDS], pr[BLK_PAIRS_LDS], pc[BLK_PAIRS_LDS];
    __shared__ int cnt[1024];
    const int w = blockIdx.x, t = threadIdx.x;
    int np = io.pair_count[w];
    np = np < sc.pairs_per_world ? np : sc.pairs_per_world;
    np = np < BLK_PAIRS_LDS ? np : BLK_PAIRS_LDS;
    const int row0 = io.row_start[w];
    for (int k = t; k < np; k += blockDim.x) {
        const size_t idx = (size_t)w * sc.pairs_per_world + k;
        pa[k] = shape_body_of(sc, io.world_pairs[2 * idx], w);
        pb[k] = shape_body_of(sc, io.world_pairs[2 * idx + 1], w);
        pr[k] = row0 + io.pair_row[idx];
        int c = io.blk[2 * idx + 1];  // a block ends where the row arrays do (rows beyond the capacity were never written)
        c = c < io.row_capacity - pr[k] ? c : io.row_capacity - pr[k];
        pc[k] = c > 0 ? c : 0;
    }
    __syncthreads();
    const int nb = sc.nb;  // <= 1024 (checked by the entry point)
    for (int b = t; b < nb; b += blockDim.x) {
        int c = 0;
        for (int k = 0; k < np; ++k)
            if (pc[k] > 0 && pa[k] != pb[k]) c += (pa[k] == b) + (pb[k] == b);
        cnt[b] = c;
    }
    __syncthreads();
    if (t == 0) {  // exclusive prefix over the bodies (tiny)
        int run = w * 2 * sc.pairs_per_world;
        for (int b = 0; b < nb; ++b) {
            const int c = cnt[b];
            cnt[b] = run;
            body_blk_start[(size_t)w * (nb + 1) + b] = run;
            run += c;
        }
        body_blk_start[(size_t)w * (nb + 1) + nb] = run;
    }
    __syncthreads();
    for (int b = t; b < nb; b += blockDim.x) {
        int o = cnt[b];
        for (int k = 0; k < np; ++k) {
            if (pc[k] <= 0 || pa[k] == pb[k]) continue;
            if (pa[k] == b) { body_blk_list[2 * (size_t)o] = pr[k] << 1; body_blk_list[2 * (size_t)o + 1] = pc[k]; ++o; }
            if (pb[k] == b) { body_blk_list[2 * (size_t)o] = (pr[k] << 1) | 1; body_blk_list[2 * (size_t)o + 1] = pc[k]; ++o; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// eval_body_contact over the flat rows of one world (one workgroup per world): per-row wrench records, then the ordered sum of
// every body through its block list, added to body_f (env-major).  No float atomics.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) flat_rows_forces_kernel(nt_sdf_scene sc, nt_flat_rows f, nt_flat_force_params p) {
    const int w = blockIdx.x, t = threadIdx.x;
    const int nb = sc.nb, ES = sc.env_stride;
    for (int r = f.row_start[w] + t; r < f.row_start[w + 1]; r += blockDim.x) {
        float* o = f.cw + 10 * (size_t)r;
        float flags = 0.0f;
        vec3 f_total, tq_a, tq_b;
        const int shape_a = f.shape0[r], shape_b = f.shape1[r];
        if (shape_a != shape_b) {
            float ke = 0.0f, kd = 0.0f, kf = 0.0f, ka = 0.0f, mu = 0.0f;
            int mat_nonzero = 0, body_a = -1, body_b = -1;
            if (shape_a >= 0) {
                mat_nonzero += 1;
                const float* m = p.shape_material + 5 * (size_t)shape_a;
                ke += m[0]; kd += m[1]; kf += m[2]; ka += m[3]; mu += m[4];
                body_a = shape_body_of(sc, shape_a, w);
            }
            if (shape_b >= 0) {
                mat_nonzero += 1;
                const float* m = p.shape_material + 5 * (size_t)shape_b;
                ke += m[0]; kd += m[1]; kf += m[2]; ka += m[3]; mu += m[4];
                body_b = shape_body_of(sc, shape_b, w);
            }
            if (mat_nonzero > 0) {
                ke /= float(mat_nonzero); kd /= float(mat_nonzero); kf /= float(mat_nonzero);
                ka /= float(mat_nonzero); mu /= float(mat_nonzero);
            }
            if (f.stiffness) {  // per-contact overrides (kernels_contact.py:452-459)
                const float cke = f.stiffness[r], ckd = f.damping[r], cmu = f.friction_scale[r];
                ke = cke > 0.0f ? cke : ke;
                kd = ckd > 0.0f ? ckd : kd;
                mu = cmu > 0.0f ? mu * cmu : mu;
            }
            const vec3 nrm = -ld3(f.normal + 3 * (size_t)r);
            vec3 bx_a = ld3(f.point0 + 3 * (size_t)r), bx_b = ld3(f.point1 + 3 * (size_t)r);
            const float margin_a = f.margin0[r], margin_b = f.margin1[r];
            vec3 r_a(0.0f), r_b(0.0f);
            if (body_a >= 0) {
                const xform X = body_xform(p.body_q, nb, ES, body_a, w);
                bx_a = xform_point(X, bx_a) - margin_a * nrm;
                const float* cm = p.body_com + ((size_t)0 * nb + body_a) * ES + w;  // body_param rows 0..2 = com, env-major
                r_a = bx_a - xform_point(X, vec3(cm[0], cm[(size_t)nb * ES], cm[2 * (size_t)nb * ES]));
            }
            if (body_b >= 0) {
                const xform X = body_xform(p.body_q, nb, ES, body_b, w);
                bx_b = xform_point(X, bx_b) + margin_b * nrm;
                const float* cm = p.body_com + ((size_t)0 * nb + body_b) * ES + w;
                r_b = bx_b - xform_point(X, vec3(cm[0], cm[(size_t)nb * ES], cm[2 * (size_t)nb * ES]));
            }
            const float d = dot(nrm, bx_a - bx_b);
            if (d < ka && body_a != body_b) {
                auto vel = [&](int b, vec3 rr) {
                    auto at = [&](int c) { return p.body_qd[((size_t)c * nb + b) * ES + w]; };
                    return vec3(at(0), at(1), at(2)) + cross(vec3(at(3), at(4), at(5)), rr);
                };
                vec3 bv_a(0.0f), bv_b(0.0f);
                if (body_a >= 0) bv_a = vel(body_a, r_a);
                if (body_b >= 0) bv_b = vel(body_b, r_b);
                const vec3 v = bv_a - bv_b;
                const float vn = dot(nrm, v);
                const vec3 vt = v - nrm * vn;
                const float fn = d * ke;
                const float fd = fminw(vn, 0.0f) * kd * (d < 0.0f ? 1.0f : 0.0f);
                vec3 ft(0.0f);
                if (d < 0.0f) {
                    const float delta = p.friction_smoothing;
                    const float a2 = dot(vt, vt);  // wp.norm_huber
                    const float vs = a2 <= delta * delta ? 0.5f * a2 : delta * (sqrtf(a2) - 0.5f * delta);
                    if (vs > 0.0f) {
                        const vec3 fr = vt / vs;
                        ft = fr * fminw(kf * vs, -mu * (fn + fd));
                    }
                }
                f_total = nrm * (fn + fd) + ft;
                tq_a = cross(r_a, f_total);
                tq_b = cross(r_b, f_total);
                flags = (body_a >= 0 ? 1.0f : 0.0f) + (body_b >= 0 ? 2.0f : 0.0f);
            }
        }
        st3(o, f_total); st3(o + 3, tq_a); st3(o + 6, tq_b);
        o[9] = flags;
    }
    __threadfence_block();
    __syncthreads();
    for (int b = t; b < nb; b += blockDim.x) {
        const int* bs = f.body_blk_start + (size_t)w * (nb + 1) + b;
        if (bs[0] == bs[1]) continue;
        vec3 ff, tt;
        for (int i = bs[0]; i < bs[1]; ++i) {
            const int code = f.body_blk_list[2 * (size_t)i], count = f.body_blk_list[2 * (size_t)i + 1];
            const int r0 = code >> 1, side = code & 1;
            for (int r = r0; r < r0 + count; ++r) {
                const float* o = f.cw + 10 * (size_t)r;
                if (((int)o[9] & (side ? 2 : 1)) == 0) continue;
                if (side == 0) { ff -= ld3(o); tt -= ld3(o + 3); }   // atomic_sub on shape0's body (kernels_contact.py:548-551)
                else { ff += ld3(o); tt += ld3(o + 6); }
            }
        }
        auto acc = [&](int c, float v) { p.body_f[((size_t)c * nb + b) * ES + w] += v; };
        acc(0, ff.x); acc(1, ff.y); acc(2, ff.z); acc(3, tt.x); acc(4, tt.y); acc(5, tt.z);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// frame-to-frame matching of the rows (geometry/contact_match.py:266-391,442-480,530-562).  Eight lanes per (world, candidate pair),
// the grouping of sdf_rows_write_kernel: a pair's rows of this frame and of the previous one are two short contiguous blocks.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int MATCH_NOT_FOUND = -1, MATCH_BROKEN = -2;
constexpr unsigned long long CLAIM_SENTINEL = ~0ull;

struct PairBlock { int w, s0, s1, row0, rows; };
NT_DI bool pair_block(const nt_sdf_scene& sc, const nt_sdf_rows_io& io, int idx, PairBlock& b) {
    b.w = idx / sc.pairs_per_world;
    int live = io.pair_count[b.w];
    live = live < sc.pairs_per_world ? live : sc.pairs_per_world;
    if (idx - b.w * sc.pairs_per_world >= live) return false;
    b.s0 = io.world_pairs[2 * (size_t)idx];
    b.s1 = io.world_pairs[2 * (size_t)idx + 1];
    b.row0 = io.row_start[b.w] + io.pair_row[idx];
    int c = (sc.template_kind && sc.world_pair_kind[idx] == 1) ? 0 : io.blk[2 * (size_t)idx + 1];  // hydroelastic rows do not match
    c = c < io.row_capacity - b.row0 ? c : io.row_capacity - b.row0;
    b.rows = c > 0 ? c : 0;
    return true;
}
NT_DI bool row_live(const nt_sdf_rows_io& io, int r) { return io.shape0[r] >= 0 && io.shape0[r] != io.shape1[r]; }
// world-space contact points of row r (body-frame records through the body transforms of world w)
NT_DI void row_points_world(const nt_sdf_scene& sc, const nt_sdf_rows_io& io, const float* __restrict__ body_q, int w, int r, vec3& p0,
                            vec3& p1) {
    p0 = ld3(io.point0 + 3 * (size_t)r);
    p1 = ld3(io.point1 + 3 * (size_t)r);
    const int b0 = shape_body_of(sc, io.shape0[r], w), b1 = shape_body_of(sc, io.shape1[r], w);
    if (b0 >= 0) p0 = xform_point(body_xform(body_q, sc.nb, sc.env_stride, b0, w), p0);
    if (b1 >= 0) p1 = xform_point(body_xform(body_q, sc.nb, sc.env_stride, b1, w), p1);
}
// low 32 bits of the reference's sort key (contact_data.py:60-90): shape1's low 9 bits, then the 23-bit sub key (the fingerprint)
NT_DI unsigned key_low32(const nt_sdf_rows_io& io, int r, int rank) {
    const unsigned sub = io.key ? (unsigned)io.key[r] : (unsigned)rank;
    return (((unsigned)io.shape1[r] & 0x1FFu) << 23) | (sub & 0x7FFFFFu);
}
// _pack_claim: float_flip(dist_sq) of a non-negative float in the high word, the key bits in the low word
NT_DI unsigned long long pack_claim(float dist_sq, unsigned key_low) {
    const unsigned flipped = __float_as_uint(dist_sq) ^ ((unsigned)(-(int)(__float_as_uint(dist_sq) >> 31)) | 0x80000000u);
    return ((unsigned long long)flipped << 32) | key_low;
}
#define NT_FOR_PAIR_GROUPS(sc, g)                                                                                     \
    for (long long g = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3,                                       \
                   g##_n = (long long)(sc).env_count * (sc).pairs_per_world;                                          \
         g < g##_n; g += ((long long)gridDim.x * blockDim.x) >> 3)

// pass 1 (_match_contacts_kernel): the closest previous row of the same shape pair within the thresholds, and a claim on it
__global__ void __launch_bounds__(256) flat_rows_match_kernel(nt_sdf_scene sc, nt_sdf_rows_io io, const float* __restrict__ body_q,
                                                              nt_flat_history h, float pos_threshold_sq, float normal_dot_threshold,
                                                              int32_t* __restrict__ match_index) {
    const int sub = threadIdx.x & 7;
    NT_FOR_PAIR_GROUPS(sc, g) {
        PairBlock b;
        if (!pair_block(sc, io, (int)g, b)) continue;
        // the previous frame's block of this shape pair: its world's candidate list is ascending in the canonical (min, max) shape
        // order.  The history stores, and the search compares, that canonical key: the vertex leg rewrites a (mesh, plane) pair of the
        // live list in contact orientation, which is (max, min) when the plane was added first (nt_mesh_plane.hip)
        const size_t base = (size_t)b.w * sc.pairs_per_world;
        const int n_prev = h.prev_pair_count[b.w];
        const int k0 = b.s0 < b.s1 ? b.s0 : b.s1, k1 = b.s0 < b.s1 ? b.s1 : b.s0;
        int lo = 0, hi = n_prev;
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            const int a0 = h.prev_world_pairs[2 * (base + mid)], a1 = h.prev_world_pairs[2 * (base + mid) + 1];
            if (a0 < k0 || (a0 == k0 && a1 < k1)) lo = mid + 1; else hi = mid;
        }
        int prow0 = 0, prows = 0;
        if (lo < n_prev && h.prev_world_pairs[2 * (base + lo)] == k0 && h.prev_world_pairs[2 * (base + lo) + 1] == k1) {
            prow0 = h.prev_row_start[b.w] + h.prev_pair_row[base + lo];
            prows = h.prev_pair_rows[base + lo];
        }
        bool any_prev = false;  // the key range of the pair counts contacts, not inert rows
        for (int j = 0; j < prows && !any_prev; ++j) any_prev = h.prev_live[prow0 + j] != 0;
        for (int r = b.row0 + sub; r < b.row0 + b.rows; r += 8) {
            if (!row_live(io, r) || !any_prev) { match_index[r] = MATCH_NOT_FOUND; continue; }
            vec3 p0, p1;
            row_points_world(sc, io, body_q, b.w, r, p0, p1);
            const vec3 pos = 0.5f * (p0 + p1), n = ld3(io.normal + 3 * (size_t)r);
            int best = -1;
            float best_dist_sq = pos_threshold_sq;
            for (int j = prow0; j < prow0 + prows; ++j) {
                if (!h.prev_live[j]) continue;
                const vec3 d = pos - ld3(h.prev_pos_world + 3 * (size_t)j);
                const float dist_sq = dot(d, d);
                if (dist_sq <= best_dist_sq && dot(n, ld3(h.prev_normal + 3 * (size_t)j)) >= normal_dot_threshold) {
                    best_dist_sq = dist_sq;
                    best = j;
                }
            }
            if (best >= 0) {
                match_index[r] = best;
                atomicMin(h.prev_claim + best, pack_claim(best_dist_sq, key_low32(io, r, r - b.row0)));
            } else {
                match_index[r] = MATCH_BROKEN;
            }
        }
    }
}
// pass 2 (_resolve_claims_kernel): the claim word names the winner by its key bits; everyone else is MATCH_BROKEN
__global__ void __launch_bounds__(256) flat_rows_resolve_kernel(nt_sdf_scene sc, nt_sdf_rows_io io, nt_flat_history h,
                                                                int32_t* __restrict__ match_index) {
    const int sub = threadIdx.x & 7;
    NT_FOR_PAIR_GROUPS(sc, g) {
        PairBlock b;
        if (!pair_block(sc, io, (int)g, b)) continue;
        for (int r = b.row0 + sub; r < b.row0 + b.rows; r += 8) {
            const int cand = match_index[r];
            if (cand < 0) continue;
            if ((unsigned)(h.prev_claim[cand] & 0xFFFFFFFFull) != key_low32(io, r, r - b.row0)) match_index[r] = MATCH_BROKEN;
        }
    }
}
// _replay_matched_kernel: matched rows that still touch keep the record used last frame
__global__ void __launch_bounds__(256) flat_rows_replay_kernel(nt_sdf_scene sc, nt_sdf_rows_io io, const float* __restrict__ body_q,
                                                               nt_flat_history h, const int32_t* __restrict__ match_index) {
    const int sub = threadIdx.x & 7;
    NT_FOR_PAIR_GROUPS(sc, g) {
        PairBlock b;
        if (!pair_block(sc, io, (int)g, b)) continue;
        for (int r = b.row0 + sub; r < b.row0 + b.rows; r += 8) {
            if (!row_live(io, r)) continue;
            const int m = match_index[r];
            if (m < 0) continue;
            vec3 p0, p1;
            row_points_world(sc, io, body_q, b.w, r, p0, p1);
            const float fresh_gap = dot(p1 - p0, ld3(io.normal + 3 * (size_t)r)) - (io.margin0[r] + io.margin1[r]);
            if (fresh_gap > 0.0f) continue;
            const float* f = h.prev_body_frame + 12 * (size_t)m;
            st3(io.point0 + 3 * (size_t)r, ld3(f));
            st3(io.point1 + 3 * (size_t)r, ld3(f + 3));
            st3(io.offset0 + 3 * (size_t)r, ld3(f + 6));
            st3(io.offset1 + 3 * (size_t)r, ld3(f + 9));
            st3(io.normal + 3 * (size_t)r, ld3(h.prev_normal + 3 * (size_t)m));
        }
    }
}
// _save_sorted_state_kernel: this frame's pair tables, midpoints, normals (and body-frame records) become the history
__global__ void __launch_bounds__(256) flat_rows_save_kernel(nt_sdf_scene sc, nt_sdf_rows_io io, const float* __restrict__ body_q,
                                                             nt_flat_history h) {
    const int sub = threadIdx.x & 7;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w <= sc.env_count; w += gridDim.x * blockDim.x) {
        h.prev_row_start[w] = io.row_start[w];
        if (w < sc.env_count) {
            const int c = io.pair_count[w];
            h.prev_pair_count[w] = c < sc.pairs_per_world ? c : sc.pairs_per_world;
        }
    }
    NT_FOR_PAIR_GROUPS(sc, g) {
        PairBlock b;
        if (!pair_block(sc, io, (int)g, b)) continue;
        if (sub == 0) {
            h.prev_world_pairs[2 * (size_t)g] = b.s0 < b.s1 ? b.s0 : b.s1;  // canonical key (see flat_rows_match_kernel)
            h.prev_world_pairs[2 * (size_t)g + 1] = b.s0 < b.s1 ? b.s1 : b.s0;
            h.prev_pair_row[g] = io.pair_row[g];
            h.prev_pair_rows[g] = b.rows;
        }
        for (int r = b.row0 + sub; r < b.row0 + b.rows; r += 8) {
            const bool live = row_live(io, r);
            h.prev_live[r] = live ? 1 : 0;
            h.prev_claim[r] = CLAIM_SENTINEL;
            if (!live) continue;
            vec3 p0, p1;
            row_points_world(sc, io, body_q, b.w, r, p0, p1);
            st3(h.prev_pos_world + 3 * (size_t)r, 0.5f * (p0 + p1));
            st3(h.prev_normal + 3 * (size_t)r, ld3(io.normal + 3 * (size_t)r));
            if (h.prev_body_frame) {
                float* f = h.prev_body_frame + 12 * (size_t)r;
                st3(f, ld3(io.point0 + 3 * (size_t)r));
                st3(f + 3, ld3(io.point1 + 3 * (size_t)r));
                st3(f + 6, ld3(io.offset0 + 3 * (size_t)r));
                st3(f + 9, ld3(io.offset1 + 3 * (size_t)r));
            }
        }
    }
}

bool scene_ok(const nt_sdf_scene* sc) {
    return sc && sc->env_count > 0 && sc->env_stride >= sc->env_count && sc->nb >= 0 && sc->ns >= 0 && sc->pairs_per_world > 0 &&
           sc->template_pairs >= 0 && (sc->template_pairs == 0 || sc->template_pair) && sc->shape_body && sc->shape_gap;
}

bool match_args_ok(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, const nt_flat_history* h) {
    return scene_ok(sc) && io && body_q && h && io->pair_count && io->world_pairs && io->blk && io->pair_row && io->row_start &&
           io->shape0 && io->shape1 && io->point0 && io->point1 && io->offset0 && io->offset1 && io->normal && io->margin0 &&
           io->margin1 && io->row_capacity > 0 && h->prev_row_start && h->prev_pair_count && h->prev_world_pairs && h->prev_pair_row &&
           h->prev_pair_rows && h->prev_live && h->prev_pos_world && h->prev_normal && h->prev_claim &&
           (!sc->template_kind || sc->world_pair_kind);
}
int pair_group_blocks(const nt_sdf_scene* sc) {
    const long long wb = ((long long)sc->env_count * sc->pairs_per_world * 8 + 255) / 256;
#ifdef NT_EMULATED_GRID
    return (int)(wb < NT_EMULATED_GRID ? wb : NT_EMULATED_GRID);
#else
    return (int)(wb < 16384 ? wb : 16384);
#endif
}

}  // namespace

extern "C" {

nt_status nt_sdf_candidate_pairs(const nt_sdf_scene* sc, const float* aabb_lower, const float* aabb_upper, int32_t* world_pairs,
                                 int32_t* pair_count, int32_t* pair_prefix, void* stream) {
    if (!scene_ok(sc) || !aabb_lower || !aabb_upper || !world_pairs || !pair_count || !pair_prefix) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sdf_pairs_kernel, dim3(sc->env_count), dim3(256), 0, (hipStream_t)stream, *sc, aabb_lower, aabb_upper,
                       world_pairs, pair_count);
    hipLaunchKernelGGL(scan_worlds_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pair_count, pair_prefix, sc->env_count,
                       sc->pairs_per_world, 0x7fffffff);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_sdf_rows_finalize(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, int32_t* world_rows,
                               int32_t* body_blk_start, int32_t* body_blk_list, void* stream) {
    if (!scene_ok(sc) || !io || !body_q || !world_rows || !body_blk_start || !body_blk_list) return NT_ERR_INVALID_ARG;
    if (!io->pair_count || !io->world_pairs || !io->blk || !io->pair_row || !io->row_start || !io->raw_count || !io->raw_pair ||
        !io->raw_data || !io->shape0 || !io->shape1 || !io->point0 || !io->point1 || !io->offset0 || !io->offset1 || !io->normal ||
        !io->margin0 || !io->margin1 || io->raw_capacity <= 0 || io->row_capacity <= 0)
        return NT_ERR_INVALID_ARG;
    if (sc->nb > 1024) return NT_ERR_UNSUPPORTED;
    if (sc->pairs_per_world > BLK_PAIRS_LDS) return NT_ERR_UNSUPPORTED;  // sdf_body_blocks_kernel stages a world's pair list in LDS
    if (io->stiffness && (!io->damping || !io->friction_scale)) return NT_ERR_INVALID_ARG;
    if (sc->template_kind && (!sc->world_pair_kind || !io->raw_rank)) return NT_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sdf_world_rows_kernel, dim3(sc->env_count), dim3(256), 0, st, *sc, io->pair_count, io->blk, io->pair_row, world_rows);
    // rows that no writer reaches (raw rows dropped by a full raw buffer, rows of a frame that had more of them) must not survive
    // from the previous call: every consumer skips a row with shape0 == shape1
    if (hipMemsetAsync(io->shape0, 0xff, sizeof(int32_t) * (size_t)io->row_capacity, st) != hipSuccess ||
        hipMemsetAsync(io->shape1, 0xff, sizeof(int32_t) * (size_t)io->row_capacity, st) != hipSuccess)
        return NT_ERR_LAUNCH;
    // row_start is limited to the row capacity: [row_start[w], row_start[w + 1]) never leaves the FlatRows arrays
    hipLaunchKernelGGL(scan_worlds_kernel, dim3(1), dim3(1024), 0, st, world_rows, io->row_start, sc->env_count, 0x7fffffff,
                       io->row_capacity);
    long long wb = ((long long)sc->env_count * sc->pairs_per_world * 8 + 255) / 256;
#ifdef NT_EMULATED_GRID
    int blocks = (int)(wb < NT_EMULATED_GRID ? wb : NT_EMULATED_GRID);
#else
    int blocks = (int)(wb < 16384 ? wb : 16384);
#endif
    hipLaunchKernelGGL(sdf_rows_write_kernel, dim3(blocks), dim3(256), 0, st, *sc, *io, body_q);
    hipLaunchKernelGGL(sdf_body_blocks_kernel, dim3(sc->env_count), dim3(256), 0, st, *sc, *io, body_blk_start, body_blk_list);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_flat_rows_forces(const nt_sdf_scene* sc, const nt_flat_rows* rows, const nt_flat_force_params* p, void* stream) {
    if (!scene_ok(sc) || !rows || !p || !rows->row_start || !rows->shape0 || !rows->shape1 || !rows->point0 || !rows->point1 ||
        !rows->normal || !rows->margin0 || !rows->margin1 || !rows->body_blk_start || !rows->body_blk_list || !rows->cw ||
        !p->body_q || !p->body_qd || !p->body_com || !p->shape_material || !p->body_f)
        return NT_ERR_INVALID_ARG;
    if (rows->stiffness && (!rows->damping || !rows->friction_scale)) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(flat_rows_forces_kernel, dim3(sc->env_count), dim3(256), 0, (hipStream_t)stream, *sc, *rows, *p);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_flat_rows_match(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, const nt_flat_history* h,
                             float pos_threshold, float normal_dot_threshold, int32_t* match_index, void* stream) {
    if (!match_args_ok(sc, io, body_q, h) || !match_index || !(pos_threshold >= 0.0f)) return NT_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(match_index, 0xff, sizeof(int32_t) * (size_t)io->row_capacity, st) != hipSuccess) return NT_ERR_LAUNCH;
    const int blocks = pair_group_blocks(sc);
    hipLaunchKernelGGL(flat_rows_match_kernel, dim3(blocks), dim3(256), 0, st, *sc, *io, body_q, *h, pos_threshold * pos_threshold,
                       normal_dot_threshold, match_index);
    hipLaunchKernelGGL(flat_rows_resolve_kernel, dim3(blocks), dim3(256), 0, st, *sc, *io, *h, match_index);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_flat_rows_replay_matched(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, const nt_flat_history* h,
                                      const int32_t* match_index, void* stream) {
    if (!match_args_ok(sc, io, body_q, h) || !match_index || !h->prev_body_frame) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(flat_rows_replay_kernel, dim3(pair_group_blocks(sc)), dim3(256), 0, (hipStream_t)stream, *sc, *io, body_q, *h,
                       match_index);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_flat_rows_save_history(const nt_sdf_scene* sc, const nt_sdf_rows_io* io, const float* body_q, const nt_flat_history* h,
                                    void* stream) {
    if (!match_args_ok(sc, io, body_q, h)) return NT_ERR_INVALID_ARG;
    // rows outside this frame's pair blocks are not contacts of the history
    if (hipMemsetAsync(h->prev_live, 0, (size_t)io->row_capacity, (hipStream_t)stream) != hipSuccess) return NT_ERR_LAUNCH;
    hipLaunchKernelGGL(flat_rows_save_kernel, dim3(pair_group_blocks(sc)), dim3(256), 0, (hipStream_t)stream, *sc, *io, body_q, *h);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
