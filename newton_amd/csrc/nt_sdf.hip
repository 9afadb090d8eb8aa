// nt_sdf.hip -- sparse "texture" SDF sampling and the mesh-vs-SDF narrow phase for gfx950.
//
// Reference behaviour (paths under /root/reference/newton/_src/geometry):
//   sampling        sdf_texture.py:786-828 (_locate_cell_coords), :1008-1126 (_texture_sample_sdf_variant: software trilinear
//                   over point-sampled texels, |p - clamp(p)| extension outside the box; nt_sdf_sample's value, the hydroelastic
//                   kernel), :1415-1538 (the one-fetch "hardware" sampler the mesh-SDF narrow phase uses), :1619-1697
//                   (centred-difference gradient from six such fetches, clamp-direction gradient outside the box)
//   edge search     sdf_contact.py:704-938 (do_edge_sdf_collision: symmetric golden pair + <= 3 Brent steps + endpoint checks)
//   pair kernel     sdf_contact.py:1098-1515 (mesh_sdf_collision_kernel, reduce_contacts=False variant: both modes of a pair,
//                   edge bounding-sphere cull against the SDF box and the midpoint value, inner-cull consistency, corner
//                   ownership, contact = (point, -/+ normalised gradient, distance))
//
// MI355X design: the reference stores the grids in CUDA 3-D textures and point-samples them at texel centres; here they are
// plain linear arrays (coarse float grid, packed (subgrid_size+1)^3 blocks in float / uint16 / uint8, indirection slots) read
// with ordinary global loads -- eight texels of a cell sit in two 64-B segments of an x-row pair.  Where the reference's CUDA
// path takes hardware-filtered fetches (8-bit weights), this path interpolates in full fp32 like its Warp-CPU path.
// One workgroup walks the pairs with stride gridDim.x; its 256 lanes take the edges of the "triangle" shape of the pair (both
// modes), cull, search and append contacts through a device-wide atomic counter (the reference appends atomically too); the
// sort key (pair, edge, mode) rides along so the host can order them deterministically.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/newton_hip.h"
#include "nt_math.hpp"

using namespace nt;

namespace {

constexpr uint32_t SLOT_LINEAR = 0xFFFFFFFEu;

NT_DI float texel(const nt_sdf& s, int x, int y, int z) {  // subgrid "texture" read at texel (x, y, z), normalised formats -> [0,1]
    const size_t i = ((size_t)z * s.tex_size + y) * s.tex_size + x;
    if (s.quantization == 4) return reinterpret_cast<const float*>(s.subgrid)[i];
    if (s.quantization == 2) return (float)reinterpret_cast<const uint16_t*>(s.subgrid)[i] * (1.0f / 65535.0f);
    return (float)reinterpret_cast<const uint8_t*>(s.subgrid)[i] * (1.0f / 255.0f);
}

// two x-adjacent texels (x, x + 1 in the same row) with ONE load: the uint16 / uint8 formats pack them into a dword / a word
// (2-byte aligned dword loads are legal on gfx950), float32 into a dwordx2.  Same values as two texel() calls, half the requests
// of a sampler whose eight taps are four such pairs.
NT_DI void texel_pair(const nt_sdf& s, int x, int y, int z, float& v0, float& v1) {
    const size_t i = ((size_t)z * s.tex_size + y) * s.tex_size + x;
    if (s.quantization == 4) {
        float w[2];
        __builtin_memcpy(w, reinterpret_cast<const float*>(s.subgrid) + i, 8);
        v0 = w[0]; v1 = w[1];
    } else if (s.quantization == 2) {
        uint32_t w;
        __builtin_memcpy(&w, reinterpret_cast<const uint16_t*>(s.subgrid) + i, 4);
        v0 = (float)(w & 0xFFFFu) * (1.0f / 65535.0f);
        v1 = (float)(w >> 16) * (1.0f / 65535.0f);
    } else {
        uint16_t w;
        __builtin_memcpy(&w, reinterpret_cast<const uint8_t*>(s.subgrid) + i, 2);
        v0 = (float)(w & 0xFFu) * (1.0f / 255.0f);
        v1 = (float)(w >> 8) * (1.0f / 255.0f);
    }
}

struct Cell {
    int ix, iy, iz, bx, by, bz;
    float tx, ty, tz;
    uint32_t slot;
};
NT_DI int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

NT_DI Cell locate(const nt_sdf& s, vec3 f) {  // _locate_cell
    Cell c;
    const float ssf = (float)s.subgrid_size;
    const float fvx = (float)s.cx * ssf, fvy = (float)s.cy * ssf, fvz = (float)s.cz * ssf;
    const float fx = clampf(f.x, 0.0f, fvx), fy = clampf(f.y, 0.0f, fvy), fz = clampf(f.z, 0.0f, fvz);
    c.ix = clampi((int)floorf(fx), 0, (int)fvx - 1);
    c.iy = clampi((int)floorf(fy), 0, (int)fvy - 1);
    c.iz = clampi((int)floorf(fz), 0, (int)fvz - 1);
    c.tx = fx - (float)c.ix;
    c.ty = fy - (float)c.iy;
    c.tz = fz - (float)c.iz;
    const float f2c = 1.0f / ssf;
    c.bx = clampi((int)((float)c.ix * f2c), 0, s.cx - 1);
    c.by = clampi((int)((float)c.iy * f2c), 0, s.cy - 1);
    c.bz = clampi((int)((float)c.iz * f2c), 0, s.cz - 1);
    c.slot = s.slots[((size_t)c.bx * s.cy + c.by) * s.cz + c.bz];
    return c;
}

// value at a point already clamped to the SDF box (+ the caller's |p - clamped| extension)
NT_DI float sample_clamped(const nt_sdf& s, vec3 clamped, float diff_mag) {
    const vec3 lo(s.box_lower[0], s.box_lower[1], s.box_lower[2]);
    vec3 f = cw_mul(clamped - lo, vec3(s.inv_dx[0], s.inv_dx[1], s.inv_dx[2]));
    Cell c = locate(s, f);
    float v000, v100, v010, v110, v001, v101, v011, v111;
    float tx = c.tx, ty = c.ty, tz = c.tz;
    bool needs_scale = false;
    if (c.slot >= SLOT_LINEAR) {
        const float f2c = 1.0f / (float)s.subgrid_size;
        tx = ((float)c.ix + c.tx) * f2c - (float)c.bx;
        ty = ((float)c.iy + c.ty) * f2c - (float)c.by;
        tz = ((float)c.iz + c.tz) * f2c - (float)c.bz;
        const int sx = s.cx + 1, sy = s.cy + 1;
        const float* g = s.coarse + ((size_t)c.bz * sy + c.by) * sx + c.bx;
        v000 = g[0]; v100 = g[1]; v010 = g[sx]; v110 = g[sx + 1];
        g += (size_t)sx * sy;
        v001 = g[0]; v101 = g[1]; v011 = g[sx]; v111 = g[sx + 1];
    } else {
        needs_scale = true;
        const int spd = s.subgrid_size + 1;
        const int ox = (int)(c.slot & 0x3FFu) * spd + (c.ix - c.bx * s.subgrid_size);
        const int oy = (int)((c.slot >> 10) & 0x3FFu) * spd + (c.iy - c.by * s.subgrid_size);
        const int oz = (int)((c.slot >> 20) & 0x3FFu) * spd + (c.iz - c.bz * s.subgrid_size);
        texel_pair(s, ox, oy, oz, v000, v100);  // (ox + 1 <= the block's last sample: the cell index inside the block is < subgrid_size)
        texel_pair(s, ox, oy + 1, oz, v010, v110);
        texel_pair(s, ox, oy, oz + 1, v001, v101);
        texel_pair(s, ox, oy + 1, oz + 1, v011, v111);
    }
    float c00 = v000 + (v100 - v000) * tx;
    float c10 = v010 + (v110 - v010) * tx;
    float c01 = v001 + (v101 - v001) * tx;
    float c11 = v011 + (v111 - v011) * tx;
    float c0 = c00 + (c10 - c00) * ty;
    float c1 = c01 + (c11 - c01) * ty;
    float val = c0 + (c1 - c0) * tz;
    if (needs_scale) val = val * s.value_range + s.min_value;
    return val + diff_mag;
}

// The "hardware" fetch of the mesh-SDF narrow phase (sdf_texture.py:1415-1461, _texture_sample_sdf_hw_clamped_variant): ONE
// filtered texture sample at a fractional coordinate -- block origin + 0.5 + t for a subgrid cell, coarse cell + 0.5 + t for a
// linear cell -- which the texture unit (on Warp's CPU device: a float32 software filter) resolves as the trilinear blend of the
// eight texels around (u - 0.5): i0 = floor(u - 0.5), t' = (u - 0.5) - i0, CLAMP addressing.  The coordinate round trip is kept
// (it costs the low bits of t, so the result differs from the software sampler above in the last place); plain global loads.
NT_DI float fetch_linear(const nt_sdf& s, bool coarse, float ux, float uy, float uz) {
    const float x = ux - 0.5f, y = uy - 0.5f, z = uz - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y), fz0 = floorf(z);
    const float tx = x - fx0, ty = y - fy0, tz = z - fz0;
    const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    float v000, v100, v010, v110, v001, v101, v011, v111;
    if (coarse) {
        const int sx = s.cx + 1, sy = s.cy + 1, sz = s.cz + 1;
        const int xa = clampi(x0, 0, sx - 1), xb = clampi(x0 + 1, 0, sx - 1), ya = clampi(y0, 0, sy - 1), yb = clampi(y0 + 1, 0, sy - 1),
                  za = clampi(z0, 0, sz - 1), zb = clampi(z0 + 1, 0, sz - 1);
        const float* g = s.coarse;
        auto at = [&](int xx, int yy, int zz) { return g[((size_t)zz * sy + yy) * sx + xx]; };
        v000 = at(xa, ya, za); v100 = at(xb, ya, za); v010 = at(xa, yb, za); v110 = at(xb, yb, za);
        v001 = at(xa, ya, zb); v101 = at(xb, ya, zb); v011 = at(xa, yb, zb); v111 = at(xb, yb, zb);
    } else {
        const int T = s.tex_size;
        const int xa = clampi(x0, 0, T - 1), xb = clampi(x0 + 1, 0, T - 1), ya = clampi(y0, 0, T - 1), yb = clampi(y0 + 1, 0, T - 1),
                  za = clampi(z0, 0, T - 1), zb = clampi(z0 + 1, 0, T - 1);
        if (xb == xa + 1) {  // (always, except at the clamped border of the texture)
            texel_pair(s, xa, ya, za, v000, v100); texel_pair(s, xa, yb, za, v010, v110);
            texel_pair(s, xa, ya, zb, v001, v101); texel_pair(s, xa, yb, zb, v011, v111);
        } else {
            v000 = texel(s, xa, ya, za); v100 = texel(s, xb, ya, za); v010 = texel(s, xa, yb, za); v110 = texel(s, xb, yb, za);
            v001 = texel(s, xa, ya, zb); v101 = texel(s, xb, ya, zb); v011 = texel(s, xa, yb, zb); v111 = texel(s, xb, yb, zb);
        }
    }
    const float c00 = v000 + (v100 - v000) * tx;
    const float c10 = v010 + (v110 - v010) * tx;
    const float c01 = v001 + (v101 - v001) * tx;
    const float c11 = v011 + (v111 - v011) * tx;
    const float c0 = c00 + (c10 - c00) * ty;
    const float c1 = c01 + (c11 - c01) * ty;
    return c0 + (c1 - c0) * tz;
}
NT_DI float sample_hw_clamped(const nt_sdf& s, vec3 clamped, float diff_mag) {
    const vec3 lo(s.box_lower[0], s.box_lower[1], s.box_lower[2]);
    const vec3 f = cw_mul(clamped - lo, vec3(s.inv_dx[0], s.inv_dx[1], s.inv_dx[2]));
    const Cell c = locate(s, f);
    float val;
    if (c.slot >= SLOT_LINEAR) {
        const float f2c = 1.0f / (float)s.subgrid_size;
        const float cx = (float)c.bx, cy = (float)c.by, cz = (float)c.bz;
        const float fx = ((float)c.ix + c.tx) * f2c, fy = ((float)c.iy + c.ty) * f2c, fz = ((float)c.iz + c.tz) * f2c;
        val = fetch_linear(s, true, cx + (fx - cx) + 0.5f, cy + (fy - cy) + 0.5f, cz + (fz - cz) + 0.5f);
    } else {
        const float ssf = (float)s.subgrid_size, samples = (float)(s.subgrid_size + 1);
        const float bx = (float)(c.slot & 0x3FFu), by = (float)((c.slot >> 10) & 0x3FFu), bz = (float)((c.slot >> 20) & 0x3FFu);
        const float lx = (float)c.ix - (float)c.bx * ssf, ly = (float)c.iy - (float)c.by * ssf, lz = (float)c.iz - (float)c.bz * ssf;
        const float ox = bx * samples + lx + 0.5f, oy = by * samples + ly + 0.5f, oz = bz * samples + lz + 0.5f;
        const float raw = fetch_linear(s, false, ox + c.tx, oy + c.ty, oz + c.tz);
        val = raw * s.value_range + s.min_value;
    }
    return val + diff_mag;
}

NT_DI vec3 clamp_to_box(const nt_sdf& s, vec3 p) {
    return vec3(clampf(p.x, s.box_lower[0], s.box_upper[0]), clampf(p.y, s.box_lower[1], s.box_upper[1]),
                clampf(p.z, s.box_lower[2], s.box_upper[2]));
}
NT_DI float sample(const nt_sdf& s, vec3 p) {  // texture_sample_sdf
    vec3 c = clamp_to_box(s, p);
    vec3 d = p - c;
    return sample_clamped(s, c, sqrtf(dot(d, d)));
}
NT_DI float sample_hw(const nt_sdf& s, vec3 p) {  // texture_sample_sdf_hw (:1495-1538)
    vec3 c = clamp_to_box(s, p);
    vec3 d = p - c;
    float mag = 0.0f;
    if (d.x != 0.0f || d.y != 0.0f || d.z != 0.0f) mag = sqrtf(dot(d, d));
    return sample_hw_clamped(s, c, mag);
}
// _texture_sample_sdf_grad_hw_impl_variant: centred differences with half-voxel steps inside the box, the clamp direction outside
NT_DI vec3 sample_grad(const nt_sdf& s, vec3 p) {
    vec3 c = clamp_to_box(s, p);
    vec3 d = p - c;
    if (d.x != 0.0f || d.y != 0.0f || d.z != 0.0f) {
        float m = length(d);
        if (m > 0.0f) return d / m;
    }
    vec3 g;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float h = 0.5f / s.inv_dx[a];
        vec3 p0 = p, p1 = p;
        vset(p0, a, vget(p, a) + h);
        vset(p1, a, vget(p, a) - h);
        const float c0 = clampf(vget(p0, a), s.box_lower[a], s.box_upper[a]);
        const float c1 = clampf(vget(p1, a), s.box_lower[a], s.box_upper[a]);
        const float d0 = vget(p0, a) - c0, d1 = vget(p1, a) - c1;
        vec3 q0 = p0, q1 = p1;
        vset(q0, a, c0);
        vset(q1, a, c1);
        const float s0 = d0 * d0, s1 = d1 * d1;  // diff_sq of the clamped-pair fetch (:1267-1352): no root when it is zero
        const float v0 = sample_hw_clamped(s, q0, s0 != 0.0f ? sqrtf(s0) : 0.0f), v1 = sample_hw_clamped(s, q1, s1 != 0.0f ? sqrtf(s1) : 0.0f);
        vset(g, a, (v0 - v1) * s.inv_dx[a]);
    }
    return g;
}

__global__ void sdf_sample_kernel(nt_sdf s, const float* __restrict__ pts, int n, float* __restrict__ dist, float* __restrict__ grad) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vec3 p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (dist) dist[i] = sample(s, p);
    if (grad) {
        vec3 g = sample_grad(s, p);
        grad[3 * i] = g.x; grad[3 * i + 1] = g.y; grad[3 * i + 2] = g.z;
    }
}

__global__ void sdf_sample_hw_kernel(nt_sdf s, const float* __restrict__ pts, int n, float* __restrict__ dist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dist[i] = sample_hw(s, vec3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
}
NT_DI float sample_at_voxel(const nt_sdf& s, int ix, int iy, int iz);
__global__ void sdf_sample_voxels_kernel(nt_sdf s, const int* __restrict__ ijk, int n, float* __restrict__ dist) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dist[i] = sample_at_voxel(s, ijk[3 * i], ijk[3 * i + 1], ijk[3 * i + 2]);
}

// do_edge_sdf_collision (texture path): deepest point of the edge v0 -> v1 in the SDF; returns (distance, t, endpoint code)
NT_DI void edge_search(const nt_sdf& s, vec3 v0, vec3 v1, float midpoint_sdf, float precision_target, float& best_f, vec3& best_p,
                       int& best_endpoint) {
    const float golden = 0.3819660112501051f;
    const vec3 dir = v1 - v0;
    const float len_sq = length_sq(dir);
    float inv_len = 1.0e12f;
    if (len_sq > 0.0f) inv_len = 1.0f / sqrtf(len_sq);
    const float tol_floor = 0.5f * precision_target * inv_len;
    float a = 0.0f, b = 1.0f, x = 0.5f, w = 0.5f, v = 0.5f;
    float fx = midpoint_sdf, fw = fx, fv = fx, d_step = 0.0f, e_step = 0.0f;
    if (tol_floor < 0.25f) {
        const float offset = 0.5f * golden, left = 0.5f - offset, right = 0.5f + offset;
        const float f_left = sample_hw(s, v0 + dir * left), f_right = sample_hw(s, v0 + dir * right);
        if (f_left < fx && f_left <= f_right) {
            b = 0.5f; x = left; fx = f_left; w = 0.5f; fw = midpoint_sdf; v = right; fv = f_right;
        } else if (f_right < fx) {
            a = 0.5f; x = right; fx = f_right; w = 0.5f; fw = midpoint_sdf; v = left; fv = f_left;
        } else {
            a = left; b = right; w = left; fw = f_left; v = right; fv = f_right;
        }
    }
    for (int it = 0; it < 3; ++it) {
        const float m = 0.5f * (a + b);
        const float tol = fmaxw(1.0e-2f * fabsf(x) + 1.0e-8f, tol_floor);
        const float tol2 = 2.0f * tol;
        if (fabsf(x - m) <= tol2 - 0.5f * (b - a)) break;
        bool parabolic = false;
        float trial = 0.0f;
        if (fabsf(e_step) > tol) {
            const float r = (x - w) * (fx - fv);
            float q = (x - v) * (fx - fw);
            float pnum = (x - v) * q - (x - w) * r;
            q = 2.0f * (q - r);
            if (q > 0.0f) pnum = -pnum;
            else q = -q;
            if (fabsf(pnum) < 0.5f * fabsf(q * e_step)) {
                trial = pnum / q;
                const float u_trial = x + trial;
                if (u_trial - a >= tol2 && b - u_trial >= tol2) parabolic = true;
            }
        }
        if (parabolic) {
            e_step = d_step;
            d_step = trial;
        } else {
            e_step = x >= m ? a - x : b - x;
            d_step = golden * e_step;
        }
        float u;
        if (fabsf(d_step) >= tol) u = x + d_step;
        else u = d_step > 0.0f ? x + tol : x - tol;
        const float fu = sample_hw(s, v0 + dir * u);
        if (fu <= fx) {
            if (u < x) b = x;
            else a = x;
            v = w; fv = fw; w = x; fw = fx; x = u; fx = fu;
        } else {
            if (u < x) a = u;
            else b = u;
            if (fu <= fw || w == x) {
                v = w; fv = fw; w = u; fw = fu;
            } else if (fu <= fv || v == x || v == w) {
                v = u; fv = fu;
            }
        }
    }
    best_endpoint = 0;
    float best_t = x;
    best_f = fx;
    if (a == 0.0f) {
        const float fe = sample_hw(s, v0);
        if (fe < best_f) { best_t = 0.0f; best_f = fe; best_endpoint = 1; }
    }
    if (b == 1.0f) {
        const float fe = sample_hw(s, v0 + dir * 1.0f);
        if (fe < best_f) { best_t = 1.0f; best_f = fe; best_endpoint = 2; }
    }
    best_p = v0 + dir * best_t;
}

NT_DI xform load_xform(const float* p) { return xform(vec3(p[0], p[1], p[2]), quat(p[3], p[4], p[5], p[6])); }

// Everything of one (pair, mode) that does not depend on the edge: the SDF side's grid, both transforms, the scale guards and the
// thresholds (sdf_contact.py:1186-1262).  Wave-uniform.
struct ModeCtx {
    nt_sdf s;
    int e0, ne;
    vec3 sdf_scale, inv_scale, blo, bhi;
    xform X_tri, X_sdf, X_m2s;
    float tri_margin, sdf_margin, min_scale, radius_scale, contact_threshold, thr_unscaled, inner, precision, gap_sum;
};
NT_DI bool mode_setup(const nt_mesh_sdf_args& a, int s0, int s1, int mode, ModeCtx& c) {
    const int tri_shape = mode == 0 ? s0 : s1, sdf_shape = mode == 0 ? s1 : s0;
    const int sdf_idx = a.shape_sdf_index[sdf_shape];
    c.e0 = a.shape_edge_range[2 * tri_shape];
    c.ne = a.shape_edge_range[2 * tri_shape + 1];
    if (sdf_idx < 0 || sdf_idx >= a.sdf_count || c.ne <= 0) return false;  // no SDF on that side / no edges on this side
    c.s = a.sdf_table[sdf_idx];
    if (c.s.cx <= 0) return false;
    c.gap_sum = a.shape_gap[s0] + a.shape_gap[s1];
    const float* dt = a.shape_data + 4 * tri_shape;
    const float* ds = a.shape_data + 4 * sdf_shape;
    c.sdf_scale = vec3(ds[0], ds[1], ds[2]);
    if (c.s.scale_baked) c.sdf_scale = vec3(1.0f, 1.0f, 1.0f);
    c.X_tri = load_xform(a.shape_transform + 7 * tri_shape);
    c.X_sdf = load_xform(a.shape_transform + 7 * sdf_shape);
    c.X_m2s = xform_inverse(c.X_sdf) * c.X_tri;
    c.tri_margin = dt[3];
    c.sdf_margin = ds[3];
    // safe_sdf_scale_inverse
    const float eps = 1.0e-10f;
    auto guard = [&](float v) { return fabsf(v) > eps ? v : (v >= 0.0f ? eps : -eps); };
    const float sx = guard(c.sdf_scale.x), sy = guard(c.sdf_scale.y), sz = guard(c.sdf_scale.z);
    c.inv_scale = vec3(1.0f / sx, 1.0f / sy, 1.0f / sz);
    c.min_scale = fminw(fminw(fabsf(sx), fabsf(sy)), fabsf(sz));
    c.radius_scale = fmaxw(fmaxw(fabsf(c.inv_scale.x), fabsf(c.inv_scale.y)), fabsf(c.inv_scale.z));
    c.contact_threshold = c.gap_sum + c.tri_margin + c.sdf_margin;
    c.thr_unscaled = c.contact_threshold / c.min_scale;
    c.inner = c.tri_margin + c.sdf_margin;
    c.precision = fminw(c.inner / c.min_scale, c.s.voxel_radius);  // mesh_sdf_contact_search_precision
    c.blo = vec3(c.s.box_lower[0], c.s.box_lower[1], c.s.box_lower[2]);
    c.bhi = vec3(c.s.box_upper[0], c.s.box_upper[1], c.s.box_upper[2]);
    return true;
}
// One edge of the "triangle" shape against the other shape's SDF (sdf_contact.py:1288-1480), in two steps so that the reduced
// kernel can compact the survivors of the cull before the (long) search:
//   edge_cull      bounding sphere of the edge against the SDF box, then against the value at its (clamped) midpoint
//   edge_resolve   Brent search, inner-cull consistency, corner ownership, gradient -> world point, normal shape0 -> shape1, distance
template <class CTX>
NT_DI bool edge_cull(const nt_mesh_sdf_args& a, const CTX& c, int e, float& mid) {
    const float* ec = a.edge_centers + 4 * (size_t)(c.e0 + e);
    const vec3 center = cw_mul(xform_point(c.X_m2s, vec3(ec[0], ec[1], ec[2])), c.inv_scale);
    const float threshold = ec[3] * c.radius_scale + c.thr_unscaled;
    const vec3 cl = vmin(vmax(center, c.blo), c.bhi);
    const float d2 = length_sq(center - cl);
    if (d2 > threshold * threshold) return false;
    mid = sample_hw_clamped(c.s, cl, d2 > 0.0f ? sqrtf(d2) : 0.0f);
    return mid <= threshold;
}
NT_DI bool edge_resolve(const nt_mesh_sdf_args& a, const ModeCtx& c, int e, int mode, float mid, vec3& pw, vec3& n, float& dist) {
    const nt_sdf& s = c.s;
    const float* ec = a.edge_centers + 4 * (size_t)(c.e0 + e);
    const float* eh = a.edge_halves + 4 * (size_t)(c.e0 + e);
    // the edge in the SDF's unscaled space + its corner ownership
    const vec3 c_loc = xform_point(c.X_m2s, vec3(ec[0], ec[1], ec[2]));
    const vec3 h_loc = xform_vector(c.X_m2s, vec3(eh[0], eh[1], eh[2]));
    const int ownership = (int)eh[3];
    const vec3 v0 = cw_mul(c_loc - h_loc, c.inv_scale), v1 = cw_mul(c_loc + h_loc, c.inv_scale);
    float dist_u;
    vec3 p_u;
    int endpoint;
    edge_search(s, v0, v1, mid, c.precision, dist_u, p_u, endpoint);
    const float dist_approx = dist_u * c.min_scale;
    // mesh_sdf_contact_passes_inner_cull_consistency
    bool consistent = true;
    if (dist_approx < c.inner) {
        const vec3 ic = (v0 + v1) * 0.5f;
        const float ir = length(v1 - v0) * 0.5f;
        const float cr = ir + c.inner / c.min_scale;
        const vec3 icl = vmin(vmax(ic, c.blo), c.bhi);
        if (length_sq(ic - icl) > cr * cr) consistent = false;
        else consistent = mid <= cr;
    }
    const bool owns = endpoint == 0 || ownership == 0 || (ownership & endpoint) != 0;
    if (!(dist_approx < c.contact_threshold && consistent && owns)) return false;
    vec3 dir_u = sample_grad(s, p_u);
    // scale_sdf_result_to_world (sdf_contact.py:155-182): gradient back through the anisotropic scale
    dist = dist_u * c.min_scale;                          // conservative distance through the smallest scale
    const vec3 dir = cw_mul(dir_u, c.inv_scale);          // normalised after the rigid rotation below
    const vec3 point = cw_mul(p_u, c.sdf_scale);
    pw = xform_point(c.X_sdf, point);
    vec3 dw = xform_vector(c.X_sdf, dir);
    const float dl2 = length_sq(dw);
    if (dl2 > 0.0f) dw = dw * (1.0f / sqrtf(dl2));
    else {
        vec3 fb = pw - c.X_sdf.p;
        const float fl2 = length_sq(fb);
        dw = fl2 > 0.0f ? fb * (1.0f / sqrtf(fl2)) : vec3(0.0f, 1.0f, 0.0f);
    }
    n = mode == 0 ? -dw : dw;
    return true;
}
NT_DI bool edge_contact(const nt_mesh_sdf_args& a, const ModeCtx& c, int e, int mode, vec3& pw, vec3& n, float& dist) {
    float mid;
    return edge_cull(a, c, e, mid) && edge_resolve(a, c, e, mode, mid, pw, n, dist);
}
// Can ANY edge of the "triangle" shape pass the cull?  Every edge lies inside the shape's local AABB (the reduction tables carry
// it, scale applied); its image in the SDF's unscaled space is bounded by the AABB of the eight transformed corners, and an edge
// whose bounding sphere is farther than its threshold from the SDF box is rejected by edge_cull.  Conservative (never rejects a
// mode that has a surviving edge): the slack is edge_cull's threshold for the shape's longest edge.
NT_DI bool mode_can_touch(const ModeCtx& c, const float* lo, const float* hi, float max_edge_radius) {
    vec3 mn(1e30f, 1e30f, 1e30f), mx(-1e30f, -1e30f, -1e30f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const vec3 q = cw_mul(xform_point(c.X_m2s, vec3((k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2])), c.inv_scale);
        mn = vmin(mn, q);
        mx = vmax(mx, q);
    }
    const float t = (max_edge_radius * c.radius_scale + c.thr_unscaled) * 1.0001f + 1e-6f;  // edge_cull's largest threshold
    return mn.x <= c.bhi.x + t && mx.x >= c.blo.x - t && mn.y <= c.bhi.y + t && mx.y >= c.blo.y - t && mn.z <= c.bhi.z + t &&
           mx.z >= c.blo.z - t;
}

NT_DI int live_pair_count(const nt_mesh_sdf_args& a) {
    if (a.pair_world_prefix) return a.pair_world_prefix[a.worlds];
    if (!a.pair_count_device) return a.pair_count;
    const int n = *a.pair_count_device;
    return n < a.pair_count ? n : a.pair_count;
}
// world-region pairs: flat live index f -> position w * pairs_per_world + k in `pairs` (identity for a plain list)
NT_DI int pair_slot(const nt_mesh_sdf_args& a, int f) {
    if (!a.pair_world_prefix) return f;
    int lo = 0, hi = a.worlds;  // the last world whose prefix is <= f
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.pair_world_prefix[mid] <= f) lo = mid;
        else hi = mid;
    }
    return lo * a.pairs_per_world + (f - a.pair_world_prefix[lo]);
}

__global__ void __launch_bounds__(256) mesh_sdf_collide_kernel(nt_mesh_sdf_args a) {
    const int pair_count = live_pair_count(a);
    for (int pair_idx = blockIdx.x; pair_idx < pair_count; pair_idx += gridDim.x) {
        const int s0 = a.pairs[2 * pair_idx], s1 = a.pairs[2 * pair_idx + 1];
        for (int mode = 0; mode < 2; ++mode) {
            ModeCtx c;
            if (!mode_setup(a, s0, s1, mode, c)) continue;
            for (int e = threadIdx.x; e < c.ne; e += blockDim.x) {
                vec3 pw, n;
                float dist;
                if (!edge_contact(a, c, e, mode, pw, n, dist)) continue;
                const int slot = atomicAdd(a.out_count, 1);
                if (slot < a.capacity) {
                    a.out_pair[slot] = pair_idx;
                    a.out_key[slot] = (e << 2) | (mode << 1);
                    float* o = a.out_data + 9 * (size_t)slot;
                    o[0] = pw.x; o[1] = pw.y; o[2] = pw.z;
                    o[3] = n.x; o[4] = n.y; o[5] = n.z;
                    o[6] = dist;
                    o[7] = a.shape_data[4 * s0 + 3];
                    o[8] = a.shape_data[4 * s1 + 3];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Global contact reduction (contact_reduction_global.py) in LDS.
//
// The reference buffers every mesh-SDF contact in global memory and races packed (rank, fingerprint, contact id) values into a
// device-wide hashtable with 64-bit atomic maxima: per (shape pair, normal bin) six spatial-extreme slots + one deepest-contact
// slot, per (shape pair, voxel group) seven deepest-contact slots; contacts inside the inner depth outrank outer ones in the
// spatial slots (:1519-1752, deterministic packing :447-558).  An export pass walks the active entries, drops roundoff twins
// within an entry and hands every surviving contact to the writer once (:2098-2290).
// Here one workgroup owns a shape pair, so the pair's whole table -- (20 + 15) entries x 7 values = 245 x 8 B -- lives in LDS
// and the maxima are ds_max_u64.  The packed value carries the fingerprint (edge, mode), which is unique inside a pair: no
// contact buffer and no contact ids are needed, the <= 245 winners are recomputed from their fingerprint after the barrier
// (same instructions, same bits), compared for roundoff twins, de-duplicated, ranked by fingerprint and written as one
// contiguous block per pair -- already in the order `deterministic=True` sorts contacts into.  The outcome of the reference
// does not depend on the arrival order either (every slot ends at the maximum); hashtable overflow and buffer exhaustion, the
// two ways the reference can lose contacts, do not exist here.
// ------------------------------------------------------------------------------------------------
#include "nt_contact_reduce.hpp"  // constants, packed values, bins / voxels / octahedral codes, red_offer, RedLds, red_finish

// mesh_sdf_collision_global_reduce_kernel (sdf_contact.py:1534-1990) + export_reduced_contacts_kernel, one workgroup per pair.
// Per mode: (1) every lane culls edges and the survivors are compacted into an LDS list (the reference's cooperative tile
// stack), (2) the lanes walk that list densely -- Brent search, consistency, ownership, gradient -- keep each contact's record
// in LDS and offer it to the pair's table.  After both modes the <= 245 winners read their record back through the fingerprint
// (a list overflow -- meshes with hundreds of near edges -- falls back to recomputing the winner: same instructions, same bits).
constexpr int HIT_CAP = 128;
constexpr int SLOT_LDS = 512;
struct HitLds {
    int fp[HIT_CAP];       // fingerprint of the survivor (edge << 2 | mode << 1); -1 once the resolve rejected it
    float mid[HIT_CAP];    // SDF value at its clamped midpoint (edge_cull)
    float pos[HIT_CAP][4]; // resolved contact: world point, distance
    float oct[HIT_CAP][2]; // octahedral code of its normal
    int n;                 // survivors appended so far (may exceed HIT_CAP: the excess was processed on the spot)
};
#ifndef NT_SDF_WAVES_PER_EU  // measurement builds (tools/build_variant.py -DNT_SDF_WAVES_PER_EU=n) cap the registers for n waves per SIMD
#define NT_SDF_OCCUPANCY
#else
#define NT_SDF_OCCUPANCY __attribute__((amdgpu_waves_per_eu(NT_SDF_WAVES_PER_EU, NT_SDF_WAVES_PER_EU)))
#endif
__global__ void __launch_bounds__(256) NT_SDF_OCCUPANCY mesh_sdf_collide_reduced_kernel(nt_mesh_sdf_args a, nt_contact_reduce_shapes r) {
    __shared__ RedLds L;
    __shared__ HitLds H;
    __shared__ uint32_t slot_lds[SLOT_LDS];  // the SDF's indirection table of the current mode (every sample reads it first)
    const int t = threadIdx.x;
    const int pair_count = live_pair_count(a);
    for (int f = blockIdx.x; f < pair_count; f += gridDim.x) {
        const int pair_idx = pair_slot(a, f);
        if (a.pair_kind && a.pair_kind[pair_idx] != 0) continue;  // another leg's pair (uniform)
        const int s0 = a.pairs[2 * (size_t)pair_idx], s1 = a.pairs[2 * (size_t)pair_idx + 1];
        for (int k = t; k < RED_SLOTS; k += blockDim.x) { L.tbl[k] = 0ull; L.fp[k] = -1; L.keep[k] = 0; }
        if (t == 0) H.n = 0;
        __syncthreads();
        for (int mode = 0; mode < 2; ++mode) {
            ModeCtx c;
            if (!mode_setup(a, s0, s1, mode, c)) continue;
            const int tri_shape = mode == 0 ? s0 : s1;
            if (r.shape_edge_radius_max && !mode_can_touch(c, r.shape_aabb_lower + 3 * tri_shape, r.shape_aabb_upper + 3 * tri_shape,
                                                           r.shape_edge_radius_max[tri_shape]))
                continue;
            const vec3 midpoint = (c.X_tri.p + c.X_sdf.p) * 0.5f;
            const float margin_sum = c.tri_margin + c.sdf_margin;
            const float inner_depth = margin_sum + fminw(c.s.voxel_radius * c.min_scale, c.gap_sum);  // base gap == gap
            const float outer_depth = margin_sum + c.gap_sum;
            auto offer = [&](int fp, vec3 pw, vec3 n, float dist) {
                const vec3 local = quat_rotate_inv(c.X_tri.q, pw - c.X_tri.p);
                red_offer(L.tbl, n, pw - midpoint, dist, inner_depth, outer_depth, local, r.shape_aabb_lower + 3 * tri_shape,
                          r.shape_aabb_upper + 3 * tri_shape, r.shape_voxel_res + 3 * tri_shape, fp);
            };
            const int seg = H.n < HIT_CAP ? H.n : HIT_CAP;  // this mode's survivors start here (uniform: read after a barrier)
            // A sample is two dependent reads (indirection slot, then texels); with the small slot table in LDS only the texel
            // fetch pays a trip to L2.  (The winner pass below recomputes through the global table: same values.)
            const int n_slots = c.s.cx * c.s.cy * c.s.cz;
            if (n_slots <= SLOT_LDS) {
                for (int k = t; k < n_slots; k += blockDim.x) slot_lds[k] = c.s.slots[k];
                c.s.slots = slot_lds;
            }
            __syncthreads();
            for (int e = t; e < c.ne; e += blockDim.x) {
                float mid;
                if (!edge_cull(a, c, e, mid)) continue;
                const int fp = (e << 2) | (mode << 1);
                const int i = atomicAdd(&H.n, 1);
                if (i < HIT_CAP) {
                    H.fp[i] = fp;
                    H.mid[i] = mid;
                } else {  // list full: resolve on the spot, the winner pass recomputes it if it wins
                    vec3 pw, n;
                    float dist;
                    if (edge_resolve(a, c, e, mode, mid, pw, n, dist)) offer(fp, pw, n, dist);
                }
            }
            __syncthreads();
            const int end = H.n < HIT_CAP ? H.n : HIT_CAP;
            for (int i = seg + t; i < end; i += blockDim.x) {
                const int fp = H.fp[i];
                vec3 pw, n;
                float dist;
                if (edge_resolve(a, c, fp >> 2, mode, H.mid[i], pw, n, dist)) {
                    H.pos[i][0] = pw.x; H.pos[i][1] = pw.y; H.pos[i][2] = pw.z; H.pos[i][3] = dist;
                    red_encode_oct(n, H.oct[i][0], H.oct[i][1]);
                    offer(fp, pw, n, dist);
                } else {
                    H.fp[i] = -1;
                }
            }
            __syncthreads();
        }
        const int hits = H.n < HIT_CAP ? H.n : HIT_CAP;
        if (H.n == 0) {  // no edge survived the cull (most candidate pairs of a pile): no rows, nothing to reduce (uniform)
            if (t == 0 && a.out_blk) { a.out_blk[2 * (size_t)pair_idx] = 0; a.out_blk[2 * (size_t)pair_idx + 1] = 0; }
            __syncthreads();
            continue;
        }
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {  // the winner of slot k: its record from the list, else recomputed
            if (L.tbl[k] == 0ull) continue;
            const int fp = (int)(L.tbl[k] & RED_FP_MASK);
            int i = 0;
            while (i < hits && H.fp[i] != fp) ++i;
            if (i < hits) {
                L.pos[k][0] = H.pos[i][0]; L.pos[k][1] = H.pos[i][1]; L.pos[k][2] = H.pos[i][2]; L.pos[k][3] = H.pos[i][3];
                L.oct[k][0] = H.oct[i][0]; L.oct[k][1] = H.oct[i][1];
            } else {
                const int mode = (fp >> 1) & 1;
                ModeCtx c;
                mode_setup(a, s0, s1, mode, c);
                vec3 pw, n;
                float dist;
                edge_contact(a, c, fp >> 2, mode, pw, n, dist);
                L.pos[k][0] = pw.x; L.pos[k][1] = pw.y; L.pos[k][2] = pw.z; L.pos[k][3] = dist;
                red_encode_oct(n, L.oct[k][0], L.oct[k][1]);
            }
            L.fp[k] = fp;
        }
        __syncthreads();
        red_finish(L, RedLdsRec{L});
        if (t == 0) {
            L.base = L.total > 0 ? atomicAdd(a.out_count, L.total) : 0;
            if (a.out_blk) {  // the pair's block: rows past the capacity do not exist for the consumers
                const int room = a.capacity - L.base;
                a.out_blk[2 * (size_t)pair_idx] = L.base;
                a.out_blk[2 * (size_t)pair_idx + 1] = L.total < room ? L.total : (room > 0 ? room : 0);
            }
        }
        __syncthreads();
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {
            const int slot = L.base + L.keep[k];
            if (L.keep[k] < 0 || slot >= a.capacity) continue;
            const vec3 n = red_decode_oct(L.oct[k][0], L.oct[k][1]);
            a.out_pair[slot] = pair_idx;
            a.out_key[slot] = L.fp[k];
            float* o = a.out_data + 9 * (size_t)slot;
            o[0] = L.pos[k][0]; o[1] = L.pos[k][1]; o[2] = L.pos[k][2];
            o[3] = n.x; o[4] = n.y; o[5] = n.z;
            o[6] = L.pos[k][3];
            o[7] = a.shape_data[4 * s0 + 3];
            o[8] = a.shape_data[4 * s1 + 3];
        }
        __syncthreads();
    }
}

// ---- the same contacts in dense stages (nt_mesh_sdf_args.hit_*): a pile's candidate pairs mostly have no surviving edge and the
// survivors of one pair fill a few lanes of a wave, so the one-workgroup-per-pair kernel above spends its time in latency chains
// (pair lookup -> shapes -> transforms -> SDF descriptor -> edge -> indirection slot -> texels) on nearly empty waves, and every
// pair with rows pays an atomic on ONE counter (measured: the serialised same-address atomics alone cost more than the arithmetic).
// Here every stage is flat over its own population and no counter is shared by more than a few hundred waves:
//   sdf_units_kernel    one LANE per live pair: everything that does not depend on the edge (mode_setup, mode_can_touch of both
//                       modes), runnable pairs compacted into a context list (one atomic per wave); clears the pair's block entry
//   sdf_cull_kernel     one wave per runnable pair: contexts by broadcast reads, edge culling of both modes, survivors compacted
//                       by ballot into the wave's STRIPE of the survivor list (one contiguous block per pair, mode 0 first;
//                       `hit_stripes` counters, one per 64 bytes)
//   sdf_resolve_kernel  one lane per survivor: Brent search + gradient -> world point, normal, distance (or rejected)
//   sdf_reduce_kernel   one workgroup per runnable pair that has survivors: table, winners (records read from the list instead of
//                       being recomputed), twins / duplicates / ranks; with out_blk the rows go to the START OF THE PAIR'S
//                       SURVIVOR BLOCK (a pair never has more rows than survivors): no counter at all
// Same arithmetic per edge and an order-independent table (atomicMax on keys that carry the fingerprint): the rows are those of
// the single kernel bit for bit, whatever order the blocks land in the lists.
constexpr int UNIT_WORDS = 24;  // CullCtx in HBM: X_m2s[7] inv_scale[3] blo[3] bhi[3] thr_unscaled radius_scale | e0 ne sdf_idx pair | inner outer
constexpr int STRIPE_PAD = 16;  // ints between stripe counters (64 bytes: one counter per cache line / atomic unit)
struct CullCtx {
    nt_sdf s;
    int e0, ne;
    vec3 inv_scale, blo, bhi;
    xform X_m2s;
    float thr_unscaled, radius_scale;
};
enum { CNT_UNITS = 1 };  // nt_mesh_sdf_args.hit_count[4]: [0] survivors beyond a stripe's room (dropped), [1] runnable pairs, [2] stripes in use
__global__ void __launch_bounds__(256) sdf_units_kernel(nt_mesh_sdf_args a, nt_contact_reduce_shapes r) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const long long cap = a.pair_world_prefix ? (long long)a.worlds * a.pairs_per_world : (long long)a.pair_count;
    const int plain_live = a.pair_world_prefix ? 0 : live_pair_count(a);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.hit_count[2] = a.hit_stripe_count;  // the stripes this call uses
    // every wave walks whole rounds (the ballot below needs all 64 lanes)
    for (long long base = ((long long)blockIdx.x * blockDim.x + threadIdx.x - lane); base < cap; base += (long long)gridDim.x * blockDim.x) {
        const long long u = base + lane;
        bool live = u < cap;
        const int pair_idx = live ? (int)u : 0;
        if (live) {
            if (a.pair_world_prefix) {
                const int w = pair_idx / a.pairs_per_world, k = pair_idx - w * a.pairs_per_world;
                live = k < a.pair_world_prefix[w + 1] - a.pair_world_prefix[w];
            } else {
                live = pair_idx < plain_live;
            }
        }
        live = live && !(a.pair_kind && a.pair_kind[pair_idx] != 0);  // another leg's pair
        bool run[2] = {false, false};
        ModeCtx c[2];
        if (live) {
            int* blk = a.hit_blk + 4 * (size_t)pair_idx;
            blk[0] = 0; blk[1] = 0; blk[2] = 0; blk[3] = 0;
            if (a.out_blk) { a.out_blk[2 * (size_t)pair_idx] = 0; a.out_blk[2 * (size_t)pair_idx + 1] = 0; }
            const int s0 = a.pairs[2 * (size_t)pair_idx], s1 = a.pairs[2 * (size_t)pair_idx + 1];
#pragma unroll
            for (int mode = 0; mode < 2; ++mode) {
                run[mode] = mode_setup(a, s0, s1, mode, c[mode]);
                if (run[mode] && r.shape_edge_radius_max) {
                    const int tri_shape = mode == 0 ? s0 : s1;
                    run[mode] = mode_can_touch(c[mode], r.shape_aabb_lower + 3 * tri_shape, r.shape_aabb_upper + 3 * tri_shape,
                                               r.shape_edge_radius_max[tri_shape]);
                }
            }
        }
        const bool any = run[0] || run[1];
        const unsigned long long m = __ballot(any);
        int at = 0;
        if (lane == 0 && m) at = atomicAdd(a.hit_count + CNT_UNITS, __popcll(m));
        at = __shfl(at, 0) + __popcll(m & below);
        if (any) {
#pragma unroll
            for (int mode = 0; mode < 2; ++mode) {
                const ModeCtx& k = c[mode];
                float* o = a.unit_ctx + UNIT_WORDS * (2 * (size_t)at + mode);
                int* oi = reinterpret_cast<int*>(o);
                if (!run[mode]) { oi[19] = 0; oi[21] = pair_idx; continue; }  // no edges: the mode is skipped
                o[0] = k.X_m2s.p.x; o[1] = k.X_m2s.p.y; o[2] = k.X_m2s.p.z;
                o[3] = k.X_m2s.q.x; o[4] = k.X_m2s.q.y; o[5] = k.X_m2s.q.z; o[6] = k.X_m2s.q.w;
                o[7] = k.inv_scale.x; o[8] = k.inv_scale.y; o[9] = k.inv_scale.z;
                o[10] = k.blo.x; o[11] = k.blo.y; o[12] = k.blo.z;
                o[13] = k.bhi.x; o[14] = k.bhi.y; o[15] = k.bhi.z;
                o[16] = k.thr_unscaled; o[17] = k.radius_scale;
                const float margin_sum = k.tri_margin + k.sdf_margin;
                o[22] = margin_sum + fminw(k.s.voxel_radius * k.min_scale, k.gap_sum);  // the reduction's inner depth (base gap == gap)
                o[23] = margin_sum + k.gap_sum;                                         // ... and outer depth
                oi[18] = k.e0; oi[19] = k.ne;
                oi[20] = a.shape_sdf_index[mode == 0 ? a.pairs[2 * (size_t)pair_idx + 1] : a.pairs[2 * (size_t)pair_idx]];
                oi[21] = pair_idx;
            }
        }
    }
}
NT_DI void cull_ctx_load(const nt_mesh_sdf_args& a, const float* o, CullCtx& c) {
    const int* oi = reinterpret_cast<const int*>(o);
    c.ne = oi[19];
    if (c.ne == 0) return;
    c.X_m2s = xform(vec3(o[0], o[1], o[2]), quat(o[3], o[4], o[5], o[6]));
    c.inv_scale = vec3(o[7], o[8], o[9]);
    c.blo = vec3(o[10], o[11], o[12]);
    c.bhi = vec3(o[13], o[14], o[15]);
    c.thr_unscaled = o[16];
    c.radius_scale = o[17];
    c.e0 = oi[18];
    c.s = a.sdf_table[__builtin_amdgcn_readfirstlane(oi[20])];
}
#ifdef NT_SDF_CULL_WAVES  // measurement builds: cap the registers for n waves per SIMD (measured: 5 / 6 / 8 waves spill and lose 4 / 12 / 23 %)
#define NT_SDF_CULL_OCC __attribute__((amdgpu_waves_per_eu(NT_SDF_CULL_WAVES, NT_SDF_CULL_WAVES)))
#else
#define NT_SDF_CULL_OCC
#endif
#ifndef NT_SDF_RESOLVE_WAVES
#define NT_SDF_RESOLVE_WAVES 4  // measured (MI355X, C5): 136 VGPR / 3 waves per SIMD 360 k env-steps/s, capped at 128 / 4 waves 381 k
#endif
#define NT_SDF_RESOLVE_OCC __attribute__((amdgpu_waves_per_eu(NT_SDF_RESOLVE_WAVES, NT_SDF_RESOLVE_WAVES)))
__global__ void __launch_bounds__(256) NT_SDF_CULL_OCC sdf_cull_kernel(nt_mesh_sdf_args a) {
    const int lane = threadIdx.x & 63, waves = blockDim.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int units = a.hit_count[CNT_UNITS];
    const int wave_id = blockIdx.x * waves + (threadIdx.x >> 6);
    const int stripe = wave_id % a.hit_stripe_count;
    const int slice = a.hit_capacity / a.hit_stripe_count;
    for (int uv = wave_id; uv < units; uv += gridDim.x * waves) {
        // the unit is the wave's: telling the compiler so turns the context / SDF descriptor loads into scalar loads (SGPRs
        // instead of ~90 VGPRs of broadcast data: more waves per SIMD to hide the sample latency behind)
        const int u = __builtin_amdgcn_readfirstlane(uv);
        const float* o = a.unit_ctx + UNIT_WORDS * 2 * (size_t)u;
        const int pair_idx = reinterpret_cast<const int*>(o)[21];
        bool hit0[2] = {false, false};
        float mid0[2] = {0.0f, 0.0f};
        int total[2] = {0, 0};
#pragma unroll
        for (int mode = 0; mode < 2; ++mode) {  // pass 1: how many survive (the pair's block is reserved in one piece)
            CullCtx c;
            cull_ctx_load(a, o + UNIT_WORDS * mode, c);
            const int iters = (c.ne + 63) >> 6;
            for (int it = 0; it < iters; ++it) {
                const int e = it * 64 + lane;
                float mid = 0.0f;
                const bool hit = e < c.ne && edge_cull(a, c, e, mid);
                if (it == 0) { hit0[mode] = hit; mid0[mode] = mid; }
                total[mode] += __popcll(__ballot(hit));
            }
        }
        const int sum = total[0] + total[1];
        if (sum == 0) continue;  // wave-uniform; the pair's block entry is already zero
        int base = 0;
        if (lane == 0) {
            const int at = atomicAdd(a.hit_stripes + STRIPE_PAD * stripe, sum);
            int room = slice - at;
            room = room < 0 ? 0 : room;
            if (room < sum) atomicAdd(a.hit_count, sum - room);  // dropped survivors: reported, the caller sizes the list up
            base = stripe * slice + at;
            int* blk = a.hit_blk + 4 * (size_t)pair_idx;
            const int n0 = total[0] < room ? total[0] : room, n1 = total[1] < room - n0 ? total[1] : room - n0;
            blk[0] = base; blk[1] = n0; blk[2] = base + n0; blk[3] = n1;
        }
        base = __shfl(base, 0);
        const int end = (stripe + 1) * slice;
        int at = base;
#pragma unroll
        for (int mode = 0; mode < 2; ++mode) {  // pass 2: write (meshes with more than 64 edges cull their later edges again)
            if (total[mode] == 0) continue;
            CullCtx c;
            cull_ctx_load(a, o + UNIT_WORDS * mode, c);
            const int iters = (c.ne + 63) >> 6;
            for (int it = 0; it < iters; ++it) {
                const int e = it * 64 + lane;
                float mid = mid0[mode];
                const bool hit = it == 0 ? hit0[mode] : (e < c.ne && edge_cull(a, c, e, mid));
                const unsigned long long m = __ballot(hit);
                const int i = at + __popcll(m & below);
                if (hit && i < end) {
                    a.hit_pair[i] = pair_idx;
                    a.hit_fp[i] = (e << 2) | (mode << 1);
                    a.hit_rec[8 * (size_t)i] = mid;
                }
                at += __popcll(m);
            }
        }
    }
}
__global__ void __launch_bounds__(256) NT_SDF_RESOLVE_OCC sdf_resolve_kernel(nt_mesh_sdf_args a) {  // grid = stripes x blocks per stripe
    const int stripe = blockIdx.x % a.hit_stripe_count, xb = blockIdx.x / a.hit_stripe_count, nxb = gridDim.x / a.hit_stripe_count;
    const int slice = a.hit_capacity / a.hit_stripe_count;
    int n = a.hit_stripes[STRIPE_PAD * stripe];
    n = n < slice ? n : slice;
    for (int j = xb * blockDim.x + threadIdx.x; j < n; j += nxb * blockDim.x) {
        const int i = stripe * slice + j;
        const int pair_idx = a.hit_pair[i], fp = a.hit_fp[i], mode = (fp >> 1) & 1;
        ModeCtx c;
        mode_setup(a, a.pairs[2 * (size_t)pair_idx], a.pairs[2 * (size_t)pair_idx + 1], mode, c);
        float* rec = a.hit_rec + 8 * (size_t)i;
        vec3 pw, nrm;
        float dist;
        if (edge_resolve(a, c, fp >> 2, mode, rec[0], pw, nrm, dist)) {
            rec[0] = pw.x; rec[1] = pw.y; rec[2] = pw.z; rec[3] = dist;
            rec[4] = nrm.x; rec[5] = nrm.y; rec[6] = nrm.z;
        } else {
            a.hit_fp[i] = -1;
        }
    }
}
constexpr int RED_FP_CACHE = 128;  // fingerprints of a pair's survivors kept in LDS for the winners' lookup
struct RedLdsIdx {  // the pair's table with ONE INDEX per slot instead of a record: 22 B per slot, 5.9 KB with the cache below
    unsigned long long tbl[RED_SLOTS];
    int fp[RED_SLOTS];     // fingerprint of the slot's winner, -1 = empty
    int hid[RED_SLOTS];    // ... and where its record sits in the survivor list
    short src[RED_SLOTS];  // red_finish: kept slots, compacted
    short keep[RED_SLOTS]; // survives the roundoff-twin pass, then its rank among the pair's rows (-1: none)
    short first[RED_SLOTS];
    int base, total;
};
struct RedIdxRec {  // records through the index: world point, distance from the list, the normal's octahedral code recomputed
    const RedLdsIdx& L;
    const float* hit_rec;
    NT_DI void operator()(int k, float* o) const {
        const float* rec = hit_rec + 8 * (size_t)L.hid[k];
        o[0] = rec[0]; o[1] = rec[1]; o[2] = rec[2]; o[3] = rec[3];
        red_encode_oct(vec3(rec[4], rec[5], rec[6]), o[4], o[5]);
    }
};
__global__ void __launch_bounds__(256) sdf_reduce_kernel(nt_mesh_sdf_args a, nt_contact_reduce_shapes r) {
    __shared__ RedLdsIdx L;
    __shared__ int hfp[RED_FP_CACHE];
    __shared__ int any;
    const int t = threadIdx.x;
    const int units = a.hit_count[CNT_UNITS];
    for (int f = blockIdx.x; f < units; f += gridDim.x) {
        const int pair_idx = reinterpret_cast<const int*>(a.unit_ctx + UNIT_WORDS * 2 * (size_t)f)[21];
        const int* blk = a.hit_blk + 4 * (size_t)pair_idx;
        const int off0 = blk[0], cnt0 = blk[1], off1 = blk[2], cnt1 = blk[3];
        if (cnt0 + cnt1 == 0) continue;  // no edge survived the cull: no rows (uniform; out_blk is already zero)
        const int s0 = a.pairs[2 * (size_t)pair_idx], s1 = a.pairs[2 * (size_t)pair_idx + 1];
        for (int k = t; k < RED_SLOTS; k += blockDim.x) { L.tbl[k] = 0ull; L.fp[k] = -1; L.keep[k] = 0; }
        if (t == 0) any = 0;
        __syncthreads();
        const xform X0 = load_xform(a.shape_transform + 7 * s0), X1 = load_xform(a.shape_transform + 7 * s1);
        const vec3 midpoint = (X0.p + X1.p) * 0.5f;  // (X_tri.p + X_sdf.p) / 2 of either mode
        for (int mode = 0; mode < 2; ++mode) {
            const int off = mode == 0 ? off0 : off1, cnt = mode == 0 ? cnt0 : cnt1;
            if (cnt == 0) continue;
            const float* uc = a.unit_ctx + UNIT_WORDS * (2 * (size_t)f + mode);
            const float inner_depth = uc[22], outer_depth = uc[23];
            const int tri_shape = mode == 0 ? s0 : s1;
            const xform X_tri = mode == 0 ? X0 : X1;
            for (int j = t; j < cnt; j += blockDim.x) {
                const int i = off + j, fp = a.hit_fp[i];
                const int h = (mode == 0 ? 0 : cnt0) + j;  // position among the pair's survivors
                if (h < RED_FP_CACHE) hfp[h] = fp;
                if (fp < 0) continue;
                const float* rec = a.hit_rec + 8 * (size_t)i;
                const vec3 pw(rec[0], rec[1], rec[2]), nrm(rec[4], rec[5], rec[6]);
                const vec3 local = quat_rotate_inv(X_tri.q, pw - X_tri.p);
                if (rec[3] < outer_depth) any = 1;  // benign race: every writer stores 1
                red_offer(L.tbl, nrm, pw - midpoint, rec[3], inner_depth, outer_depth, local, r.shape_aabb_lower + 3 * tri_shape,
                          r.shape_aabb_upper + 3 * tri_shape, r.shape_voxel_res + 3 * tri_shape, fp);
            }
        }
        __syncthreads();
        if (!any) {  // every survivor was rejected by the search or lies beyond the outer depth: no rows (uniform; out_blk is zero)
            __syncthreads();
            continue;
        }
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {  // the winner of slot k: where its record sits in the pair's blocks
            if (L.tbl[k] == 0ull) continue;
            const int fp = (int)(L.tbl[k] & RED_FP_MASK);
            const int total = cnt0 + cnt1, cached = total < RED_FP_CACHE ? total : RED_FP_CACHE;
            int h = 0;
            while (h < cached && hfp[h] != fp) ++h;
            if (h == cached)  // beyond the cache: the table only holds fingerprints offered from these blocks, so it is there
                while (a.hit_fp[h < cnt0 ? off0 + h : off1 + (h - cnt0)] != fp) ++h;
            L.hid[k] = h < cnt0 ? off0 + h : off1 + (h - cnt0);
            L.fp[k] = fp;
        }
        __syncthreads();
        red_finish(L, RedIdxRec{L, a.hit_rec});
        if (t == 0) {
            if (a.out_blk) {  // the pair's rows start where its survivors do (rows <= survivors): no counter
                L.base = off0;
                a.out_blk[2 * (size_t)pair_idx] = off0;
                a.out_blk[2 * (size_t)pair_idx + 1] = L.total;
            } else {
                L.base = L.total > 0 ? atomicAdd(a.out_count, L.total) : 0;
            }
        }
        __syncthreads();
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {
            const int slot = L.base + L.keep[k];
            if (L.keep[k] < 0 || slot >= a.capacity) continue;
            const float* rec = a.hit_rec + 8 * (size_t)L.hid[k];
            float ox, oy;
            red_encode_oct(vec3(rec[4], rec[5], rec[6]), ox, oy);
            const vec3 n = red_decode_oct(ox, oy);
            a.out_pair[slot] = pair_idx;
            a.out_key[slot] = L.fp[k];
            float* o = a.out_data + 9 * (size_t)slot;
            o[0] = rec[0]; o[1] = rec[1]; o[2] = rec[2];
            o[3] = n.x; o[4] = n.y; o[5] = n.z;
            o[6] = rec[3];
            o[7] = a.shape_data[4 * s0 + 3];
            o[8] = a.shape_data[4 * s1 + 3];
        }
        __syncthreads();
    }
}

// nt_contact_reduce_shapes.keep_all: the last stage of reduce_contacts=False (mesh_sdf_collision_kernel's contact set,
// sdf_contact.py:1098-1515: every edge the search admits) -- one wave per runnable pair walks the pair's survivor block and writes
// the admitted records as rows from the START of the block, ascending fingerprint (the two modes' lists are each ascending in the
// edge, so a row's place is its rank in its own list plus the admitted entries of the other list below its fingerprint).  The
// normal is the search's own (the octahedral round trip belongs to the reducer's export).
__global__ void __launch_bounds__(64) sdf_keep_all_kernel(nt_mesh_sdf_args a) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int units = a.hit_count[CNT_UNITS];
    for (int f = blockIdx.x; f < units; f += gridDim.x) {
        const int pair_idx = reinterpret_cast<const int*>(a.unit_ctx + UNIT_WORDS * 2 * (size_t)f)[21];
        const int* blk = a.hit_blk + 4 * (size_t)pair_idx;
        const int off0 = blk[0], cnt0 = blk[1], off1 = blk[2], cnt1 = blk[3];
        if (cnt0 + cnt1 == 0) continue;  // uniform; out_blk is already zero
        const int s0 = a.pairs[2 * (size_t)pair_idx], s1 = a.pairs[2 * (size_t)pair_idx + 1];
        const float m0 = a.shape_data[4 * s0 + 3], m1 = a.shape_data[4 * s1 + 3];
        int total = 0;
        for (int mode = 0; mode < 2; ++mode) {
            const int off = mode == 0 ? off0 : off1, cnt = mode == 0 ? cnt0 : cnt1;
            const int ooff = mode == 0 ? off1 : off0, ocnt = mode == 0 ? cnt1 : cnt0;
            int run = 0;
            for (int j0 = 0; j0 < cnt; j0 += 64) {
                const int j = j0 + lane;
                const int fp = j < cnt ? a.hit_fp[off + j] : -1;
                const unsigned long long m = __ballot(fp >= 0);
                if (fp >= 0) {
                    int less = 0;
                    for (int k = 0; k < ocnt; ++k) {
                        const int fo = a.hit_fp[ooff + k];
                        less += (fo >= 0 && fo < fp) ? 1 : 0;
                    }
                    const int slot = off0 + run + __popcll(m & below) + less;
                    if (slot < a.capacity) {
                        const float* rec = a.hit_rec + 8 * (size_t)(off + j);
                        a.out_pair[slot] = pair_idx;
                        a.out_key[slot] = fp;
                        float* o = a.out_data + 9 * (size_t)slot;
                        o[0] = rec[0]; o[1] = rec[1]; o[2] = rec[2];
                        o[3] = rec[4]; o[4] = rec[5]; o[5] = rec[6];
                        o[6] = rec[3];
                        o[7] = m0;
                        o[8] = m1;
                    }
                }
                run += __popcll(m);
            }
            total += run;
        }
        if (lane == 0) {
            a.out_blk[2 * (size_t)pair_idx] = off0;
            a.out_blk[2 * (size_t)pair_idx + 1] = total;
        }
    }
}

// The same reduction over a caller-supplied unreduced list grouped by shape pair (segment_start): the stage on its own, so that
// it can be held against the record of the reference's reducer on arbitrary contact sets.
__global__ void __launch_bounds__(256) contacts_reduce_list_kernel(nt_contact_reduce_list a) {
    __shared__ RedLds L;
    const int t = threadIdx.x;
    for (int seg = blockIdx.x; seg < a.segments; seg += gridDim.x) {
        const int i0 = a.segment_start[seg], i1 = a.segment_start[seg + 1];
        for (int k = t; k < RED_SLOTS; k += blockDim.x) { L.tbl[k] = 0ull; L.fp[k] = -1; L.keep[k] = 0; L.src[k] = -1; }
        __syncthreads();
        for (int i = i0 + t; i < i1; i += blockDim.x)
            red_offer(L.tbl, vec3(a.normal[3 * i], a.normal[3 * i + 1], a.normal[3 * i + 2]),
                      vec3(a.centered[3 * i], a.centered[3 * i + 1], a.centered[3 * i + 2]), a.depth[i], a.inner[i], a.outer[i],
                      vec3(a.local[3 * i], a.local[3 * i + 1], a.local[3 * i + 2]), a.aabb_lo + 3 * i, a.aabb_hi + 3 * i,
                      a.res + 3 * i, a.fp[i]);
        __syncthreads();
        for (int i = i0 + t; i < i1; i += blockDim.x) {  // the winners find their slots through the fingerprint
            if (!(a.depth[i] < a.outer[i])) continue;
            const unsigned long long fp = (unsigned long long)a.fp[i] & RED_FP_MASK;
            for (int k = 0; k < RED_SLOTS; ++k)
                if (L.tbl[k] != 0ull && (L.tbl[k] & RED_FP_MASK) == fp) L.src[k] = i;
        }
        __syncthreads();
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {
            const int i = L.src[k];
            if (i < 0) continue;
            L.pos[k][0] = a.pos[3 * i]; L.pos[k][1] = a.pos[3 * i + 1]; L.pos[k][2] = a.pos[3 * i + 2]; L.pos[k][3] = a.depth[i];
            red_encode_oct(vec3(a.normal[3 * i], a.normal[3 * i + 1], a.normal[3 * i + 2]), L.oct[k][0], L.oct[k][1]);
            L.fp[k] = a.fp[i];
        }
        __syncthreads();
        for (int k = t; k < RED_SLOTS; k += blockDim.x) L.srcidx[k] = L.src[k];  // red_finish compacts into L.src
        __syncthreads();
        red_finish(L, RedLdsRec{L});
        if (t == 0) L.base = L.total > 0 ? atomicAdd(a.out_count, L.total) : 0;
        __syncthreads();
        for (int k = t; k < RED_SLOTS; k += blockDim.x) {
            const int slot = L.base + L.keep[k];
            if (L.keep[k] < 0 || slot >= a.capacity) continue;
            const vec3 n = red_decode_oct(L.oct[k][0], L.oct[k][1]);
            a.out_index[slot] = L.srcidx[k];
            a.out_normal[3 * slot] = n.x; a.out_normal[3 * slot + 1] = n.y; a.out_normal[3 * slot + 2] = n.z;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Hydroelastic contacts between two SDF shapes (sdf_hydroelastic.py): the contact surface is the iso-surface p_a == p_b of the
// two pressure fields p = -kh * signed_depth, extracted with marching cubes over the finer SDF's voxels; every face becomes a
// contact (centre, normal a -> b, margin-relative separation) with stiffness area * pressure / |separation|
// (decode_contacts_kernel :1823-1928).  Unreduced path.  One workgroup per pair; its lanes take the voxels of shape B's fine
// grid that lie inside shape A's SDF box (the reference reaches the same voxels through a block broad phase + octree, :1026-1170).
// ------------------------------------------------------------------------------------------------
NT_DI float sample_at_voxel(const nt_sdf& s, int ix, int iy, int iz) {  // texture_sample_sdf_at_voxel :949-1004
    const float f2c = 1.0f / (float)s.subgrid_size;
    const int bx = clampi((int)((float)ix * f2c), 0, s.cx - 1), by = clampi((int)((float)iy * f2c), 0, s.cy - 1),
              bz = clampi((int)((float)iz * f2c), 0, s.cz - 1);
    const uint32_t slot = s.slots[((size_t)bx * s.cy + by) * s.cz + bz];
    if (slot < SLOT_LINEAR) {
        const int spd = s.subgrid_size + 1;
        const int ox = (int)(slot & 0x3FFu) * spd + (ix - bx * s.subgrid_size);
        const int oy = (int)((slot >> 10) & 0x3FFu) * spd + (iy - by * s.subgrid_size);
        const int oz = (int)((slot >> 20) & 0x3FFu) * spd + (iz - bz * s.subgrid_size);
        return texel(s, ox, oy, oz) * s.value_range + s.min_value;
    }
    return sample(s, vec3(s.box_lower[0] + (float)ix * s.voxel_size[0], s.box_lower[1] + (float)iy * s.voxel_size[1],
                          s.box_lower[2] + (float)iz * s.voxel_size[2]));
}
NT_DI float triangle_fraction(float a0, float a1, float a2, int num_inside) {  // sdf_mc.py:112-162
    if (num_inside == 3) return 1.0f;
    if (num_inside == 0) return 0.0f;
    float d0 = a0, d1 = a1, d2 = a2;
    if (num_inside == 1) {
        if (a1 < 0.0f) { d0 = a1; d1 = a2; d2 = a0; }
        else if (a2 < 0.0f) { d0 = a2; d1 = a0; d2 = a1; }
    } else {
        if (a1 >= 0.0f) { d0 = a1; d1 = a2; d2 = a0; }
        else if (a2 >= 0.0f) { d0 = a2; d1 = a0; d2 = a1; }
    }
    const float denom = (d0 - d1) * (d0 - d2);
    if (fabsf(denom) < 1e-8f) return num_inside == 1 ? 0.0f : 1.0f;
    const float fr = clampf((d0 * d0) / denom, 0.0f, 1.0f);
    return num_inside == 2 ? 1.0f - fr : fr;
}
NT_DI int mc_cx(int i) { return ((i & 3) ^ ((i & 3) >> 1)) & 1; }  // _mc_corner_offset :206-213
NT_DI int mc_cy(int i) { return (i >> 1) & 1; }
NT_DI int mc_cz(int i) { return (i >> 2) & 1; }

__global__ void __launch_bounds__(256) hydro_collide_kernel(nt_hydro_args a) {
    __shared__ int range[6];
    for (int pair_idx = blockIdx.x; pair_idx < a.pair_count; pair_idx += gridDim.x) {
        int sa = a.pairs[2 * pair_idx], sb = a.pairs[2 * pair_idx + 1];
        int ia = a.shape_sdf_index[sa], ib = a.shape_sdf_index[sb];
        if (ia < 0 || ib < 0 || ia >= a.sdf_count || ib >= a.sdf_count) continue;
        if (a.sdf_table[ib].voxel_radius > a.sdf_table[ia].voxel_radius) {  // keep the finer SDF as shape B (:1362-1366)
            int t = sa; sa = sb; sb = t;
            t = ia; ia = ib; ib = t;
        }
        const nt_sdf A = a.sdf_table[ia], B = a.sdf_table[ib];
        if (A.cx <= 0 || B.cx <= 0) continue;
        const float gap_sum = a.shape_gap[sa] + a.shape_gap[sb];
        const float margin_a = a.shape_data[4 * sa + 3], margin_b = a.shape_data[4 * sb + 3];
        const float kh_a = a.shape_kh[sa], kh_b = a.shape_kh[sb];
        const xform X_b = load_xform(a.shape_transform + 7 * sb);
        const xform X_b2a = xform_inverse(load_xform(a.shape_transform + 7 * sa)) * X_b;
        const vec3 vs(B.voxel_size[0], B.voxel_size[1], B.voxel_size[2]), blo(B.box_lower[0], B.box_lower[1], B.box_lower[2]);
        const int nx = B.cx * B.subgrid_size, ny = B.cy * B.subgrid_size, nz = B.cz * B.subgrid_size;
        __syncthreads();
        if (threadIdx.x == 0) {  // candidate voxel range: A's SDF box (widened by the gap) seen from B's grid
            const xform X_a2b = xform_inverse(X_b2a);
            vec3 lo(1e30f, 1e30f, 1e30f), hi(-1e30f, -1e30f, -1e30f);
            for (int k = 0; k < 8; ++k) {
                vec3 c((k & 1) ? A.box_upper[0] : A.box_lower[0], (k & 2) ? A.box_upper[1] : A.box_lower[1],
                       (k & 4) ? A.box_upper[2] : A.box_lower[2]);
                vec3 q = xform_point(X_a2b, c);
                lo = vmin(lo, q);
                hi = vmax(hi, q);
            }
            const int n[3] = {nx, ny, nz};
            for (int k = 0; k < 3; ++k) {
                int l = (int)floorf((vget(lo, k) - gap_sum - vget(blo, k)) / vget(vs, k)) - 1;
                int h = (int)ceilf((vget(hi, k) + gap_sum - vget(blo, k)) / vget(vs, k)) + 1;
                range[k] = l < 0 ? 0 : l;
                range[3 + k] = h > n[k] ? n[k] : h;
            }
        }
        __syncthreads();
        const int x0 = range[0], y0 = range[1], z0 = range[2];
        const int wx = range[3] - x0, wy = range[4] - y0, wz = range[5] - z0;
        if (wx <= 0 || wy <= 0 || wz <= 0) continue;
        const vec3 step_x = xform_vector(X_b2a, vec3(vs.x, 0.0f, 0.0f)), step_y = xform_vector(X_b2a, vec3(0.0f, vs.y, 0.0f)),
                   step_z = xform_vector(X_b2a, vec3(0.0f, 0.0f, vs.z));
        const float cmin = a.edge_clamp_min, cmax = 1.0f - a.edge_clamp_min;
        for (int v = threadIdx.x; v < wx * wy * wz; v += blockDim.x) {
            const int x = x0 + v % wx, y = y0 + (v / wx) % wy, z = z0 + v / (wx * wy);
            // mc_iterate_voxel_vertices (:1716-1798)
            const vec3 base_b = blo + cw_mul(vec3((float)x, (float)y, (float)z), vs);
            const vec3 base_a = xform_point(X_b2a, base_b);
            float cv[8], cself[8], cother[8];
            int cube = 0;
            bool any_gap = false, valid = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ox = mc_cx(i), oy = mc_cy(i), oz = mc_cz(i);
                const vec3 pa = base_a + (float)ox * step_x + (float)oy * step_y + (float)oz * step_z;
                const float v_self = sample_at_voxel(B, x + ox, y + oy, z + oz);
                const float v_other = sample(A, pa);
                if (v_self != v_self || v_other != v_other) valid = false;
                const float es = v_self - margin_b, eo = v_other - margin_a;
                const float vd = (-kh_a * eo) - (-kh_b * es);  // p_other - p_self, linear_pressure :237-248
                cv[i] = vd; cself[i] = es; cother[i] = eo;
                if (vd < 0.0f) cube |= 1 << i;
                if (es + eo <= gap_sum) any_gap = true;
            }
            if (!valid || !any_gap) continue;
            const int t0 = a.tri_range[cube], t1 = a.tri_range[cube + 1];
            for (int fi = 0; fi < (t1 - t0) / 3; ++fi) {
                // mc_calc_face_texture (:282-362)
                vec3 fv[3];
                float vsdf[3], vsep[3];
                int n_in = 0;
#pragma unroll
                for (int vi = 0; vi < 3; ++vi) {
                    const int ca = a.flat_edge_verts[2 * (t0 + 3 * fi + vi)], cb = a.flat_edge_verts[2 * (t0 + 3 * fi + vi) + 1];
                    const float vd = cv[cb] - cv[ca];
                    const float t = fabsf(vd) < 1.0e-10f ? 0.5f : clampf((0.0f - cv[ca]) / vd, cmin, cmax);
                    const vec3 p0((float)mc_cx(ca), (float)mc_cy(ca), (float)mc_cz(ca)), p1((float)mc_cx(cb), (float)mc_cy(cb), (float)mc_cz(cb));
                    const vec3 vol = p0 + t * (p1 - p0) + vec3((float)x, (float)y, (float)z);
                    fv[vi] = blo + cw_mul(vol, vs);
                    const float s_self = cself[ca] + t * (cself[cb] - cself[ca]);
                    const float s_other = cother[ca] + t * (cother[cb] - cother[ca]);
                    vsdf[vi] = s_self;
                    vsep[vi] = s_self + s_other;
                    if (vsep[vi] < 0.0f) n_in += 1;
                }
                const vec3 n = cross(fv[1] - fv[0], fv[2] - fv[0]);
                const float n_sq = dot(n, n);
                float garea = 0.0f;
                vec3 normal(0.0f, 0.0f, 1.0f);
                if (!(n_sq < 1.0e-20f)) {
                    const float inv = 1.0f / sqrtf(n_sq);
                    normal = n * inv;
                    garea = (n_sq * inv) * 0.5f;
                }
                const vec3 center = ((fv[0] + fv[1]) + fv[2]) / 3.0f;
                const float adj = ((vsdf[0] + vsdf[1]) + vsdf[2]) / 3.0f;
                const float sep = ((vsep[0] + vsep[1]) + vsep[2]) / 3.0f;
                const float farea = garea * triangle_fraction(vsep[0], vsep[1], vsep[2], n_in);
                if (garea <= 0.0f) continue;
                if (!(sep < 0.0f) && sep > gap_sum) continue;  // classify_hydroelastic_contact > 0: outside the gap band
                const float pressure = sep < 0.0f ? fmaxw(-kh_b * adj, 0.0f) : 0.0f;
                const float area = sep < 0.0f ? farea : garea;
                float stiff;
                if (sep < 0.0f) stiff = (area * pressure) / fmaxw(-sep, 1e-20f);
                else {
                    const float den = kh_a + kh_b;
                    stiff = a.margin_contact_area * (den <= 0.0f ? 0.0f : (kh_a * kh_b) / den);
                }
                const int slot = atomicAdd(a.out_count, 1);
                if (slot < a.capacity) {
                    a.out_pair[slot] = pair_idx;
                    a.out_key[slot] = ((z * ny + y) * nx + x) * 5 + fi;
                    a.out_shapes[2 * slot] = sa;
                    a.out_shapes[2 * slot + 1] = sb;
                    const vec3 pw = xform_point(X_b, center), nw = xform_vector(X_b, normal);
                    float* o = a.out_data + 10 * (size_t)slot;
                    o[0] = pw.x; o[1] = pw.y; o[2] = pw.z; o[3] = nw.x; o[4] = nw.y; o[5] = nw.z;
                    o[6] = sep; o[7] = stiff; o[8] = area; o[9] = pressure;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// HydroelasticSDF.launch (sdf_hydroelastic.py:905-1296) for the collide pipeline, one workgroup per shape pair:
//   broad phase      broadphase_collision_pairs_count (:1330-1376): SAT of the two SDF boxes, the finer SDF becomes shape B,
//                    every 8^3 subgrid of B is a block
//   octree           count_iso_voxels_block / count_iso_voxel_children + scatter_iso_subblock (:1444-1700): blocks of 8, 4, 2, 1
//                    voxels survive while the pressure intervals of the two shapes over the block (1-Lipschitz bound of either
//                    SDF around the block centre) can still meet and the centre is within 2 r + gap of both surfaces
//   faces            generate_contacts_kernel without pre-pruning (:1982-2180) + decode_contacts_kernel (:1823-1928): marching
//                    cubes on p_a == p_b in every surviving voxel, one contact per face, stiffness area * pressure / |separation|
// The reference runs each level as count -> scan -> scatter launches over device-wide buffers.  Here the pair's workgroup walks
// B's blocks itself: the level-8 test for all blocks at once, then per surviving block the three lower levels as bit masks in
// LDS (8 / 64 / 512 lanes), an ordered compaction of the surviving voxels -- the order the reference's ordered scatter produces:
// (block, child of 4, child of 2, voxel), children by x + 2 y + 4 z -- and marching cubes on them.  Faces leave through one
// atomic per block of voxels; every row carries its rank inside the pair, so the final placement is deterministic.
// Fingerprint = (rank of the voxel in the pair's traversal) * 5 + face: the reference numbers voxels across ALL pairs in the
// arrival order of its pair list, which changes from run to run; inside a pair the order is the same.
// ------------------------------------------------------------------------------------------------
constexpr int HYDRO_MAX_BLOCKS = 4096;
struct HydroLds {
    unsigned char blk[HYDRO_MAX_BLOCKS];  // level-8 survivors
    unsigned char l4[8], l2[64], l1[512];
    int vox[512];                         // surviving voxels of the current block, traversal order
    int wsum[4];
    int n_vox, n_face, base, collide;
    int pair_vox, pair_face;              // running totals of the pair
};
struct HydroPair {
    nt_sdf A, B;
    int sa, sb;
    xform X_b, X_b2a;
    float gap_sum, margin_a, margin_b, kh_a, kh_b;
};
// one octree node of shape B: cube of `size` voxels at (x, y, z) (count_iso_voxels_block's body)
NT_DI bool hydro_node_survives(const HydroPair& p, int x, int y, int z, int size) {
    const nt_sdf& A = p.A;
    const nt_sdf& B = p.B;
    const float r = (float)size * B.voxel_radius;
    const float h = 0.5f * (float)size;
    const vec3 centre((float)x + h, (float)y + h, (float)z + h);
    const vec3 local_b = vec3(B.box_lower[0], B.box_lower[1], B.box_lower[2]) +
                         cw_mul(centre, vec3(B.voxel_size[0], B.voxel_size[1], B.voxel_size[2]));
    const vec3 point_a = xform_point(p.X_b2a, local_b);
    const float vb = (size & 1) == 0 ? sample_at_voxel(B, x + size / 2, y + size / 2, z + size / 2) : sample(B, local_b);
    const float va = sample(A, point_a);
    if (vb != vb || va != va) return false;
    const float eva = va - p.margin_a, evb = vb - p.margin_b;
    if (eva + evb > 2.0f * r + p.gap_sum) return false;
    const float pa_lo = -p.kh_a * (eva + r), pa_hi = -p.kh_a * (eva - r);  // linear_pressure (:237-248)
    const float pb_lo = -p.kh_b * (evb + r), pb_hi = -p.kh_b * (evb - r);
    return !(pa_hi < pb_lo || pb_hi < pa_lo);
}
// sat_box_intersection (collision_core.py:1281-1374) of the two SDF boxes
NT_DI bool hydro_sat(const xform& Ta, vec3 ea, const xform& Tb, vec3 eb) {
    vec3 axa[3] = {quat_rotate(Ta.q, vec3(1.0f, 0.0f, 0.0f)), quat_rotate(Ta.q, vec3(0.0f, 1.0f, 0.0f)), quat_rotate(Ta.q, vec3(0.0f, 0.0f, 1.0f))};
    vec3 axb[3] = {quat_rotate(Tb.q, vec3(1.0f, 0.0f, 0.0f)), quat_rotate(Tb.q, vec3(0.0f, 1.0f, 0.0f)), quat_rotate(Tb.q, vec3(0.0f, 0.0f, 1.0f))};
    auto separated = [&](vec3 axis) {
        const float len = length(axis);
        if (len < 1e-8f) return false;
        const vec3 n = axis / len;
        auto project = [&](const xform& T, const vec3* ax, vec3 e, float& lo, float& hi) {
            const float c = dot(T.p, n);
            float ext = 0.0f;
            ext += e.x * fabsf(dot(ax[0], n));
            ext += e.y * fabsf(dot(ax[1], n));
            ext += e.z * fabsf(dot(ax[2], n));
            lo = c - ext;
            hi = c + ext;
        };
        float la, ha, lb, hb;
        project(Ta, axa, ea, la, ha);
        project(Tb, axb, eb, lb, hb);
        return ha < lb || hb < la;
    };
    for (int i = 0; i < 3; ++i)
        if (separated(axa[i])) return false;
    for (int i = 0; i < 3; ++i)
        if (separated(axb[i])) return false;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            if (separated(cross(axa[i], axb[j]))) return false;
    return true;
}
struct HydroFace { vec3 pos; float oct0, oct1, depth, stiff; vec3 normal; float area, pressure; };  // normal / area / pressure: reduction only
// marching cubes of one voxel of B (mc_iterate_voxel_vertices :1716-1798, mc_calc_face_texture :282-362, the face filters and the
// decode of the unreduced path) in two parts: the corner samples of the voxel, and the evaluation of ONE face of its case.
struct HydroCorners {
    float cv[8], cself[8], cother[8];
    int t0, nfaces;  // triangle range of the marching-cubes case, faces of the case (0: nothing to do)
};
// one corner of a voxel: the two margin-relative signed distances (shape B at the voxel corner, shape A at the same point)
NT_DI void hydro_corner_sample(const HydroPair& p, int x, int y, int z, int i, float& es, float& eo) {
    const nt_sdf& A = p.A;
    const nt_sdf& B = p.B;
    const vec3 vs(B.voxel_size[0], B.voxel_size[1], B.voxel_size[2]), blo(B.box_lower[0], B.box_lower[1], B.box_lower[2]);
    const vec3 base_b = blo + cw_mul(vec3((float)x, (float)y, (float)z), vs);
    const vec3 base_a = xform_point(p.X_b2a, base_b);
    const vec3 step_x = xform_vector(p.X_b2a, vec3(vs.x, 0.0f, 0.0f)), step_y = xform_vector(p.X_b2a, vec3(0.0f, vs.y, 0.0f)),
               step_z = xform_vector(p.X_b2a, vec3(0.0f, 0.0f, vs.z));
    const int ox = mc_cx(i), oy = mc_cy(i), oz = mc_cz(i);
    const vec3 pa = base_a + (float)ox * step_x + (float)oy * step_y + (float)oz * step_z;
    es = sample_at_voxel(B, x + ox, y + oy, z + oz) - p.margin_b;
    eo = sample(A, pa) - p.margin_a;
}
// the voxel's marching-cubes case from its eight corner samples (es[i], eo[i] as hydro_corner_sample returns them)
NT_DI void hydro_voxel_classify(const nt_hydro_args& a, const HydroPair& p, const float* es8, const float* eo8, HydroCorners& c) {
    int cube = 0;
    bool any_gap = false, nan = false;
    c.nfaces = 0;
    c.t0 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float es = es8[i], eo = eo8[i];
        if (es != es || eo != eo) nan = true;  // (v - margin is NaN exactly when v is)
        const float vd = (-p.kh_a * eo) - (-p.kh_b * es);
        c.cv[i] = vd; c.cself[i] = es; c.cother[i] = eo;
        if (vd < 0.0f) cube |= 1 << i;
        if (es + eo <= p.gap_sum) any_gap = true;
    }
    if (nan || !any_gap) return;
    c.t0 = a.tri_range[cube];
    c.nfaces = (a.tri_range[cube + 1] - c.t0) / 3;
}
NT_DI void hydro_voxel_corners(const nt_hydro_args& a, const HydroPair& p, int x, int y, int z, HydroCorners& c) {
    const nt_sdf& A = p.A;
    const nt_sdf& B = p.B;
    const vec3 vs(B.voxel_size[0], B.voxel_size[1], B.voxel_size[2]), blo(B.box_lower[0], B.box_lower[1], B.box_lower[2]);
    const vec3 base_b = blo + cw_mul(vec3((float)x, (float)y, (float)z), vs);
    const vec3 base_a = xform_point(p.X_b2a, base_b);
    const vec3 step_x = xform_vector(p.X_b2a, vec3(vs.x, 0.0f, 0.0f)), step_y = xform_vector(p.X_b2a, vec3(0.0f, vs.y, 0.0f)),
               step_z = xform_vector(p.X_b2a, vec3(0.0f, 0.0f, vs.z));
    int cube = 0;
    bool any_gap = false;
    c.nfaces = 0;
    c.t0 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ox = mc_cx(i), oy = mc_cy(i), oz = mc_cz(i);
        const vec3 pa = base_a + (float)ox * step_x + (float)oy * step_y + (float)oz * step_z;
        const float v_self = sample_at_voxel(B, x + ox, y + oy, z + oz);
        const float v_other = sample(A, pa);
        if (v_self != v_self || v_other != v_other) return;
        const float es = v_self - p.margin_b, eo = v_other - p.margin_a;
        const float vd = (-p.kh_a * eo) - (-p.kh_b * es);
        c.cv[i] = vd; c.cself[i] = es; c.cother[i] = eo;
        if (vd < 0.0f) cube |= 1 << i;
        if (es + eo <= p.gap_sum) any_gap = true;
    }
    if (!any_gap) return;
    c.t0 = a.tri_range[cube];
    c.nfaces = (a.tri_range[cube + 1] - c.t0) / 3;
}
// face fi of the voxel's case; false: filtered out (degenerate, or beyond the gap band)
NT_DI bool hydro_voxel_face(const nt_hydro_args& a, const HydroPair& p, int x, int y, int z, const HydroCorners& c, int fi, HydroFace& f) {
    const nt_sdf& B = p.B;
    const vec3 vs(B.voxel_size[0], B.voxel_size[1], B.voxel_size[2]), blo(B.box_lower[0], B.box_lower[1], B.box_lower[2]);
    const float cmin = a.edge_clamp_min, cmax = 1.0f - a.edge_clamp_min;
    vec3 fv[3];
    float vsdf[3], vsep[3];
    int n_in = 0;
#pragma unroll
    for (int vi = 0; vi < 3; ++vi) {
        const int ca = a.flat_edge_verts[2 * (c.t0 + 3 * fi + vi)], cb = a.flat_edge_verts[2 * (c.t0 + 3 * fi + vi) + 1];
        auto sel = [](const float* v, int k) {  // value select, no private array indexing
            return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : k == 3 ? v[3] : k == 4 ? v[4] : k == 5 ? v[5] : k == 6 ? v[6] : v[7];
        };
        const float va0 = sel(c.cv, ca), va1 = sel(c.cv, cb);
        const float vd = va1 - va0;
        const float t = fabsf(vd) < 1.0e-10f ? 0.5f : clampf((0.0f - va0) / vd, cmin, cmax);
        const vec3 p0((float)mc_cx(ca), (float)mc_cy(ca), (float)mc_cz(ca)), p1((float)mc_cx(cb), (float)mc_cy(cb), (float)mc_cz(cb));
        const vec3 vol = p0 + t * (p1 - p0) + vec3((float)x, (float)y, (float)z);
        fv[vi] = blo + cw_mul(vol, vs);
        const float s_self = sel(c.cself, ca) + t * (sel(c.cself, cb) - sel(c.cself, ca));
        const float s_other = sel(c.cother, ca) + t * (sel(c.cother, cb) - sel(c.cother, ca));
        vsdf[vi] = s_self;
        vsep[vi] = s_self + s_other;
        if (vsep[vi] < 0.0f) n_in += 1;
    }
    const vec3 n = cross(fv[1] - fv[0], fv[2] - fv[0]);
    const float n_sq = dot(n, n);
    float garea = 0.0f;
    vec3 normal(0.0f, 0.0f, 1.0f);
    if (!(n_sq < 1.0e-20f)) {
        const float inv = 1.0f / sqrtf(n_sq);
        normal = n * inv;
        garea = (n_sq * inv) * 0.5f;
    }
    const vec3 center = ((fv[0] + fv[1]) + fv[2]) / 3.0f;
    const float adj = ((vsdf[0] + vsdf[1]) + vsdf[2]) / 3.0f;
    const float sep = ((vsep[0] + vsep[1]) + vsep[2]) / 3.0f;
    const float farea = garea * triangle_fraction(vsep[0], vsep[1], vsep[2], n_in);
    if (garea <= 0.0f) return false;
    if (!(sep < 0.0f) && sep > p.gap_sum) return false;  // classify_hydroelastic_contact > 0
    const float pressure = sep < 0.0f ? fmaxw(-p.kh_b * adj, 0.0f) : 0.0f;
    const float area = sep < 0.0f ? farea : garea;
    float stiff;
    if (sep < 0.0f) stiff = area * pressure / fmaxw(-sep, 1e-20f);
    else {
        const float den = p.kh_a + p.kh_b;
        stiff = a.margin_contact_area * (den <= 0.0f ? 0.0f : (p.kh_a * p.kh_b) / den);
    }
    f.pos = center;  // B's frame; the buffer keeps the normal as its octahedral code (export_hydroelastic_contact_to_buffer)
    red_encode_oct(normal, f.oct0, f.oct1);
    f.depth = sep;
    f.stiff = stiff;
    f.normal = normal;
    f.area = area;
    f.pressure = pressure;
    return true;
}
// all faces of a voxel at once (the one-workgroup-per-pair kernels); returns the number kept (<= 5), in face order
NT_DI int hydro_voxel_faces(const nt_hydro_args& a, const HydroPair& p, int x, int y, int z, HydroFace* out, int* face_id) {
    HydroCorners c;
    hydro_voxel_corners(a, p, x, y, z, c);
    int kept = 0;
    for (int fi = 0; fi < c.nfaces; ++fi)
        if (hydro_voxel_face(a, p, x, y, z, c, fi, out[kept])) {
            face_id[kept] = fi;
            kept += 1;
        }
    return kept;
}

#ifdef NT_HYDRO_TIMING  // measurement builds (tools/hydro_timing.py): cycles of workgroup lane 0 per phase, summed over all pairs
__device__ unsigned long long nt_hydro_timing[16];
#define NT_HT(i, t_last)                                                  \
    do {                                                                  \
        if (threadIdx.x == 0) {                                           \
            const unsigned long long now_ = clock64();                    \
            atomicAdd(&nt_hydro_timing[i], now_ - (t_last));              \
            (t_last) = now_;                                              \
        }                                                                 \
    } while (0)
#else
#define NT_HT(i, t_last) do { } while (0)
#endif
// ---- reduce_contacts = True: what the workgroup keeps of a pair between the face pass and the reduction (see hydro_reduce_pair)
constexpr int HYDRO_CHUNK_CAP = 384;   // face blocks of one pair (staged: one per 16 iso voxels; single kernel: one per 256)
constexpr int HYDRO_ENTRIES = 50;      // 20 normal bins + 15 voxel groups + 15 speculative voxel groups
constexpr int HYDRO_STAGE = 128;
constexpr int HYDRO_FACE_WORDS = 12;   // centre[3] normal[3] separation area pressure | key | contact id << 5 | normal bin | pad
struct HydroRedLds {
    int chunk[HYDRO_CHUNK_CAP][2];
    int cstart[HYDRO_CHUNK_CAP];                  // rank of the block's first face inside the pair (blocks are small and scattered:
                                                  // the passes below walk the pair's faces by RANK, all lanes busy, not block by block)
    int idbase[HYDRO_CHUNK_CAP], voxbase[HYDRO_CHUNK_CAP];  // what to add to the contact ids / voxel ranks stored in a block's records
                                                  // (staged face pass: they are relative to the block; single kernel: zero)
    int tkey[HYDRO_ENTRIES][RED_VALUES];          // the winner's key (voxel rank * 5 + face)
    float agg[RED_BINS][10];                      // agg_force[3] weighted_pos_sum[3] weight_sum agg_depth_volume[3]
    float tdepth[RED_BINS], tnormal[RED_BINS][3]; // total_depth_reduced / total_normal_reduced
    unsigned long long tbl[HYDRO_ENTRIES][RED_VALUES];
    int tslot[HYDRO_ENTRIES][RED_VALUES];         // where the winner's face record sits
    unsigned int ekey[HYDRO_ENTRIES];             // first use of the entry (hashtable insertion order), ~0u = never
    int order[HYDRO_ENTRIES], ucount[HYDRO_ENTRIES], ubase[HYDRO_ENTRIES], n_entries;
    float wpen[HYDRO_ENTRIES * RED_VALUES], wn[HYDRO_ENTRIES * RED_VALUES][3];
    int wnbin[HYDRO_ENTRIES * RED_VALUES];
    unsigned char wuniq[HYDRO_ENTRIES * RED_VALUES];
    short seq[HYDRO_ENTRIES * RED_VALUES];
    float m_unr[RED_BINS], s1[RED_BINS], s2[RED_BINS];  // moment matching: unreduced / reduced friction moments per normal bin
    float wlever[HYDRO_ENTRIES * RED_VALUES];           // ... lever arm of a winner about its bin's centre of pressure
    float max_pen[HYDRO_ENTRIES];                        // deepest winner of the entry
    unsigned char anchor[HYDRO_ENTRIES];                 // the entry exports an anchor contact
    int n_chunk, n_faces, pair_kept, overflow, rows, row_base;
    float stage[HYDRO_STAGE][9];  // a tile of face records for the ordered aggregate sums
    signed char stage_bin[HYDRO_STAGE];
};
NT_DI unsigned long long hydro_value(float score, int cid) {  // _make_contact_value_fast
    return ((unsigned long long)red_float_flip(score) << 32) | (unsigned long long)(unsigned int)cid;
}
NT_DI int hydro_entry_of_voxel(int vox, bool speculative) { return (speculative ? RED_BINS + 15 : RED_BINS) + vox / RED_VALUES; }
NT_DI quat hydro_matching_rotation(vec3 nsum, vec3 agg, float agg_mag) {  // _compute_normal_matching_rotation :261-292
    quat q(0.0f, 0.0f, 0.0f, 1.0f);
    const float sel_mag = length(nsum);
    if (sel_mag > 1e-8f && agg_mag > 1e-20f) {
        const vec3 sel = nsum / sel_mag, ad = agg / agg_mag;
        const vec3 cr = cross(sel, ad);
        const float cr_mag = length(cr), d = dot(sel, ad);
        bool have = false;
        vec3 axis;
        float angle = 0.0f;
        if (cr_mag > 1e-8f) {
            axis = cr / cr_mag;
            angle = acosf(fminw(fmaxw(d, -1.0f), 1.0f));
            have = true;
        } else if (d < 0.0f) {
            vec3 perp(1.0f, 0.0f, 0.0f);
            if (fabsf(dot(sel, perp)) > 0.9f) perp = vec3(0.0f, 1.0f, 0.0f);
            axis = normalize(cross(sel, perp));
            angle = 3.14159265359f;
            have = true;
        }
        if (have) {  // wp.quat_from_axis_angle
            const float half = angle * 0.5f, sn = sinf(half);
            q = quat(axis.x * sn, axis.y * sn, axis.z * sn, cosf(half));
        }
    }
    return q;
}
// The reduction of ONE pair after its faces are in a.face_rec (blocks listed in R.chunk, face order): aggregates per normal bin
// (ordered sums, one lane per bin), table registration of the buffered contacts, winners, reduced depth sums in the hashtable's
// insertion order, export.  contact_reduction_hydroelastic.py:596-755 (reduce), :756-850 (accumulate depth), :983-1460 (export).
struct EntryExport { vec3 anchor_pos; float shared, alpha, l_avg, uniform_fs, anchor_fs; };
// EXTRAS: anchor contacts / moment matching compiled in.  The default options do not use them, and inlined next to the face pass
// their registers push the kernel past 256 VGPR -- one workgroup per CU instead of two (measured: 309 -> 577 ms per hydro_bin
// frame); the instance that has them is therefore called, not inlined.
template <bool EXTRAS>
NT_DI void hydro_reduce_pair_impl(const nt_hydro_args& a, const HydroPair& p, int pair_idx, HydroRedLds& R) {
    const int t = threadIdx.x, nt_ = blockDim.x;
    const bool normal_matching = (a.reduce & 4) != 0, moment_matching = EXTRAS && (a.reduce & 16) != 0;
    const bool anchor_contact = EXTRAS && ((a.reduce & 8) != 0 || moment_matching);
#ifdef NT_HYDRO_TIMING
    unsigned long long ht = clock64();
#endif
    auto face_block = [&](int rank) {  // rank of a face inside the pair -> the last block that starts at or before it
        int lo = 0, hi = R.n_chunk;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (R.cstart[mid] <= rank) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    auto face_slot = [&](int rank) {  // ... -> its record
        const int lo = face_block(rank);
        return R.chunk[lo][0] + (rank - R.cstart[lo]);
    };
    auto face_cid = [&](int rank) {  // ... -> its contact id inside the pair (0: not buffered)
        const int lo = face_block(rank);
        const int cid = reinterpret_cast<const int*>(a.face_rec + HYDRO_FACE_WORDS * (size_t)(R.chunk[lo][0] + (rank - R.cstart[lo])))[10] >> 5;
        return cid > 0 ? cid + R.idbase[lo] : 0;
    };
    for (int k = t; k < HYDRO_ENTRIES * RED_VALUES; k += nt_) {
        R.tbl[k / RED_VALUES][k % RED_VALUES] = 0ull;
        R.tslot[k / RED_VALUES][k % RED_VALUES] = -1;
        R.wuniq[k] = 0;
    }
    for (int k = t; k < HYDRO_ENTRIES; k += nt_) { R.ekey[k] = ~0u; R.ucount[k] = 0; }
    __syncthreads();
    // ---- aggregates of ALL penetrating faces per (exact-normal) bin, face order: lane = bin.  The faces pass through LDS in tiles
    // (every lane loads one record, coalesced) so that the twenty summing lanes walk LDS instead of paying an HBM round trip per face
    {
        vec3 force, wps, adv;
        float ws = 0.0f;
        unsigned int first = ~0u;
        for (int r0 = 0; r0 < R.n_faces; r0 += HYDRO_STAGE) {
            const int m = R.n_faces - r0 < HYDRO_STAGE ? R.n_faces - r0 : HYDRO_STAGE;
            for (int k = t; k < m; k += nt_) {
                const float* rec = a.face_rec + HYDRO_FACE_WORDS * (size_t)face_slot(r0 + k);
                float* o = R.stage[k];
                o[0] = rec[0]; o[1] = rec[1]; o[2] = rec[2]; o[3] = rec[3]; o[4] = rec[4]; o[5] = rec[5];
                o[6] = rec[6]; o[7] = rec[7]; o[8] = rec[8];
                R.stage_bin[k] = rec[6] < 0.0f ? (reinterpret_cast<const int*>(rec)[10] & 31) : -1;
            }
            __syncthreads();
            if (t < RED_BINS)
                for (int k = 0; k < m; ++k) {
                    if (R.stage_bin[k] != t) continue;
                    const float* rec = R.stage[k];
                    const vec3 n(rec[3], rec[4], rec[5]), ctr(rec[0], rec[1], rec[2]);
                    const float fw = rec[7] * rec[8];
                    force += fw * n;
                    wps += fw * ctr;
                    ws += fw;
                    adv += (rec[7] * (-rec[6])) * n;
                    if (first == ~0u) first = (unsigned int)(r0 + k);
                }
            __syncthreads();
        }
      if (t < RED_BINS) {
        float* g = R.agg[t];
        g[0] = force.x; g[1] = force.y; g[2] = force.z; g[3] = wps.x; g[4] = wps.y; g[5] = wps.z; g[6] = ws;
        g[7] = adv.x; g[8] = adv.y; g[9] = adv.z;
        R.tdepth[t] = 0.0f;
        R.tnormal[t][0] = R.tnormal[t][1] = R.tnormal[t][2] = 0.0f;
        if (first != ~0u) R.ekey[t] = first;
      }
    }
    if (t < RED_BINS) R.m_unr[t] = R.s1[t] = R.s2[t] = 0.0f;
    __syncthreads();
    if (moment_matching) {
        // unreduced friction moment of every buffered penetrating contact about its bin's centre of pressure (:717-727), summed per
        // bin in CONTACT order.  Contact ids follow the voxels but, inside a pruned voxel, not the faces: word 11 of the record at
        // rank (cid - 1) receives the rank of contact cid, then the contacts pass through LDS in tiles like the faces above.
        for (int j = t; j < R.n_faces; j += nt_) {
            const int cid = face_cid(j);
            if (cid > 0) reinterpret_cast<int*>(a.face_rec + HYDRO_FACE_WORDS * (size_t)face_slot(cid - 1))[11] = j;
        }
        __threadfence();
        __syncthreads();
        for (int c0 = 0; c0 < R.pair_kept; c0 += HYDRO_STAGE) {
            const int m = R.pair_kept - c0 < HYDRO_STAGE ? R.pair_kept - c0 : HYDRO_STAGE;
            for (int k = t; k < m; k += nt_) {
                const int rank = reinterpret_cast<const int*>(a.face_rec + HYDRO_FACE_WORDS * (size_t)face_slot(c0 + k))[11];
                const float* rec = a.face_rec + HYDRO_FACE_WORDS * (size_t)face_slot(rank);
                R.stage_bin[k] = -1;
                if (!(rec[6] < 0.0f)) continue;
                float ox, oy;
                red_encode_oct(vec3(rec[3], rec[4], rec[5]), ox, oy);
                const vec3 n = red_decode_oct(ox, oy);
                const int b = red_get_slot(n);
                const float* g = R.agg[b];
                if (!(g[6] > 1e-20f)) continue;
                const vec3 anchor_pos = vec3(g[3], g[4], g[5]) / g[6];
                R.stage[k][0] = rec[7] * rec[8] * length(cross(vec3(rec[0], rec[1], rec[2]) - anchor_pos, n));
                R.stage_bin[k] = (signed char)b;
            }
            __syncthreads();
            if (t < RED_BINS)
                for (int k = 0; k < m; ++k)
                    if (R.stage_bin[k] == t) R.m_unr[t] += R.stage[k][0];
            __syncthreads();
        }
    }
    NT_HT(1, ht);
    // ---- table registration (pass 0) and the winners' record positions (pass 1), one lane per buffered contact
    const float* lo = a.shape_aabb_lower + 3 * p.sb;
    const float* hi = a.shape_aabb_upper + 3 * p.sb;
    const float aabb_size = length(vec3(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]));
    for (int pass = 0; pass < 2; ++pass) {
        {
            for (int j = t; j < R.n_faces; j += nt_) {
                const int blk = face_block(j);
                const int fslot = R.chunk[blk][0] + (j - R.cstart[blk]);
                const float* rec = a.face_rec + HYDRO_FACE_WORDS * (size_t)fslot;
                int cid = reinterpret_cast<const int*>(rec)[10] >> 5;
                if (cid <= 0) continue;
                cid += R.idbase[blk];
                const unsigned int ucid = (unsigned int)cid;
                const int face_key = reinterpret_cast<const int*>(rec)[9] + 5 * R.voxbase[blk];
                auto offer = [&](int e, int s, float score, unsigned int key) {
                    if (pass == 0) {
                        atomicMax(&R.tbl[e][s], hydro_value(score, cid));
                        atomicMin(&R.ekey[e], key);
                    } else if ((unsigned int)(R.tbl[e][s] & 0xFFFFFFFFull) == ucid) {
                        R.tslot[e][s] = fslot;
                        R.tkey[e][s] = face_key;
                    }
                };
                const vec3 ctr(rec[0], rec[1], rec[2]);
                const float depth = rec[6];
                float ox, oy;
                red_encode_oct(vec3(rec[3], rec[4], rec[5]), ox, oy);
                const vec3 n = red_decode_oct(ox, oy);
                int vox = red_voxel_index(ctr, lo, hi, a.shape_voxel_res + 3 * p.sb);
                vox = vox < 0 ? 0 : (vox > RED_VOXELS - 1 ? RED_VOXELS - 1 : vox);
                const unsigned int key = (1u << 30) + 2u * ucid;
                if (!(depth < 0.0f)) {  // speculative contacts compete in their own voxel groups
                    offer(hydro_entry_of_voxel(vox, true), vox % RED_VALUES, -depth, key);
                    continue;
                }
                const int b = red_get_slot(n);
                if (depth < 0.0001f * aabb_size) {  // BETA_THRESHOLD
                    const float* g = R.agg[b];
                    const vec3 anchor = vec3(g[3], g[4], g[5]) / g[6];
                    vec3 u, v;
                    red_face_frame(b, u, v);
                    const vec3 rel = ctr - anchor;
                    const float px = dot(rel, u), py = dot(rel, v), pen_w = fmaxw(-depth, 0.0f);
                    for (int d = 0; d < RED_DIRS; ++d) offer(b, d, (px * RED_DIR[d][0] + py * RED_DIR[d][1]) * pen_w, key);
                }
                offer(b, RED_DIRS, -depth, key);
                offer(hydro_entry_of_voxel(vox, false), vox % RED_VALUES, -depth, key + 1u);
            }
        }
        __syncthreads();
    }
    NT_HT(2, ht);
    // ---- winners: unique contacts of every entry in slot order, their depth / decoded normal / normal bin
    for (int i = t; i < HYDRO_ENTRIES * RED_VALUES; i += nt_) {
        const int e = i / RED_VALUES, sl = i % RED_VALUES;
        const unsigned long long v = R.tbl[e][sl];
        if (v == 0ull) continue;
        bool uniq = true;
        for (int s2 = 0; s2 < sl; ++s2) uniq = uniq && (R.tbl[e][s2] & 0xFFFFFFFFull) != (v & 0xFFFFFFFFull);
        const float* rec = a.face_rec + HYDRO_FACE_WORDS * (size_t)R.tslot[e][sl];
        R.wpen[i] = -rec[6];  // (every occupied slot: a voxel entry reads its bin's deepest-contact slot)
        if (!uniq) continue;
        float ox, oy;
        red_encode_oct(vec3(rec[3], rec[4], rec[5]), ox, oy);
        const vec3 n = red_decode_oct(ox, oy);
        R.wuniq[i] = 1;
        R.wn[i][0] = n.x; R.wn[i][1] = n.y; R.wn[i][2] = n.z;
        R.wnbin[i] = e < RED_BINS ? e : (rec[6] < 0.0f ? red_get_slot(n) : -1);
        atomicAdd(&R.ucount[e], 1);
    }
    __syncthreads();
    NT_HT(3, ht);
    // ---- entries in insertion order; reduced depth / normal sums in that order (one lane: <= 350 short steps on LDS)
    if (t < HYDRO_ENTRIES) {
        float mp = 0.0f;
        for (int sl = 0; sl < RED_VALUES; ++sl)
            if (R.wuniq[t * RED_VALUES + sl] && R.wpen[t * RED_VALUES + sl] > 0.0f) mp = fmaxw(mp, R.wpen[t * RED_VALUES + sl]);
        R.max_pen[t] = mp;
        bool anc = false;
        if (anchor_contact && t < RED_BINS && R.ucount[t] > 0) {  // reliable aggregate direction, a penetrating winner, a weight
            const float* g = R.agg[t];
            anc = length(vec3(g[7], g[8], g[9])) > 1e-8f && length(vec3(g[0], g[1], g[2])) > 1e-20f && mp > 0.0f && g[6] > 1e-20f;
        }
        R.anchor[t] = anc ? 1 : 0;
        int rank = -1;
        if (R.ekey[t] != ~0u) {
            rank = 0;
            for (int e2 = 0; e2 < HYDRO_ENTRIES; ++e2) rank += (R.ekey[e2] < R.ekey[t]) ? 1 : 0;  // keys are distinct
            R.order[rank] = t;
        }
    }
    if (t == 0) {
        int n = 0;
        for (int e = 0; e < HYDRO_ENTRIES; ++e) n += R.ekey[e] != ~0u ? 1 : 0;
        R.n_entries = n;
    }
    __syncthreads();
    if (t == 0) {
        int rows = 0;
        for (int r = 0; r < R.n_entries; ++r) {
            const int e = R.order[r];
            R.ubase[e] = rows;
            rows += R.ucount[e] + R.anchor[e];
        }
        R.rows = rows;
        R.row_base = rows > 0 ? atomicAdd(a.out_count, rows) : 0;
    }
    __syncthreads();
    for (int i = t; i < HYDRO_ENTRIES * RED_VALUES; i += nt_) {  // the winners in export order (entry order, then slot order)
        if (i % RED_VALUES == 0 && R.anchor[i / RED_VALUES]) R.seq[R.ubase[i / RED_VALUES] + R.ucount[i / RED_VALUES]] = -1;  // its anchor
        if (!R.wuniq[i]) continue;
        const int e = i / RED_VALUES, sl = i % RED_VALUES;
        int idx = 0;
        for (int s2 = 0; s2 < sl; ++s2) idx += R.wuniq[e * RED_VALUES + s2];
        R.seq[R.ubase[e] + idx] = (short)i;
    }
    __syncthreads();
    if (t == 0) {  // ... whose depths / normals are summed per normal bin in that order: a short serial walk (the pair's rows)
        for (int r = 0; r < R.rows; ++r) {
            const int i = R.seq[r];
            if (i < 0 || !(R.wpen[i] > 0.0f) || R.wnbin[i] < 0) continue;  // depth < 0 <=> pen > 0
            const int nb = R.wnbin[i];
            const float pen = R.wpen[i];
            R.tdepth[nb] += pen;
            R.tnormal[nb][0] += pen * R.wn[i][0];
            R.tnormal[nb][1] += pen * R.wn[i][1];
            R.tnormal[nb][2] += pen * R.wn[i][2];
        }
    }
    __syncthreads();
    NT_HT(4, ht);
    // ---- export
    const float den = p.kh_a + p.kh_b;
    const float mca_k = a.margin_contact_area * (den <= 0.0f ? 0.0f : (p.kh_a * p.kh_b) / den);
    auto matched_normal = [&](int nb, vec3 n) {  // the winner's normal after its bin's matching rotation (gate: reliable direction)
        const float* g = R.agg[nb];
        const vec3 agg(g[0], g[1], g[2]);
        const float mag = length(agg);
        if (normal_matching && length(vec3(g[7], g[8], g[9])) > 1e-8f && mag > 1e-20f)
            return normalize(quat_rotate(hydro_matching_rotation(vec3(R.tnormal[nb][0], R.tnormal[nb][1], R.tnormal[nb][2]), agg, mag), n));
        return n;
    };
    if (moment_matching) {  // accumulate_moments_kernel :852-980
        for (int i = t; i < HYDRO_ENTRIES * RED_VALUES; i += nt_) {
            R.wlever[i] = -1.0f;
            if (!R.wuniq[i] || !(R.wpen[i] > 0.0f) || R.wnbin[i] < 0) continue;
            const int nb = R.wnbin[i];
            const float* g = R.agg[nb];
            if (!(g[6] > 1e-20f)) continue;
            const float* rec = a.face_rec + HYDRO_FACE_WORDS * (size_t)R.tslot[i / RED_VALUES][i % RED_VALUES];
            const vec3 anchor_pos = vec3(g[3], g[4], g[5]) / g[6];
            const vec3 rn = matched_normal(nb, vec3(R.wn[i][0], R.wn[i][1], R.wn[i][2]));
            R.wlever[i] = length(cross(vec3(rec[0], rec[1], rec[2]) - anchor_pos, rn));
        }
        __syncthreads();
        if (t == 0)
            for (int r = 0; r < R.rows; ++r) {
                const int i = R.seq[r];
                if (i < 0 || R.wlever[i] < 0.0f) continue;
                const int nb = R.wnbin[i];
                const float pl = R.wpen[i] * R.wlever[i];
                R.s1[nb] += pl;
                R.s2[nb] += pl * R.wlever[i];
            }
        __syncthreads();
    }
    auto bin_values = [&](int b, vec3& agg, float& agg_mag, bool& reliable, float& eff) {
        const float* g = R.agg[b];
        agg = vec3(g[0], g[1], g[2]);
        agg_mag = length(agg);
        reliable = length(vec3(g[7], g[8], g[9])) > 1e-8f && agg_mag > 1e-20f;
        const vec3 ns(R.tnormal[b][0], R.tnormal[b][1], R.tnormal[b][2]);
        if (normal_matching) {
            eff = length(ns);
            if (eff < 1e-8f) eff = R.tdepth[b];
        } else {
            eff = R.tdepth[b];
        }
    };
    // what a normal-bin entry with a reliable aggregate direction shares among its rows (export kernel :1141-1251)
    auto entry_export = [&](int e, float agg_mag, float eff) {
        EntryExport x;
        const float* g = R.agg[e];
        const int add_anchor = R.anchor[e];
        const float anchor_depth = R.max_pen[e];
        if (add_anchor) x.anchor_pos = vec3(g[3], g[4], g[5]) / g[6];
        const float tdwa = eff + (float)add_anchor * anchor_depth;
        x.shared = (agg_mag > 1e-20f && tdwa > 0.0f) ? agg_mag / tdwa : 0.0f;
        x.alpha = 0.0f; x.l_avg = 0.0f; x.uniform_fs = 1.0f; x.anchor_fs = 1.0f;
        if (moment_matching) {
            const float m_unr = R.m_unr[e], m_red = R.s1[e], m_red2 = R.s2[e];
            const float s0 = R.tdepth[e] + (float)add_anchor * anchor_depth;
            if (m_unr > 1e-20f && s0 > 1e-20f && m_red > 1e-20f && agg_mag > 1e-20f) {
                const float m_target = m_unr * tdwa / agg_mag;
                if (m_target < m_red) {
                    x.uniform_fs = m_target / m_red;
                } else {
                    x.l_avg = m_red / s0;
                    const float variance = m_red2 * s0 - m_red * m_red;
                    if (variance > 1e-20f) x.alpha = fminw(fmaxw((m_target - m_red) * m_red / variance, 0.0f), 1.0f);
                }
            }
            if (add_anchor == 1 && anchor_depth > 0.0f)
                x.anchor_fs = fmaxw(1e-2f, 1.0f + (R.tdepth[e] / anchor_depth) * (1.0f - x.uniform_fs) - x.alpha);
        }
        return x;
    };
    for (int i = t; i < HYDRO_ENTRIES * RED_VALUES; i += nt_) {
        if (!R.wuniq[i]) continue;
        const int e = i / RED_VALUES, sl = i % RED_VALUES;
        int idx = 0;
        for (int s2 = 0; s2 < sl; ++s2) idx += R.wuniq[e * RED_VALUES + s2];
        const int rank = R.ubase[e] + idx, slot = R.row_base + rank;
        if (slot >= a.capacity) continue;
        const float* rec = a.face_rec + HYDRO_FACE_WORDS * (size_t)R.tslot[e][sl];
        const float depth = rec[6];
        const vec3 n(R.wn[i][0], R.wn[i][1], R.wn[i][2]);
        vec3 final_n = n;
        float stiff, fscale = 1.0f;
        vec3 agg;
        float agg_mag = 0.0f, eff = 0.0f;
        bool reliable = false;
        if (e < RED_BINS) bin_values(e, agg, agg_mag, reliable, eff);
        if (reliable) {
            EntryExport x = entry_export(e, agg_mag, eff);
            if (normal_matching && depth < 0.0f)
                final_n = normalize(quat_rotate(hydro_matching_rotation(vec3(R.tnormal[e][0], R.tnormal[e][1], R.tnormal[e][2]), agg, agg_mag), n));
            stiff = x.shared;
            if (x.shared == 0.0f) stiff = depth < 0.0f ? rec[7] * rec[8] / fmaxw(-depth, 1e-20f) : mca_k;
            if (moment_matching && depth < 0.0f) {
                if (x.l_avg > 1e-20f) {
                    const float lever = length(cross(vec3(rec[0], rec[1], rec[2]) - x.anchor_pos, final_n));
                    fscale = fmaxw(1e-2f, 1.0f + x.alpha * (lever - x.l_avg) / x.l_avg);
                } else {
                    fscale = x.uniform_fs;
                }
            }
        } else {
            const int nb = R.wnbin[i];
            if (nb >= 0 && depth < 0.0f) {
                vec3 t_agg;
                float t_mag, t_eff;
                bool t_rel;
                bin_values(nb, t_agg, t_mag, t_rel, t_eff);
                if (normal_matching && t_rel)
                    final_n = normalize(quat_rotate(hydro_matching_rotation(vec3(R.tnormal[nb][0], R.tnormal[nb][1], R.tnormal[nb][2]), t_agg, t_mag), n));
                float t_anchor_depth = 0.0f;
                if (anchor_contact && t_rel) {  // the bin's deepest contact (its max-depth slot) sets the anchor's depth
                    if (R.tbl[nb][RED_DIRS] != 0ull && R.wpen[nb * RED_VALUES + RED_DIRS] > 0.0f) t_anchor_depth = R.wpen[nb * RED_VALUES + RED_DIRS];
                    if (R.agg[nb][6] > 1e-20f && t_anchor_depth > 0.0f) t_eff = t_eff + t_anchor_depth;
                }
                stiff = (t_mag > 1e-20f && t_eff > 0.0f) ? t_mag / t_eff : rec[7] * rec[8] / fmaxw(-depth, 1e-20f);
                if (moment_matching) {
                    const float v_unr = R.m_unr[nb], v_s1 = R.s1[nb], v_s2 = R.s2[nb], v_s0 = R.tdepth[nb] + t_anchor_depth;
                    if (v_unr > 1e-20f && v_s0 > 1e-20f && v_s1 > 1e-20f && t_mag > 1e-20f) {
                        const float v_target = v_unr * t_eff / t_mag;
                        if (v_target < v_s1) {
                            fscale = v_target / v_s1;
                        } else {
                            const float v_lavg = v_s1 / v_s0, v_var = v_s2 * v_s0 - v_s1 * v_s1;
                            float v_alpha = 0.0f;
                            if (v_var > 1e-20f) v_alpha = fminw(fmaxw((v_target - v_s1) * v_s1 / v_var, 0.0f), 1.0f);
                            vec3 v_anchor;
                            if (R.agg[nb][6] > 1e-20f) v_anchor = vec3(R.agg[nb][3], R.agg[nb][4], R.agg[nb][5]) / R.agg[nb][6];
                            const float v_lever = length(cross(vec3(rec[0], rec[1], rec[2]) - v_anchor, final_n));
                            if (v_lavg > 1e-20f) fscale = fmaxw(1e-2f, 1.0f + v_alpha * (v_lever - v_lavg) / v_lavg);
                        }
                    }
                }
            } else if (depth < 0.0f) {
                stiff = rec[7] * rec[8] / fmaxw(-depth, 1e-20f);
            } else {
                stiff = mca_k;
            }
        }
        if (!(depth < 0.0f)) { stiff = mca_k; fscale = 1.0f; }
        const vec3 pw = xform_point(p.X_b, vec3(rec[0], rec[1], rec[2])), nw = xform_vector(p.X_b, final_n);
        a.out_pair[slot] = pair_idx;
        a.out_key[slot] = R.tkey[e][sl];
        a.out_rank[slot] = rank;
        float* o = a.out_data + 9 * (size_t)slot;
        o[0] = pw.x; o[1] = pw.y; o[2] = pw.z; o[3] = nw.x; o[4] = nw.y; o[5] = nw.z;
        o[6] = depth; o[7] = 0.0f; o[8] = 0.0f;
        a.out_stiffness[slot] = stiff;
        if (a.out_friction) a.out_friction[slot] = fscale;
    }
    if (anchor_contact && t < RED_BINS && R.anchor[t]) {  // the entry's anchor contact: centre of pressure, aggregate force direction
        vec3 agg;
        float agg_mag, eff;
        bool reliable;
        bin_values(t, agg, agg_mag, reliable, eff);
        const EntryExport x = entry_export(t, agg_mag, eff);
        const int rank = R.ubase[t] + R.ucount[t], slot = R.row_base + rank;
        if (slot < a.capacity) {
            const vec3 pw = xform_point(p.X_b, x.anchor_pos), nw = xform_vector(p.X_b, normalize(agg));
            a.out_pair[slot] = pair_idx;
            a.out_key[slot] = 0x400000 | t;
            a.out_rank[slot] = rank;
            float* o = a.out_data + 9 * (size_t)slot;
            o[0] = pw.x; o[1] = pw.y; o[2] = pw.z; o[3] = nw.x; o[4] = nw.y; o[5] = nw.z;
            o[6] = -R.max_pen[t]; o[7] = 0.0f; o[8] = 0.0f;
            a.out_stiffness[slot] = x.shared;
            if (a.out_friction) a.out_friction[slot] = x.anchor_fs;
        }
    }
    __syncthreads();
    NT_HT(5, ht);
}

NT_DI void hydro_reduce_pair(const nt_hydro_args& a, const HydroPair& p, int pair_idx, HydroRedLds& R) {
    hydro_reduce_pair_impl<false>(a, p, pair_idx, R);
}
__device__ __attribute__((noinline)) void hydro_reduce_pair_extras(const nt_hydro_args& a, const HydroPair& p, int pair_idx, HydroRedLds& R) {
    hydro_reduce_pair_impl<true>(a, p, pair_idx, R);
}

template <bool REDUCE, bool EXTRAS = false>
__global__ void __launch_bounds__(256) hydro_pairs_kernel(nt_hydro_args a) {
    __shared__ HydroLds L;
    __shared__ typename std::conditional<REDUCE, HydroRedLds, int>::type R;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int pair_total = a.pair_world_prefix[a.worlds];
    for (int f = blockIdx.x; f < pair_total; f += gridDim.x) {
        int lo = 0, hi = a.worlds;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.pair_world_prefix[mid] <= f) lo = mid;
            else hi = mid;
        }
        const int pair_idx = lo * a.pairs_per_world + (f - a.pair_world_prefix[lo]);
        if (a.pair_kind[pair_idx] != 1) continue;
#ifdef NT_HYDRO_TIMING
        unsigned long long hk = clock64();
#endif
        HydroPair p;
        p.sa = a.pairs[2 * (size_t)pair_idx];
        p.sb = a.pairs[2 * (size_t)pair_idx + 1];
        int ia = a.shape_sdf_index[p.sa], ib = a.shape_sdf_index[p.sb];
        bool ok = ia >= 0 && ib >= 0 && ia < a.sdf_count && ib < a.sdf_count;
        if (ok) {
            p.A = a.sdf_table[ia];
            p.B = a.sdf_table[ib];
            ok = p.A.cx > 0 && p.B.cx > 0;
        }
        __syncthreads();
        if (t == 0) {
            L.pair_vox = 0;
            L.pair_face = 0;
            L.collide = 0;
            if constexpr (REDUCE) { R.n_chunk = 0; R.n_faces = 0; R.pair_kept = 0; R.overflow = 0; R.rows = 0; }
            if (ok) {  // SAT of the two SDF boxes (centred transforms), before the finer-is-B swap like the reference
                const xform Xa = load_xform(a.shape_transform + 7 * p.sa), Xb = load_xform(a.shape_transform + 7 * p.sb);
                const vec3 alo(p.A.box_lower[0], p.A.box_lower[1], p.A.box_lower[2]), ahi(p.A.box_upper[0], p.A.box_upper[1], p.A.box_upper[2]);
                const vec3 blo(p.B.box_lower[0], p.B.box_lower[1], p.B.box_lower[2]), bhi(p.B.box_upper[0], p.B.box_upper[1], p.B.box_upper[2]);
                const xform Ca = Xa * xform(0.5f * (alo + ahi), quat(0.0f, 0.0f, 0.0f, 1.0f));
                const xform Cb = Xb * xform(0.5f * (blo + bhi), quat(0.0f, 0.0f, 0.0f, 1.0f));
                L.collide = hydro_sat(Ca, 0.5f * (ahi - alo), Cb, 0.5f * (bhi - blo)) ? 1 : 0;
            }
        }
        __syncthreads();
        if (ok && p.B.voxel_radius > p.A.voxel_radius) {  // keep the finer SDF as shape B (:1362-1366)
            const int s_ = p.sa; p.sa = p.sb; p.sb = s_;
            const nt_sdf tmp = p.A; p.A = p.B; p.B = tmp;
        }
        if (t == 0 && a.out_pairs_normalized) {
            a.out_pairs_normalized[2 * (size_t)pair_idx] = p.sa;
            a.out_pairs_normalized[2 * (size_t)pair_idx + 1] = p.sb;
        }
        const int nbx = ok ? p.B.cx : 0, nby = ok ? p.B.cy : 0, nbz = ok ? p.B.cz : 0;
        const int nblocks = nbx * nby * nbz;
        if (ok && L.collide && nblocks <= HYDRO_MAX_BLOCKS) {
            p.gap_sum = a.shape_gap[p.sa] + a.shape_gap[p.sb];
            p.margin_a = a.shape_data[4 * p.sa + 3];
            p.margin_b = a.shape_data[4 * p.sb + 3];
            p.kh_a = a.shape_kh[p.sa];
            p.kh_b = a.shape_kh[p.sb];
            p.X_b = load_xform(a.shape_transform + 7 * p.sb);
            p.X_b2a = xform_inverse(load_xform(a.shape_transform + 7 * p.sa)) * p.X_b;
            const int sgs = p.B.subgrid_size;  // 8
            for (int b = t; b < nblocks; b += blockDim.x) {  // level 8: block b = (bz * nby + by) * nbx + bx
                const int bz = b / (nbx * nby), rem = b - bz * nbx * nby, by = rem / nbx, bx = rem - by * nbx;
                L.blk[b] = hydro_node_survives(p, bx * sgs, by * sgs, bz * sgs, sgs) ? 1 : 0;
            }
            __syncthreads();
            for (int b = 0; b < nblocks; ++b) {
                if (!L.blk[b]) continue;  // uniform
                const int bz = b / (nbx * nby), rem = b - bz * nbx * nby, by = rem / nbx, bx = rem - by * nbx;
                const int x0 = bx * sgs, y0 = by * sgs, z0 = bz * sgs;
                auto child = [](int code, int& cx, int& cy, int& cz) { cx = code & 1; cy = (code >> 1) & 1; cz = (code >> 2) & 1; };
                if (t < 8) {
                    int cx, cy, cz;
                    child(t, cx, cy, cz);
                    L.l4[t] = hydro_node_survives(p, x0 + 4 * cx, y0 + 4 * cy, z0 + 4 * cz, 4) ? 1 : 0;
                }
                __syncthreads();
                if (t < 64) {
                    int ax, ay, az, bx_, by_, bz_;
                    child(t >> 3, ax, ay, az);
                    child(t & 7, bx_, by_, bz_);
                    L.l2[t] = L.l4[t >> 3] && hydro_node_survives(p, x0 + 4 * ax + 2 * bx_, y0 + 4 * ay + 2 * by_, z0 + 4 * az + 2 * bz_, 2) ? 1 : 0;
                }
                __syncthreads();
                for (int j = t; j < 512; j += blockDim.x) {
                    int ax, ay, az, bx_, by_, bz_, cx, cy, cz;
                    child(j >> 6, ax, ay, az);
                    child((j >> 3) & 7, bx_, by_, bz_);
                    child(j & 7, cx, cy, cz);
                    L.l1[j] = L.l2[j >> 3] && hydro_node_survives(p, x0 + 4 * ax + 2 * bx_ + cx, y0 + 4 * ay + 2 * by_ + cy, z0 + 4 * az + 2 * bz_ + cz, 1) ? 1 : 0;
                }
                __syncthreads();
                // ordered compaction of the block's surviving voxels (two passes of 256 over the 512 flags)
                if (t == 0) L.n_vox = 0;
                __syncthreads();
                for (int j0 = 0; j0 < 512; j0 += 256) {
                    const int j = j0 + t;
                    const bool hit = t < 256 && L.l1[j] != 0;
                    const unsigned long long mask = __ballot(hit);
                    if (lane == 0) L.wsum[wave] = __popcll(mask);
                    __syncthreads();
                    int off = L.n_vox;
                    for (int k = 0; k < wave; ++k) off += L.wsum[k];
                    if (hit) L.vox[off + __popcll(mask & ((1ull << lane) - 1ull))] = j;
                    __syncthreads();
                    if (t == 0) L.n_vox += L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
                    __syncthreads();
                }
                const int n_vox = L.n_vox;
                for (int i0 = 0; i0 < n_vox; i0 += 256) {  // marching cubes, one lane per voxel, faces leave in voxel order
                    const int i = i0 + t;
                    HydroFace faces[5];
                    int face_id[5];
                    int kept = 0, vx = 0;
                    if (t < 256 && i < n_vox) {
                        const int j = L.vox[i];
                        int ax, ay, az, bx_, by_, bz_, cx, cy, cz;
                        child(j >> 6, ax, ay, az);
                        child((j >> 3) & 7, bx_, by_, bz_);
                        child(j & 7, cx, cy, cz);
                        kept = hydro_voxel_faces(a, p, x0 + 4 * ax + 2 * bx_ + cx, y0 + 4 * ay + 2 * by_ + cy, z0 + 4 * az + 2 * bz_ + cz, faces, face_id);
                        vx = 1;
                    }
                    (void)vx;
                    int x = kept;  // inclusive scan of the kept counts over the workgroup
                    for (int d = 1; d < 64; d <<= 1) {
                        const int y = __shfl_up(x, d);
                        if (lane >= d) x += y;
                    }
                    if (lane == 63) L.wsum[wave] = x;
                    __syncthreads();
                    int before = x - kept;
                    for (int k = 0; k < wave; ++k) before += L.wsum[k];
                    if constexpr (REDUCE) {
                        // buffered contacts of this voxel in buffer order: every face, or (pre_prune) the two strongest penetrating
                        // faces and the closest non-penetrating one (sdf_hydroelastic.py:2156-2312)
                        int sel[3] = {-1, -1, -1};
                        const bool prune = (a.reduce & 2) != 0;
                        if (prune) {
                            float s0 = 0.0f, s1 = 0.0f, best_np = 1.0e10f;
                            for (int k = 0; k < kept; ++k) {
                                const HydroFace& fc = faces[k];
                                if (fc.depth < 0.0f) {
                                    const float score = fc.area * fc.pressure;
                                    if (sel[0] < 0 || score > s0) { sel[1] = sel[0]; s1 = s0; sel[0] = k; s0 = score; }
                                    else if (sel[1] < 0 || score > s1) { sel[1] = k; s1 = score; }
                                } else if (fc.depth < best_np) {
                                    best_np = fc.depth;
                                    sel[2] = k;
                                }
                            }
                        }
                        const int nsel = prune ? (sel[0] >= 0) + (sel[1] >= 0) + (sel[2] >= 0) : kept;
                        int xs = nsel;  // inclusive scan of the buffered counts (contact ids follow the voxel order)
                        for (int d = 1; d < 64; d <<= 1) {
                            const int y = __shfl_up(xs, d);
                            if (lane >= d) xs += y;
                        }
                        __syncthreads();  // (wsum is read above by every lane)
                        if (lane == 63) L.wsum[wave] = xs;
                        __syncthreads();
                        int before_sel = xs - nsel;
                        for (int k = 0; k < wave; ++k) before_sel += L.wsum[k];
                        const int sel_total = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
                        __syncthreads();
                        // the chunk's block of face records
                        if (lane == 63) L.wsum[wave] = x;
                        __syncthreads();
                        if (t == 0) {
                            const int total = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
                            L.n_face = total;
                            L.base = total > 0 ? atomicAdd(a.face_count, total) : 0;
                            if (total > 0) {
                                if (R.n_chunk < HYDRO_CHUNK_CAP && L.base + total <= a.face_capacity) {
                                    R.chunk[R.n_chunk][0] = L.base;
                                    R.chunk[R.n_chunk][1] = total;
                                    R.cstart[R.n_chunk] = R.n_faces;
                                    R.idbase[R.n_chunk] = 0;  // (this kernel writes pair-absolute ids and voxel ranks)
                                    R.voxbase[R.n_chunk] = 0;
                                    R.n_faces += total;
                                    R.n_chunk += 1;
                                } else {
                                    R.overflow = 1;  // the pair loses these faces: reported through face_count[1]
                                }
                            }
                        }
                        __syncthreads();
                        for (int k = 0; k < kept; ++k) {
                            const int slot = L.base + before + k;
                            if (slot >= a.face_capacity) continue;
                            const HydroFace& fc = faces[k];
                            int cid = 0;
                            if (!prune) cid = L.pair_face + before + k + 1;
                            else {
                                int ord = 0;
                                for (int j = 0; j < 3; ++j) {
                                    if (sel[j] == k) cid = R.pair_kept + before_sel + ord + 1;
                                    ord += sel[j] >= 0 ? 1 : 0;
                                }
                            }
                            float* o = a.face_rec + HYDRO_FACE_WORDS * (size_t)slot;
                            o[0] = fc.pos.x; o[1] = fc.pos.y; o[2] = fc.pos.z;
                            o[3] = fc.normal.x; o[4] = fc.normal.y; o[5] = fc.normal.z;
                            o[6] = fc.depth; o[7] = fc.area; o[8] = fc.pressure;
                            int* oi = reinterpret_cast<int*>(o);
                            oi[9] = (L.pair_vox + i) * 5 + face_id[k];
                            oi[10] = (cid << 5) | red_get_slot(fc.normal);
                            oi[11] = 0;
                        }
                        __syncthreads();  // (no device-scope fence here: it would drop the L1 the SDF samples of the next voxels hit)
                        if (t == 0) { L.pair_face += L.n_face; R.pair_kept += sel_total; }
                        __syncthreads();
                        continue;
                    }
                    if (t == 0) {
                        const int total = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
                        L.n_face = total;
                        L.base = total > 0 ? atomicAdd(a.out_count, total) : 0;
                    }
                    __syncthreads();
                    for (int k = 0; k < kept; ++k) {
                        const int slot = L.base + before + k;
                        if (slot >= a.capacity) continue;
                        const HydroFace& fc = faces[k];
                        const vec3 nrm = red_decode_oct(fc.oct0, fc.oct1);
                        const vec3 pw = xform_point(p.X_b, fc.pos), nw = xform_vector(p.X_b, nrm);
                        a.out_pair[slot] = pair_idx;
                        a.out_key[slot] = (L.pair_vox + i) * 5 + face_id[k];
                        a.out_rank[slot] = L.pair_face + before + k;
                        float* o = a.out_data + 9 * (size_t)slot;
                        o[0] = pw.x; o[1] = pw.y; o[2] = pw.z; o[3] = nw.x; o[4] = nw.y; o[5] = nw.z;
                        o[6] = fc.depth; o[7] = 0.0f; o[8] = 0.0f;
                        a.out_stiffness[slot] = fc.stiff;
                    }
                    __syncthreads();
                    if (t == 0) L.pair_face += L.n_face;
                    __syncthreads();
                }
                if (t == 0) L.pair_vox += n_vox;
                __syncthreads();
            }
        }
        __syncthreads();
        NT_HT(0, hk);
        if constexpr (REDUCE) {
#ifdef NT_HYDRO_TIMING
            if (t == 0) { atomicAdd(&nt_hydro_timing[8], 1ull); if (L.pair_face > 0) { atomicAdd(&nt_hydro_timing[9], 1ull); atomicAdd(&nt_hydro_timing[10], (unsigned long long)R.n_chunk); } }
#endif
            if (L.pair_face > 0) {  // (uniform)
                __threadfence();  // the pair's face records, written by all lanes, are read back by other lanes below
                __syncthreads();
                if constexpr (EXTRAS) hydro_reduce_pair_extras(a, p, pair_idx, R);
                else hydro_reduce_pair(a, p, pair_idx, R);
            }
            if (t == 0) {
                if (R.overflow) atomicAdd(a.face_count + 1, 1);
                a.out_blk[2 * (size_t)pair_idx] = 0;
                a.out_blk[2 * (size_t)pair_idx + 1] = L.pair_face > 0 ? R.rows : 0;
            }
            __syncthreads();
        } else if (t == 0) {
            const int total = L.pair_face;
            a.out_blk[2 * (size_t)pair_idx] = 0;
            a.out_blk[2 * (size_t)pair_idx + 1] = total;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The reduce_contacts = True pipeline in DENSE STAGES (nt_hydro_args.stage_*; hydro_pairs_kernel<true> above gives a 256-lane
// workgroup to every pair and walks its blocks one after the other: in a pile a pair has a handful of surviving 8^3 blocks with a
// few dozen iso voxels each, so most lanes idle through ten workgroup barriers per block, and the two nt_sdf descriptors held per
// lane cost 247 VGPR).  Here the population of every stage is its own grid and the unit of work is a WAVE:
//   hydro_stage_blocks_kernel   wave per hydroelastic candidate pair: SAT, finer-SDF-is-B, the level-8 test of all of B's blocks
//                               (64 per round, ballot masks in LDS); the surviving blocks leave as one contiguous run of
//                               (pair, block) items in block order (one atomic per pair) -> stage_queue, stage_pair = (first item,
//                               items)
//   hydro_stage_faces_kernel    wave per item: levels 4 / 2 / 1 as ballot masks (8 lanes, 64 lanes, then the children of the
//                               surviving level-2 nodes dealt 64 at a time), voxels compacted in traversal order into the wave's
//                               LDS list; marching cubes 64 voxels per round: the corner samples one lane per (voxel, corner)
//                               through LDS, then one lane per voxel for the case and its faces (two passes: rank, write); a
//                               round's faces are one chunk of the face buffer (one atomic) recorded in stage_chunk = (first face,
//                               faces, buffered contacts, voxels); ids inside a record are relative to the chunk.  No workgroup
//                               barrier anywhere; the pair's descriptors are wave-uniform (scalar registers).
//   hydro_stage_reduce_kernel   workgroup per pair with items: lists the pair's chunks in (block, round) order -- the traversal order
//                               of the single kernel -- with the running totals that turn a record's block-relative voxel rank and contact id into the pair's, and
//                               runs the same hydro_reduce_pair on them.
// Faces, ids, order and rows are those of hydro_pairs_kernel<true>: the voxel order inside a block, the block order inside a pair
// and every arithmetic operation are unchanged; only who computes what moved.
// ---------------------------------------------------------------------------------------------------------------------------
#ifdef NT_EMULATED_GRID
#define HY_WAVE_SYNC() emu_wave_sync(64)
#else
#define HY_WAVE_SYNC() HY_WAVE_SYNC_HW()
#endif
#define HY_WAVE_SYNC_HW()                                      \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

NT_DI int hy_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// flat pair index f (world-major over the clamped per-world counts) -> world * pairs_per_world + k
NT_DI int hydro_pair_of_flat(const nt_hydro_args& a, int f) {
    int lo = 0, hi = a.worlds;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.pair_world_prefix[mid] <= f) lo = mid;
        else hi = mid;
    }
    return lo * a.pairs_per_world + (f - a.pair_world_prefix[lo]);
}
// the pair as the face pass sees it (descriptors, finer-is-B swap, relative transform); false: no SDF on one side
NT_DI bool hydro_pair_load(const nt_hydro_args& a, int pair_idx, HydroPair& p, bool with_sat, bool& collide) {
    p.sa = a.pairs[2 * (size_t)pair_idx];
    p.sb = a.pairs[2 * (size_t)pair_idx + 1];
    const int ia = a.shape_sdf_index[p.sa], ib = a.shape_sdf_index[p.sb];
    bool ok = ia >= 0 && ib >= 0 && ia < a.sdf_count && ib < a.sdf_count;
    if (ok) {
        p.A = a.sdf_table[ia];
        p.B = a.sdf_table[ib];
        ok = p.A.cx > 0 && p.B.cx > 0;
    }
    collide = false;
    if (!ok) return false;
    if (with_sat) {  // SAT of the two SDF boxes (centred transforms), before the finer-is-B swap like the reference
        const xform Xa = load_xform(a.shape_transform + 7 * p.sa), Xb = load_xform(a.shape_transform + 7 * p.sb);
        const vec3 alo(p.A.box_lower[0], p.A.box_lower[1], p.A.box_lower[2]), ahi(p.A.box_upper[0], p.A.box_upper[1], p.A.box_upper[2]);
        const vec3 blo(p.B.box_lower[0], p.B.box_lower[1], p.B.box_lower[2]), bhi(p.B.box_upper[0], p.B.box_upper[1], p.B.box_upper[2]);
        const xform Ca = Xa * xform(0.5f * (alo + ahi), quat(0.0f, 0.0f, 0.0f, 1.0f));
        const xform Cb = Xb * xform(0.5f * (blo + bhi), quat(0.0f, 0.0f, 0.0f, 1.0f));
        collide = hydro_sat(Ca, 0.5f * (ahi - alo), Cb, 0.5f * (bhi - blo));
    }
    if (p.B.voxel_radius > p.A.voxel_radius) {  // keep the finer SDF as shape B (:1362-1366)
        const int s_ = p.sa; p.sa = p.sb; p.sb = s_;
        const nt_sdf tmp = p.A; p.A = p.B; p.B = tmp;
    }
    p.gap_sum = a.shape_gap[p.sa] + a.shape_gap[p.sb];
    p.margin_a = a.shape_data[4 * p.sa + 3];
    p.margin_b = a.shape_data[4 * p.sb + 3];
    p.kh_a = a.shape_kh[p.sa];
    p.kh_b = a.shape_kh[p.sb];
    p.X_b = load_xform(a.shape_transform + 7 * p.sb);
    p.X_b2a = xform_inverse(load_xform(a.shape_transform + 7 * p.sa)) * p.X_b;
    return true;
}
NT_DI void hy_child(int code, int& cx, int& cy, int& cz) { cx = code & 1; cy = (code >> 1) & 1; cz = (code >> 2) & 1; }
// voxel j of a block (traversal code: child of 4 | child of 2 | voxel, three bits each) -> offset inside the block
NT_DI void hy_voxel(int j, int& x, int& y, int& z) {
    int ax, ay, az, bx, by, bz, cx, cy, cz;
    hy_child(j >> 6, ax, ay, az);
    hy_child((j >> 3) & 7, bx, by, bz);
    hy_child(j & 7, cx, cy, cz);
    x = 4 * ax + 2 * bx + cx; y = 4 * ay + 2 * by + cy; z = 4 * az + 2 * bz + cz;
}
constexpr int HY_STAGE_WAVES = 4;  // waves per workgroup of the wave-per-unit stages (independent: no workgroup barrier)
constexpr int HY_ROUND = 64;       // iso voxels per marching-cubes round = per chunk of face records
constexpr int HY_ITEMS = 8;                 // (pair, block) items of one pair a wave walks together (level 4: eight lanes per item)
constexpr int HY_VOX_CAP = 768;             // iso voxels of a batch held in LDS (>= 512: a batch of one item always fits)
constexpr int HY_FACE_CAP = 5 * HY_ROUND;   // candidate faces of a round
struct HyWaveFaces {
    int ox[HY_ITEMS], oy[HY_ITEMS], oz[HY_ITEMS];  // first voxel of every item's block
    unsigned char l4[64];                          // surviving level-4 nodes, traversal order: item << 3 | child
    unsigned short l2[64 * HY_ITEMS];              // surviving level-2 nodes: item << 6 | child of 4 << 3 | child of 2
    unsigned short vox[HY_VOX_CAP];                // iso voxels: item << 9 | (child of 4, child of 2, voxel)
    float es[8 * HY_ROUND], eo[8 * HY_ROUND];      // corner samples of the round's voxels: [voxel][corner]
    // the round's candidate faces (voxel lane << 3 | face), what the face lanes report to their voxel's lane, what the voxel lanes answer
    unsigned short face_list[HY_FACE_CAP];
    unsigned char fkeep[HY_FACE_CAP];
    float fdepth[HY_FACE_CAP], fscore[HY_FACE_CAP];
    int vfirst[64], vbefore[64], vbefore_sel[64], vsel[64];
};
#ifdef NT_HYDRO_FACES_WAVES  // measurement builds: cap the registers for this many waves per SIMD
#define NT_HYDRO_FACES_OCC __attribute__((amdgpu_waves_per_eu(NT_HYDRO_FACES_WAVES, NT_HYDRO_FACES_WAVES)))
#else
#define NT_HYDRO_FACES_OCC
#endif

// counters (stage_count): [0] queue items, [1] chunk records, [2] pairs / blocks lost to a full queue / chunk pool (-> overflow
// report), [3] -, [4] pairs with queued blocks (stage_active)
// A wave takes HY_BATCH consecutive pairs at a time and allocates for all of them with ONE atomic per counter: half a million
// same-address atomics (one per pair and counter) were most of this stage's 11 ms at C5's size (22 ns each, serialised in L2).
constexpr int HY_BATCH = 8;
struct HyWaveBatch {
    unsigned long long mask[HY_BATCH][HYDRO_MAX_BLOCKS / 64];
    int pair[HY_BATCH], total[HY_BATCH], rounds[HY_BATCH];
};
__global__ void __launch_bounds__(64 * HY_STAGE_WAVES) hydro_stage_blocks_kernel(nt_hydro_args a) {
    __shared__ HyWaveBatch W[HY_STAGE_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    HyWaveBatch& w = W[wave];
    const int pair_total = a.pair_world_prefix[a.worlds];
    for (int f0 = (blockIdx.x * HY_STAGE_WAVES + wave) * HY_BATCH; f0 < pair_total; f0 += gridDim.x * HY_STAGE_WAVES * HY_BATCH) {
        int sum = 0, active = 0;
        for (int s = 0; s < HY_BATCH; ++s) {  // ---- level 8 of every pair of the batch; survivors as ballot masks in LDS
            const int f = f0 + s;
            int pair_idx = -1, total = 0, rounds = 0;
            if (f < pair_total) {
                pair_idx = hy_uniform(hydro_pair_of_flat(a, f));
                if (a.pair_kind[pair_idx] != 1) pair_idx = -1;
            }
            if (pair_idx >= 0) {
                HydroPair p;
                bool collide;
                const bool ok = hydro_pair_load(a, pair_idx, p, true, collide);
                if (lane == 0 && a.out_pairs_normalized) {
                    a.out_pairs_normalized[2 * (size_t)pair_idx] = p.sa;
                    a.out_pairs_normalized[2 * (size_t)pair_idx + 1] = p.sb;
                }
                const int nbx = ok ? p.B.cx : 0, nby = ok ? p.B.cy : 0, nbz = ok ? p.B.cz : 0;
                const int nblocks = nbx * nby * nbz;
                if (ok && collide && nblocks <= HYDRO_MAX_BLOCKS) {
                    const int sgs = p.B.subgrid_size;  // 8
                    rounds = (nblocks + 63) >> 6;
                    for (int r = 0; r < rounds; ++r) {  // block b = (bz * nby + by) * nbx + bx
                        const int b = r * 64 + lane;
                        bool sv = false;
                        if (b < nblocks) {
                            const int bz = b / (nbx * nby), rem = b - bz * nbx * nby, by = rem / nbx, bx = rem - by * nbx;
                            sv = hydro_node_survives(p, bx * sgs, by * sgs, bz * sgs, sgs);
                        }
                        const unsigned long long m = __ballot(sv);
                        if (lane == 0) w.mask[s][r] = m;
                        total += __popcll(m);
                    }
                }
            }
            if (lane == 0) { w.pair[s] = pair_idx; w.total[s] = total; w.rounds[s] = rounds; }
            sum += total;
            active += total > 0 ? 1 : 0;
        }
        HY_WAVE_SYNC();
        int qb = 0, ab = 0;
        if (lane == 0) {
            if (sum > 0) qb = atomicAdd(a.stage_count, sum);
            if (active > 0) ab = atomicAdd(a.stage_count + 4, active);
        }
        qb = hy_uniform(__shfl(qb, 0));
        ab = hy_uniform(__shfl(ab, 0));
        for (int s = 0; s < HY_BATCH; ++s) {  // ---- the pairs' items: contiguous runs in block order
            const int pair_idx = w.pair[s];
            if (pair_idx < 0) continue;  // (uniform)
            int total = w.total[s];
            const int q0 = qb;
            qb += total;
            if (total > 0 && q0 + total > a.stage_queue_capacity) {  // does not fit the queue: the pair contributes nothing, reported
                if (lane == 0) atomicAdd(a.stage_count + 2, 1);
                if (lane == 0) a.stage_active[ab] = pair_idx;  // (keeps the active list dense; the pair has no items)
                ab += 1;
                total = 0;
                if (lane == 0) { a.stage_pair[2 * (size_t)pair_idx] = 0; a.stage_pair[2 * (size_t)pair_idx + 1] = 0; }
                continue;
            }
            int off = 0;
            for (int r = 0; r < w.rounds[s] && total > 0; ++r) {
                const unsigned long long m = w.mask[s][r];
                if ((m >> lane) & 1ull) {
                    const int q = q0 + off + __popcll(m & ((1ull << lane) - 1ull));
                    a.stage_queue[2 * (size_t)q] = pair_idx;
                    a.stage_queue[2 * (size_t)q + 1] = r * 64 + lane;
                }
                off += __popcll(m);
            }
            if (lane == 0) {
                a.stage_pair[2 * (size_t)pair_idx] = q0;
                a.stage_pair[2 * (size_t)pair_idx + 1] = total;
                if (total > 0) {  // the reduce stage only visits pairs that queued blocks
                    a.stage_active[ab] = pair_idx;
                } else {
                    a.out_blk[2 * (size_t)pair_idx] = 0;
                    a.out_blk[2 * (size_t)pair_idx + 1] = 0;
                }
            }
            ab += total > 0 ? 1 : 0;
        }
        HY_WAVE_SYNC();  // the next batch reuses the wave's LDS
    }
}

// One wave per BATCH of up to HY_ITEMS consecutive (pair, block) items of one pair (round 6).  The item-per-wave form of rounds 4-5
// walked every item as its own chain -- level 4 on 8 lanes, level 2, level 1, the corner samples of ~11 iso voxels, their ~2 faces each
// -- six to ten dependent sample round trips deep on a fifth of the lanes, at the two waves per SIMD the registers leave
// (profiles/r06Q..r06U: 66.8 ms per collide at C5's size; octree 13.4, corner samples + cases 14.6, faces 38.8 -> 23 with a lane per face).
// A batch shares the pair's descriptors (scalar registers) and its lanes sample neighbouring blocks of the same two SDFs, so every
// level is ONE dense list over the batch: level 4 on 8 lanes per item, the children of the survivors dealt 64 at a time, the iso voxels of
// all items compacted in (item, traversal) order, marching cubes 64 voxels per round whatever item they belong to, a lane per candidate
// face.  A round's faces are one chunk of the face buffer (one atomic; voxel ranks and contact ids inside a record are relative to
// the chunk); the batch's chunks are one run of stage_chunk (one atomic) booked on its first item -- the reduce stage lists a pair's
// chunks item by item, so the faces keep the order of the single kernel: (block, child of 4, child of 2, voxel, face).
// (The alternative measured in round 4, profiles/r04e_*: four blocks per wave on fixed 16-lane groups, 121 instead of 90 ms.)
constexpr int HY_MC_EDGE_BYTES = 2 * 2460;  // marching-cubes case tables: 820 triangles x 3 vertices x (corner, corner), newton_amd.mc_tables
__global__ void __launch_bounds__(64 * HY_STAGE_WAVES) NT_HYDRO_FACES_OCC hydro_stage_faces_kernel(nt_hydro_args a_) {
    __shared__ HyWaveFaces W[HY_STAGE_WAVES];
    // the case tables in LDS: every face fetches six table bytes behind its case's range -- dependent GLOBAL loads otherwise
    __shared__ int s_tri[257];
    __shared__ unsigned char s_edge[HY_MC_EDGE_BYTES];
    nt_hydro_args a = a_;
    {
        const int nbytes = 2 * a_.tri_range[256];
        if (nbytes <= HY_MC_EDGE_BYTES) {  // (else: an unexpected table, keep reading it from global memory)
            for (int i = threadIdx.x; i < 257; i += blockDim.x) s_tri[i] = a_.tri_range[i];
            for (int i = threadIdx.x; i < nbytes; i += blockDim.x) s_edge[i] = a_.flat_edge_verts[i];
            a.tri_range = s_tri;
            a.flat_edge_verts = s_edge;
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    HyWaveFaces& w = W[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
    int n_items = a.stage_count[0];
    n_items = n_items < a.stage_queue_capacity ? n_items : a.stage_queue_capacity;
    const bool prune = (a.reduce & 2) != 0;
    // a wave takes a CONTIGUOUS run of items: a pair's blocks are consecutive in the queue, so the pair's descriptors (three dependent
    // memory round trips + the relative transform) are fetched once per pair and run, and a batch is the next items of the same pair
    const int total_waves = gridDim.x * HY_STAGE_WAVES;
    const int per_wave = (n_items + total_waves - 1) / total_waves;
    const int q_begin = (blockIdx.x * HY_STAGE_WAVES + wave) * per_wave;
    const int q_end = q_begin + per_wave < n_items ? q_begin + per_wave : n_items;
    HydroPair p;
    int loaded_pair = -1;
    int q = q_begin;
    while (q < q_end) {
        // ---- the batch: items q .. q + nI - 1 (same pair), their block origins
        int my_pair = -1, my_block = 0;
        if (lane < HY_ITEMS && q + lane < q_end) {
            my_pair = a.stage_queue[2 * (size_t)(q + lane)];
            my_block = a.stage_queue[2 * (size_t)(q + lane) + 1];
        }
        const int pair_idx = hy_uniform(__shfl(my_pair, 0));
        if (pair_idx != loaded_pair) {  // (uniform)
            bool collide;
            hydro_pair_load(a, pair_idx, p, false, collide);
            loaded_pair = pair_idx;
        }
        const unsigned long long same = __ballot(my_pair == pair_idx);  // (bit 0 is set)
        int nI = hy_uniform(__builtin_ctzll(~same));
        const int q0 = q;
        {
            const int nbx = p.B.cx, nby = p.B.cy, sgs = p.B.subgrid_size;
            if (lane < nI) {
                const int bz = my_block / (nbx * nby), rem = my_block - bz * nbx * nby, by = rem / nbx, bx = rem - by * nbx;
                w.ox[lane] = bx * sgs; w.oy[lane] = by * sgs; w.oz[lane] = bz * sgs;
            }
        }
        HY_WAVE_SYNC();
        // ---- levels 4 and 2: the batch's surviving nodes as dense lists.  The voxel list bounds the batch: when the level-2 survivors
        // could leave more iso voxels than it holds, the batch is halved and walked again (one item always fits)
        int n2 = 0;
        for (;;) {
            bool s4 = false;
            if ((lane >> 3) < nI) {
                int cx, cy, cz;
                hy_child(lane & 7, cx, cy, cz);
                s4 = hydro_node_survives(p, w.ox[lane >> 3] + 4 * cx, w.oy[lane >> 3] + 4 * cy, w.oz[lane >> 3] + 4 * cz, 4);
            }
            const unsigned long long m4 = __ballot(s4);
            const int n4 = __popcll(m4);
            if (s4) w.l4[__popcll(m4 & lt)] = (unsigned char)lane;
            HY_WAVE_SYNC();
            n2 = 0;
            for (int t0 = 0; t0 < 8 * n4; t0 += 64) {
                const int t = t0 + lane;
                bool s2 = false;
                int code = 0;
                if (t < 8 * n4) {
                    const int e = (int)w.l4[t >> 3], it = e >> 3;
                    int ax, ay, az, bx_, by_, bz_;
                    hy_child(e & 7, ax, ay, az);
                    hy_child(t & 7, bx_, by_, bz_);
                    s2 = hydro_node_survives(p, w.ox[it] + 4 * ax + 2 * bx_, w.oy[it] + 4 * ay + 2 * by_, w.oz[it] + 4 * az + 2 * bz_, 2);
                    code = (e << 3) | (t & 7);
                }
                const unsigned long long m2 = __ballot(s2);
                if (s2) w.l2[n2 + __popcll(m2 & lt)] = (unsigned short)code;
                n2 += __popcll(m2);
            }
            HY_WAVE_SYNC();
            if (8 * n2 <= HY_VOX_CAP || nI == 1) break;
            nI = nI >> 1;
        }
        q = q0 + nI;
        // ---- level 1: the eight children of every surviving level-2 node, 64 tests per round, survivors in traversal order
        int n_vox = 0;
        for (int i0 = 0; i0 < 8 * n2; i0 += 64) {
            const int i = i0 + lane;
            bool s1 = false;
            int code = 0;
            if (i < 8 * n2) {
                const int e2 = (int)w.l2[i >> 3], it = e2 >> 6, j = ((e2 & 63) << 3) | (i & 7);
                int vx, vy, vz;
                hy_voxel(j, vx, vy, vz);
                s1 = hydro_node_survives(p, w.ox[it] + vx, w.oy[it] + vy, w.oz[it] + vz, 1);
                code = (it << 9) | j;
            }
            const unsigned long long m1 = __ballot(s1);
            if (s1) w.vox[n_vox + __popcll(m1 & lt)] = (unsigned short)code;
            n_vox += __popcll(m1);
        }
        HY_WAVE_SYNC();
#ifdef NT_HYDRO_SKIP_MC  // measurement builds only: the octree levels alone (no marching cubes, no faces: results are meaningless)
        n_vox = 0;
#endif
        const int nb = (n_vox + HY_ROUND - 1) / HY_ROUND;
        int chunk0 = 0;
        if (nb > 0) {
            int base = 0;
#ifdef NT_HYDRO_FAKE_ALLOC  // measurement builds only: no shared counters (regions overlap: results are meaningless)
            base = (int)(((long long)q0 * 2) % (a.stage_chunk_capacity - 16));
#else
            if (lane == 0) base = atomicAdd(a.stage_count + 1, nb);
#endif
            chunk0 = hy_uniform(__shfl(base, 0));
        }
        const bool chunks_ok = chunk0 + nb <= a.stage_chunk_capacity;
        if (lane == 0 && !chunks_ok) atomicAdd(a.stage_count + 2, 1);
        if (lane < nI) {  // the batch's chunks are booked on its first item
            a.stage_item[2 * (size_t)(q0 + lane)] = chunk0;
            a.stage_item[2 * (size_t)(q0 + lane) + 1] = (lane == 0 && chunks_ok) ? nb : 0;
        }
        if (!chunks_ok) continue;
        // ---- marching cubes, HY_ROUND voxels per round = one chunk of face records:
        // (a) the corner samples, one lane per (voxel, corner), through LDS;
        // (b) one lane per voxel: the case and its candidate faces (<= 5);
        // (c) one lane per candidate face, 64 per sub-round: the face lanes report (kept, depth, score) to their voxel's lane, which
        //     runs the reference's selection loop over them in face order and answers with the voxel's offsets; the face lanes write.
        //     The faces of the first sub-round stay in registers, the later ones (a round with more than 64 candidates) are evaluated
        //     again for the write.
        for (int k = 0; k < nb; ++k) {
            const int v0 = k * HY_ROUND;
            const int nv = (n_vox - v0) < HY_ROUND ? (n_vox - v0) : HY_ROUND;
            for (int t0 = 0; t0 < 8 * nv; t0 += 64) {
                const int t = t0 + lane;
                if (t < 8 * nv) {
                    const int code = (int)w.vox[v0 + (t >> 3)], it = code >> 9;
                    int vx, vy, vz;
                    hy_voxel(code & 511, vx, vy, vz);
                    float es, eo;
                    hydro_corner_sample(p, w.ox[it] + vx, w.oy[it] + vy, w.oz[it] + vz, t & 7, es, eo);
                    w.es[t] = es;
                    w.eo[t] = eo;
                }
            }
            HY_WAVE_SYNC();
            const bool mine = lane < nv;
            int nfv = 0;
            if (mine) {
                HydroCorners cn;
                float es8[8], eo8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { es8[i] = w.es[8 * lane + i]; eo8[i] = w.eo[8 * lane + i]; }
                hydro_voxel_classify(a, p, es8, eo8, cn);
                nfv = cn.nfaces;
            }
#ifdef NT_HYDRO_SKIP_FACES  // measurement builds only: corner samples + classification, no faces (results are meaningless)
            nfv = 0;
#endif
            int fx = nfv;
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(fx, d);
                if (lane >= d) fx += y;
            }
            const int F = hy_uniform(__shfl(fx, 63)), fb = fx - nfv;
            for (int fi = 0; fi < nfv; ++fi) w.face_list[fb + fi] = (unsigned short)((lane << 3) | fi);
            w.vfirst[lane] = fb;
            HY_WAVE_SYNC();
            // face lanes, first visit: evaluate, report
            auto face_of = [&](int fidx, HydroFace& fc) {
                const int code = (int)w.face_list[fidx];
                const int fv_ = code >> 3, ffi = code & 7;
                const int vcode = (int)w.vox[v0 + fv_], it = vcode >> 9;
                int ux, uy, uz;
                hy_voxel(vcode & 511, ux, uy, uz);
                float es8[8], eo8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { es8[i] = w.es[8 * fv_ + i]; eo8[i] = w.eo[8 * fv_ + i]; }
                HydroCorners fcn;
                hydro_voxel_classify(a, p, es8, eo8, fcn);
                return hydro_voxel_face(a, p, ux + w.ox[it], uy + w.oy[it], uz + w.oz[it], fcn, ffi, fc);
            };
            HydroFace fc0;
            for (int f0 = 0; f0 < F; f0 += 64) {
                const int fidx = f0 + lane;
                if (fidx < F) {
                    HydroFace fc;
                    const bool ok = face_of(fidx, fc);
                    w.fkeep[fidx] = ok ? 1 : 0;
                    w.fdepth[fidx] = ok ? fc.depth : 0.0f;
                    w.fscore[fidx] = ok ? fc.area * fc.pressure : 0.0f;
                    if (f0 == 0) fc0 = fc;
                }
            }
            HY_WAVE_SYNC();
            // voxel lanes: which faces stay, and (pre_prune) the two strongest penetrating faces + the closest non-penetrating one
            // (sdf_hydroelastic.py:2156-2312; indices are ordinals among the kept faces)
            int kept = 0, sel0 = -1, sel1 = -1, sel2 = -1;
            float sc0 = 0.0f, sc1 = 0.0f, best_np = 1.0e10f;
            for (int fi = 0; fi < nfv; ++fi) {
                if (!w.fkeep[fb + fi]) continue;
                if (prune) {
                    const float depth = w.fdepth[fb + fi];
                    if (depth < 0.0f) {
                        const float score = w.fscore[fb + fi];
                        if (sel0 < 0 || score > sc0) { sel1 = sel0; sc1 = sc0; sel0 = kept; sc0 = score; }
                        else if (sel1 < 0 || score > sc1) { sel1 = kept; sc1 = score; }
                    } else if (depth < best_np) {
                        best_np = depth;
                        sel2 = kept;
                    }
                }
                kept += 1;
            }
            const int nsel = prune ? (sel0 >= 0) + (sel1 >= 0) + (sel2 >= 0) : kept;
            int x = kept, xs = nsel;  // inclusive scans over the wave
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d), ys = __shfl_up(xs, d);
                if (lane >= d) { x += y; xs += ys; }
            }
            const int total = hy_uniform(__shfl(x, 63)), sel_total = hy_uniform(__shfl(xs, 63));
            w.vbefore[lane] = x - kept;
            w.vbefore_sel[lane] = xs - nsel;
            w.vsel[lane] = ((sel0 + 1) & 15) | (((sel1 + 1) & 15) << 4) | (((sel2 + 1) & 15) << 8);
            int base = 0;
#ifdef NT_HYDRO_FAKE_ALLOC
            base = (int)(((long long)(q0 + k) * 64) % (a.face_capacity - 512));
#else
            if (lane == 0 && total > 0) base = atomicAdd(a.face_count, total);
#endif
            base = hy_uniform(__shfl(base, 0));
            const bool fits = base + total <= a.face_capacity;
            if (lane == 0) {
                int* c = a.stage_chunk + 4 * (size_t)(chunk0 + k);
                c[0] = base;
                c[1] = fits ? total : -total;  // negative: the faces did not fit the buffer (counted, not stored)
                c[2] = sel_total;
                c[3] = nv;
            }
            HY_WAVE_SYNC();
            // face lanes, second visit: the kept faces into their records
            if (fits) {
                for (int f0 = 0; f0 < F; f0 += 64) {
                    const int fidx = f0 + lane;
                    if (fidx >= F || !w.fkeep[fidx]) continue;
                    HydroFace fc = fc0;
                    if (f0 != 0) face_of(fidx, fc);
                    const int code = (int)w.face_list[fidx];
                    const int fv_ = code >> 3, ffi = code & 7, first = w.vfirst[fv_];
                    int ord = 0;
                    for (int g = 0; g < ffi; ++g) ord += (int)w.fkeep[first + g];
                    const int vs_ = w.vsel[fv_], s0 = (vs_ & 15) - 1, s1 = ((vs_ >> 4) & 15) - 1, s2 = ((vs_ >> 8) & 15) - 1;
                    int cid = 0;  // contact id, relative to the chunk (the reduce stage adds what came before in the pair)
                    if (!prune) cid = w.vbefore[fv_] + ord + 1;
                    else {
                        const int bs = w.vbefore_sel[fv_];
                        int rank = 0;
                        if (s0 == ord) cid = bs + rank + 1;
                        rank += s0 >= 0 ? 1 : 0;
                        if (s1 == ord) cid = bs + rank + 1;
                        rank += s1 >= 0 ? 1 : 0;
                        if (s2 == ord) cid = bs + rank + 1;
                    }
                    float* o = a.face_rec + HYDRO_FACE_WORDS * (size_t)(base + w.vbefore[fv_] + ord);
                    o[0] = fc.pos.x; o[1] = fc.pos.y; o[2] = fc.pos.z;
                    o[3] = fc.normal.x; o[4] = fc.normal.y; o[5] = fc.normal.z;
                    o[6] = fc.depth; o[7] = fc.area; o[8] = fc.pressure;
                    int* oi = reinterpret_cast<int*>(o);
                    oi[9] = fv_ * 5 + ffi;  // voxel rank inside the CHUNK (rebased by the reduce stage)
                    oi[10] = (cid << 5) | red_get_slot(fc.normal);
                    oi[11] = 0;
                }
            }
            HY_WAVE_SYNC();  // (the next round overwrites the corner samples and the lists)
        }
        HY_WAVE_SYNC();  // the next batch reuses the wave's LDS lists
    }
}

constexpr int HY_REDUCE_THREADS = 128;  // same-box ABAB on hydro_bin: 256 lanes 545.5 ms per frame, 128 lanes 528.9, 64 lanes 556.7 (profiles/r06AH*_ab.txt)
template <bool EXTRAS>
__global__ void __launch_bounds__(256) hydro_stage_reduce_kernel(nt_hydro_args a) {
    __shared__ HydroRedLds R;
    int* const idbase = R.idbase;    // contact ids / voxels of the pair in front of each listed chunk
    int* const voxbase = R.voxbase;
    constexpr int HY_ITEM_TILE = 128;
    __shared__ int item_c0[HY_ITEM_TILE], item_nc[HY_ITEM_TILE + 1], n_raw;
    const int t = threadIdx.x;
    const int n_active = a.stage_count[4];
    const bool prune = (a.reduce & 2) != 0;
    for (int f = blockIdx.x; f < n_active; f += gridDim.x) {
        const int pair_idx = a.stage_active[f];
        const int q0 = a.stage_pair[2 * (size_t)pair_idx], nq = a.stage_pair[2 * (size_t)pair_idx + 1];
        __syncthreads();
#ifdef NT_POISON_LDS  // (tests/emu: every pair starts from garbage LDS, as on a CU that ran other workgroups before)
        if (t == 0) memset((void*)&R, 0xCD, sizeof(R));
        __syncthreads();
#endif
        // ---- the pair's chunk records, (block, round) order: items and records are fetched by all lanes (they sit wherever the
        // face stage's atomics put them), then lane 0 folds them into the chunk list on LDS copies
#ifdef NT_HYDRO_TIMING
        unsigned long long hk = clock64();
#endif
        if (t == 0) { R.n_chunk = 0; R.n_faces = 0; R.pair_kept = 0; R.overflow = 0; R.rows = 0; n_raw = 0; }
        __syncthreads();
        for (int q_tile = 0; q_tile < nq; q_tile += HY_ITEM_TILE) {
            const int m = nq - q_tile < HY_ITEM_TILE ? nq - q_tile : HY_ITEM_TILE;
            for (int k = t; k < m; k += blockDim.x) {  // (any workgroup size from one wave up)
                item_c0[k] = a.stage_item[2 * (size_t)(q0 + q_tile + k)];
                item_nc[k] = a.stage_item[2 * (size_t)(q0 + q_tile + k) + 1];
            }
            __syncthreads();
            if (t == 0) {
                int run = n_raw;
                for (int k = 0; k < m; ++k) { const int nc = item_nc[k]; item_nc[k] = run; run += nc; }  // -> first raw index of the item
                item_nc[m] = run;
                if (run > HYDRO_CHUNK_CAP) { R.overflow = 1; run = HYDRO_CHUNK_CAP; }
                n_raw = run;
            }
            __syncthreads();
            const int first = item_nc[0], last = n_raw;
            for (int g = first + t; g < last; g += blockDim.x) {
                int k = 0;
                while (k + 1 < m && item_nc[k + 1] <= g) ++k;
                const int* rec = a.stage_chunk + 4 * (size_t)(item_c0[k] + (g - item_nc[k]));
                R.chunk[g][0] = rec[0]; R.chunk[g][1] = rec[1];
                idbase[g] = rec[2]; voxbase[g] = rec[3];
            }
            __syncthreads();
        }
        if (t == 0) {
            int vox = 0, faces_all = 0, n = 0;
            for (int g = 0; g < n_raw; ++g) {  // in place: n <= g
                const int base = R.chunk[g][0], signed_total = R.chunk[g][1], kept = idbase[g], nv = voxbase[g];
                const int total = signed_total < 0 ? -signed_total : signed_total;
                if (total > 0) {
                    if (signed_total > 0) {
                        R.chunk[n][0] = base;
                        R.chunk[n][1] = total;
                        R.cstart[n] = R.n_faces;
                        idbase[n] = prune ? R.pair_kept : faces_all;
                        voxbase[n] = vox;
                        R.n_faces += total;
                        n += 1;
                    } else {
                        R.overflow = 1;  // the pair loses these faces: reported through face_count[1]
                    }
                }
                faces_all += total;
                R.pair_kept += kept;
                vox += nv;
            }
            R.n_chunk = n;
        }
        __syncthreads();
        NT_HT(6, hk);
        if (R.n_faces > 0) {  // (uniform)
            // (the records keep their block-relative voxel ranks and contact ids: hydro_reduce_pair adds R.voxbase / R.idbase where it
            // reads them -- a rewrite of the records plus the device-scope fence it needs was 29 % of this stage's cycles)
            HydroPair p;
            bool collide;
            hydro_pair_load(a, pair_idx, p, false, collide);
            NT_HT(7, hk);
#ifdef NT_HYDRO_TIMING
            if (t == 0) { atomicAdd(&nt_hydro_timing[9], 1ull); atomicAdd(&nt_hydro_timing[10], (unsigned long long)R.n_chunk); }
#endif
            if constexpr (EXTRAS) hydro_reduce_pair_extras(a, p, pair_idx, R);
            else hydro_reduce_pair(a, p, pair_idx, R);
        }
        if (t == 0) {
            if (R.overflow) atomicAdd(a.face_count + 1, 1);
            a.out_blk[2 * (size_t)pair_idx] = 0;
            a.out_blk[2 * (size_t)pair_idx + 1] = R.n_faces > 0 ? R.rows : 0;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

nt_status nt_hydro_collide(const nt_hydro_args* a, void* stream) {
    if (!a || a->pair_count < 0 || !a->out_count || !a->out_pair || !a->out_key || !a->out_shapes || !a->out_data || a->capacity <= 0 ||
        !a->tri_range || !a->flat_edge_verts || !a->shape_kh)
        return NT_ERR_INVALID_ARG;
    if (!(a->edge_clamp_min >= 0.0f && a->edge_clamp_min <= 0.5f)) return NT_ERR_INVALID_ARG;
    if (a->pair_count == 0) return NT_OK;
    int blocks = a->pair_count < 2048 ? a->pair_count : 2048;
    hipLaunchKernelGGL(hydro_collide_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

#ifdef NT_HYDRO_TIMING
nt_status nt_hydro_timing_read(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(nt_hydro_timing), sizeof(unsigned long long) * 16) == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}
#endif
nt_status nt_hydro_pairs(const nt_hydro_args* a, void* stream) {
    if (!a || !a->pairs || !a->pair_world_prefix || a->worlds <= 0 || a->pairs_per_world <= 0 || !a->pair_kind || !a->out_blk ||
        !a->out_rank || !a->out_stiffness || !a->out_count || !a->out_pair || !a->out_key || !a->out_data || a->capacity <= 0 ||
        !a->tri_range || !a->flat_edge_verts || !a->shape_kh || !a->shape_transform || !a->shape_data || !a->shape_gap ||
        !a->shape_sdf_index || !a->sdf_table)
        return NT_ERR_INVALID_ARG;
    if (!(a->edge_clamp_min >= 0.0f && a->edge_clamp_min <= 0.5f)) return NT_ERR_INVALID_ARG;
    const long long cap = (long long)a->worlds * a->pairs_per_world;
    const int blocks = cap < 16384 ? (int)cap : 16384;
    if (a->reduce & 1) {
        if (!a->shape_aabb_lower || !a->shape_aabb_upper || !a->shape_voxel_res || !a->face_count || !a->face_rec || a->face_capacity <= 0)
            return NT_ERR_INVALID_ARG;
#ifdef NT_EMULATED_GRID
        const int rblocks = blocks < NT_EMULATED_GRID ? blocks : NT_EMULATED_GRID;
#else
        const int rblocks = blocks;
#endif
        if (a->stage_count) {  // dense stages: wave per pair -> wave per (pair, block) -> workgroup per pair with faces
            if (!a->stage_queue || !a->stage_pair || !a->stage_item || !a->stage_chunk || !a->stage_active || a->stage_queue_capacity <= 0 ||
                a->stage_chunk_capacity <= 0)
                return NT_ERR_INVALID_ARG;
            hipStream_t st = (hipStream_t)stream;
            if (hipMemsetAsync(a->stage_count, 0, 8 * sizeof(int32_t), st) != hipSuccess) return NT_ERR_LAUNCH;
#ifdef NT_EMULATED_GRID
            const int wgrid = NT_EMULATED_GRID, igrid = NT_EMULATED_GRID;
#else
            const long long wb = (cap + HY_STAGE_WAVES * HY_BATCH - 1) / (HY_STAGE_WAVES * HY_BATCH);
            const int wgrid = (int)(wb < 16384 ? wb : 16384);
            const long long ib = ((long long)a->stage_queue_capacity + HY_STAGE_WAVES - 1) / HY_STAGE_WAVES;
            const int igrid = (int)(ib < 8192 ? ib : 8192);  // grid-stride over the items the first stage queued (count on the device)
#endif
            hipLaunchKernelGGL(hydro_stage_blocks_kernel, dim3(wgrid), dim3(64 * HY_STAGE_WAVES), 0, st, *a);
            hipLaunchKernelGGL(hydro_stage_faces_kernel, dim3(igrid), dim3(64 * HY_STAGE_WAVES), 0, st, *a);
            // lanes per pair of the reduce stage: a pile's pair has a few dozen faces (NT_HYDRO_REDUCE_THREADS: measurements)
            static const int rthreads = [] {
                const char* e = getenv("NT_HYDRO_REDUCE_THREADS");
                const int v = e ? atoi(e) : HY_REDUCE_THREADS;
                return v == 64 || v == 128 || v == 256 ? v : HY_REDUCE_THREADS;
            }();
            if (a->reduce & (8 | 16)) hipLaunchKernelGGL(hydro_stage_reduce_kernel<true>, dim3(rblocks), dim3(rthreads), 0, st, *a);
            else hipLaunchKernelGGL(hydro_stage_reduce_kernel<false>, dim3(rblocks), dim3(rthreads), 0, st, *a);
        } else if (a->reduce & (8 | 16)) hipLaunchKernelGGL((hydro_pairs_kernel<true, true>), dim3(rblocks), dim3(256), 0, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL((hydro_pairs_kernel<true, false>), dim3(rblocks), dim3(256), 0, (hipStream_t)stream, *a);
    } else {
        hipLaunchKernelGGL(hydro_pairs_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    }
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_sdf_sample(const nt_sdf* sdf, const float* points, int32_t n, float* out_dist, float* out_grad, void* stream) {
    if (!sdf || !points || n <= 0 || (!out_dist && !out_grad) || !sdf->coarse || !sdf->slots || sdf->cx <= 0) return NT_ERR_INVALID_ARG;
    if (sdf->quantization != 4 && sdf->quantization != 2 && sdf->quantization != 1) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sdf_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *sdf, points, n, out_dist, out_grad);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_sdf_sample_hw(const nt_sdf* sdf, const float* points, int32_t n, float* out_dist, void* stream) {
    if (!sdf || !points || n <= 0 || !out_dist || !sdf->coarse || !sdf->slots || sdf->cx <= 0) return NT_ERR_INVALID_ARG;
    if (sdf->quantization != 4 && sdf->quantization != 2 && sdf->quantization != 1) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sdf_sample_hw_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *sdf, points, n, out_dist);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_sdf_sample_voxels(const nt_sdf* sdf, const int32_t* ijk, int32_t n, float* out_dist, void* stream) {
    if (!sdf || !ijk || n <= 0 || !out_dist || !sdf->coarse || !sdf->slots || sdf->cx <= 0) return NT_ERR_INVALID_ARG;
    if (sdf->quantization != 4 && sdf->quantization != 2 && sdf->quantization != 1) return NT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sdf_sample_voxels_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *sdf, ijk, n, out_dist);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_mesh_sdf_collide(const nt_mesh_sdf_args* a, void* stream) {
    if (!a || a->pair_count < 0 || !a->out_count || !a->out_pair || !a->out_key || !a->out_data || a->capacity <= 0) return NT_ERR_INVALID_ARG;
    if (a->pair_count == 0) return NT_OK;
    int blocks = a->pair_count < 2048 ? a->pair_count : 2048;
    hipLaunchKernelGGL(mesh_sdf_collide_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_mesh_sdf_collide_reduced(const nt_mesh_sdf_args* a, const nt_contact_reduce_shapes* r, void* stream) {
    if (!a || !r || a->pair_count < 0 || !a->out_count || !a->out_pair || !a->out_key || !a->out_data || a->capacity <= 0)
        return NT_ERR_INVALID_ARG;
    if (!r->shape_aabb_lower || !r->shape_aabb_upper || !r->shape_voxel_res) return NT_ERR_INVALID_ARG;
    if (a->pair_count == 0 && !a->pair_world_prefix) return NT_OK;
    // workgroup size: meshes with few edges (C5's hulls have ~40) would leave most of 256 lanes idle in the edge loops; one
    // wave per pair then, and four times the pairs in flight per CU
    const int threads = r->threads == 64 || r->threads == 128 || r->threads == 256 ? r->threads : 256;
    int blocks = a->pair_count < 16384 ? a->pair_count : 16384;  // grid-stride over the pairs
    if (a->pair_world_prefix) {
        if (a->worlds <= 0 || a->pairs_per_world <= 0) return NT_ERR_INVALID_ARG;
        const long long cap = (long long)a->worlds * a->pairs_per_world;
        blocks = cap < 16384 ? (int)cap : 16384;
    }
    if (a->hit_count) {  // the staged variant: units -> cull -> resolve -> reduce, each dense over its own population
        if (!a->hit_pair || !a->hit_fp || !a->hit_rec || !a->hit_blk || a->hit_capacity <= 0 || !a->unit_ctx || !a->hit_stripes ||
            a->hit_stripe_count <= 0 || a->hit_capacity < a->hit_stripe_count)
            return NT_ERR_INVALID_ARG;
        // with out_blk the rows live in the survivor list's index space: the row arrays must span it
        if (a->out_blk && a->capacity < a->hit_capacity) return NT_ERR_INVALID_ARG;
        if (r->keep_all && !a->out_blk) return NT_ERR_UNSUPPORTED;
        nt_mesh_sdf_args k = *a;
        // grid-stride kernels: the grids only bound the parallelism (tests/emu runs every lane as an OS thread and caps them)
#ifdef NT_EMULATED_GRID
        const long long grid_cap = NT_EMULATED_GRID;
        if (blocks > NT_EMULATED_GRID) blocks = NT_EMULATED_GRID;
        if (k.hit_stripe_count > 3) k.hit_stripe_count = 3;
#else
        const long long grid_cap = 8192;
#endif
        const long long pairs = k.pair_world_prefix ? (long long)k.worlds * k.pairs_per_world : (long long)k.pair_count;
        auto grid = [&](long long work_items, int per_block) {
            const long long b = (work_items + per_block - 1) / per_block;
            return (unsigned)(b < 1 ? 1 : (b < grid_cap ? b : grid_cap));
        };
        // every stripe is shared by at least 8 of the cull kernel's waves (a small scene uses one stripe = the whole list);
        // the units kernel leaves the count in hit_count[2]
        const long long cull_waves = 4ll * grid(pairs, 4);
        if (k.hit_stripe_count > cull_waves / 8) k.hit_stripe_count = (int)(cull_waves / 8 > 1 ? cull_waves / 8 : 1);
        hipStream_t st = (hipStream_t)stream;
        if (hipMemsetAsync(k.hit_count, 0, 4 * sizeof(int32_t), st) != hipSuccess ||
            hipMemsetAsync(k.hit_stripes, 0, (size_t)a->hit_stripe_count * STRIPE_PAD * sizeof(int32_t), st) != hipSuccess)
            return NT_ERR_LAUNCH;
        hipLaunchKernelGGL(sdf_units_kernel, dim3(grid(pairs, 256)), dim3(256), 0, st, k, *r);
        hipLaunchKernelGGL(sdf_cull_kernel, dim3(grid(pairs, 4)), dim3(256), 0, st, k);  // one wave per runnable pair
        hipLaunchKernelGGL(sdf_resolve_kernel, dim3(grid(k.hit_capacity / k.hit_stripe_count, 256) * k.hit_stripe_count), dim3(256), 0, st, k);
        if (r->keep_all) hipLaunchKernelGGL(sdf_keep_all_kernel, dim3(blocks), dim3(64), 0, st, k);
        else hipLaunchKernelGGL(sdf_reduce_kernel, dim3(blocks), dim3(64), 0, st, k, *r);
        return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
    }
    if (r->keep_all) return NT_ERR_UNSUPPORTED;  // every contact: the staged variant with out_blk only
    hipLaunchKernelGGL(mesh_sdf_collide_reduced_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, *a, *r);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_contacts_reduce_list(const nt_contact_reduce_list* a, void* stream) {
    if (!a || a->segments < 0 || !a->segment_start || !a->out_count || !a->out_index || !a->out_normal || a->capacity <= 0)
        return NT_ERR_INVALID_ARG;
    if (a->segments == 0) return NT_OK;
    int blocks = a->segments < 4096 ? a->segments : 4096;
    hipLaunchKernelGGL(contacts_reduce_list_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
