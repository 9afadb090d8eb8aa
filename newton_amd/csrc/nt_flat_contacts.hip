// nt_flat_contacts.hip -- contacts that live outside the fixed-slot environment tiles (mesh-SDF / hydroelastic rows): the
// contact writer and the penalty-force evaluation on Newton's flat arrays, for gfx950.
//
// Reference behaviour:
//   write_contact        newton/_src/sim/collide.py:203-254 (+ _write_contact_at_index :165-201): ContactData -> the Contacts
//                        arrays (body-frame points and offsets, normal, margins); rows whose separation exceeds the pair's gap
//                        are not written
//   eval_body_contact    newton/_src/solvers/semi_implicit/kernels_contact.py:381-556: per-contact penalty force (stiffness,
//                        damping, Huber-smoothed friction, adhesion distance, optional per-contact stiffness / damping /
//                        friction scale), added to body_f of both bodies
//
// MI355X design: both are one-lane-per-row streaming kernels over AoS rows (36 B in, 92 B out for the writer; 100 B in for the
// force kernel): HBM-bound, coalesced by row, no LDS.  The writer keeps the row order of its input (the reduced mesh-SDF stage
// already emits deterministic-sort order) -- a rejected row stays in place as an inert (-1, -1) contact, which eval_body_contact
// skips (shape_a == shape_b), instead of being compacted away through an atomic counter as in the reference.  The force kernel
// accumulates with float atomics like the reference (wp.atomic_add / atomic_sub on body_f).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/newton_hip.h"
#include "nt_math.hpp"

using namespace nt;

namespace {

NT_DI xform load_xform7(const float* p) { return xform(vec3(p[0], p[1], p[2]), quat(p[3], p[4], p[5], p[6])); }
NT_DI void st3(float* p, vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
NT_DI vec3 ld3(const float* p) { return vec3(p[0], p[1], p[2]); }

__global__ void __launch_bounds__(256) contact_rows_write_kernel(nt_contact_rows a) {
    int n = a.row_count;
    if (a.row_count_device) { const int live = *a.row_count_device; n = live < n ? live : n; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {  // rows past the live count stay
        int sa = -1, sb = -1;                                                                    // untouched (stale)
        vec3 p0, p1, o0, o1, nrm;
        float m0 = 0.0f, m1 = 0.0f;
        {
            const int pr = a.row_pair[i];
            const int shape_a = a.pairs[2 * pr], shape_b = a.pairs[2 * pr + 1];
            const float* d = a.row_data + 9 * (size_t)i;
            const float dist = d[6], margin_a = d[7], margin_b = d[8];
            const float ra = 0.0f, rb = 0.0f;  // meshes: no effective radius (compute_effective_radius)
            const float total = ra + rb + margin_a + margin_b;
            const vec3 nab = normalize(ld3(d + 3));
            const vec3 center = ld3(d);
            const vec3 aw = center - nab * (0.5f * dist + ra);
            const vec3 bw = center + nab * (0.5f * dist + rb);
            const float sep = dot(bw - aw, nab) - total;
            if (!(sep > a.shape_gap[shape_a] + a.shape_gap[shape_b])) {
                sa = shape_a;
                sb = shape_b;
                const int ba = a.shape_body[sa], bb = a.shape_body[sb];
                const xform Xa = ba < 0 ? xform() : xform_inverse(load_xform7(a.body_q + 7 * (size_t)ba));
                const xform Xb = bb < 0 ? xform() : xform_inverse(load_xform7(a.body_q + 7 * (size_t)bb));
                m0 = ra + margin_a;
                m1 = rb + margin_b;
                p0 = xform_point(Xa, aw);
                p1 = xform_point(Xb, bw);
                o0 = xform_vector(Xa, m0 * nab);
                o1 = xform_vector(Xb, -m1 * nab);
                nrm = nab;
            }
        }
        a.out_shape0[i] = sa;
        a.out_shape1[i] = sb;
        st3(a.out_point0 + 3 * (size_t)i, p0);
        st3(a.out_point1 + 3 * (size_t)i, p1);
        st3(a.out_offset0 + 3 * (size_t)i, o0);
        st3(a.out_offset1 + 3 * (size_t)i, o1);
        st3(a.out_normal + 3 * (size_t)i, nrm);
        a.out_margin0[i] = m0;
        a.out_margin1[i] = m1;
    }
}

__global__ void __launch_bounds__(256) eval_body_contact_flat_kernel(nt_flat_contact_forces a) {
    int n = a.contact_max;
    if (a.contact_count) { const int live = *a.contact_count; n = live < n ? live : n; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int shape_a = a.shape0[i], shape_b = a.shape1[i];
        if (shape_a == shape_b) continue;
        float ke = 0.0f, kd = 0.0f, kf = 0.0f, ka = 0.0f, mu = 0.0f;
        int mat_nonzero = 0, body_a = -1, body_b = -1;
        if (shape_a >= 0) {
            mat_nonzero += 1;
            ke += a.shape_ke[shape_a]; kd += a.shape_kd[shape_a]; kf += a.shape_kf[shape_a];
            ka += a.shape_ka[shape_a]; mu += a.shape_mu[shape_a];
            body_a = a.shape_body[shape_a];
        }
        if (shape_b >= 0) {
            mat_nonzero += 1;
            ke += a.shape_ke[shape_b]; kd += a.shape_kd[shape_b]; kf += a.shape_kf[shape_b];
            ka += a.shape_ka[shape_b]; mu += a.shape_mu[shape_b];
            body_b = a.shape_body[shape_b];
        }
        if (mat_nonzero > 0) {
            ke /= float(mat_nonzero); kd /= float(mat_nonzero); kf /= float(mat_nonzero);
            ka /= float(mat_nonzero); mu /= float(mat_nonzero);
        }
        if (a.contact_stiffness) {  // per-contact overrides (kernels_contact.py:452-459)
            const float cke = a.contact_stiffness[i], ckd = a.contact_damping[i], cmu = a.contact_friction_scale[i];
            ke = cke > 0.0f ? cke : ke;
            kd = ckd > 0.0f ? ckd : kd;
            mu = cmu > 0.0f ? mu * cmu : mu;
        }
        const vec3 nrm = -ld3(a.normal + 3 * (size_t)i);
        vec3 bx_a = ld3(a.point0 + 3 * (size_t)i), bx_b = ld3(a.point1 + 3 * (size_t)i);
        const float margin_a = a.margin0[i], margin_b = a.margin1[i];
        vec3 r_a(0.0f), r_b(0.0f);
        if (body_a >= 0) {
            const xform X = load_xform7(a.body_q + 7 * (size_t)body_a);
            bx_a = xform_point(X, bx_a) - margin_a * nrm;
            r_a = bx_a - xform_point(X, ld3(a.body_com + 3 * (size_t)body_a));
        }
        if (body_b >= 0) {
            const xform X = load_xform7(a.body_q + 7 * (size_t)body_b);
            bx_b = xform_point(X, bx_b) + margin_b * nrm;
            r_b = bx_b - xform_point(X, ld3(a.body_com + 3 * (size_t)body_b));
        }
        const float d = dot(nrm, bx_a - bx_b);
        if (d >= ka) continue;
        vec3 bv_a(0.0f), bv_b(0.0f);
        if (body_a >= 0) {  // spatial_vector = (linear, angular)
            const float* q = a.body_qd + 6 * (size_t)body_a;
            bv_a = ld3(q) + cross(ld3(q + 3), r_a);
        }
        if (body_b >= 0) {
            const float* q = a.body_qd + 6 * (size_t)body_b;
            bv_b = ld3(q) + cross(ld3(q + 3), r_b);
        }
        const vec3 v = bv_a - bv_b;
        const float vn = dot(nrm, v);
        const vec3 vt = v - nrm * vn;
        const float fn = d * ke;
        const float fd = fminw(vn, 0.0f) * kd * (d < 0.0f ? 1.0f : 0.0f);
        vec3 ft(0.0f);
        if (d < 0.0f) {
            const float delta = a.friction_smoothing;
            const float a2 = dot(vt, vt);  // wp.norm_huber
            const float vs = a2 <= delta * delta ? 0.5f * a2 : delta * (sqrtf(a2) - 0.5f * delta);
            if (vs > 0.0f) {
                const vec3 fr = vt / vs;
                ft = fr * fminw(kf * vs, -mu * (fn + fd));
            }
        }
        const vec3 f_total = nrm * (fn + fd) + ft;
        if (body_a >= 0) {
            const vec3 tq = cross(r_a, f_total);
            float* f = a.body_f + 6 * (size_t)body_a;
            atomicAdd(f + 0, -f_total.x); atomicAdd(f + 1, -f_total.y); atomicAdd(f + 2, -f_total.z);
            atomicAdd(f + 3, -tq.x); atomicAdd(f + 4, -tq.y); atomicAdd(f + 5, -tq.z);
        }
        if (body_b >= 0) {
            const vec3 tq = cross(r_b, f_total);
            float* f = a.body_f + 6 * (size_t)body_b;
            atomicAdd(f + 0, f_total.x); atomicAdd(f + 1, f_total.y); atomicAdd(f + 2, f_total.z);
            atomicAdd(f + 3, tq.x); atomicAdd(f + 4, tq.y); atomicAdd(f + 5, tq.z);
        }
    }
}

}  // namespace

extern "C" {

nt_status nt_contact_rows_write(const nt_contact_rows* a, void* stream) {
    if (!a || a->row_count < 0 || !a->row_pair || !a->pairs || !a->row_data || !a->body_q || !a->shape_body || !a->shape_gap ||
        !a->out_shape0 || !a->out_shape1 || !a->out_point0 || !a->out_point1 || !a->out_offset0 || !a->out_offset1 ||
        !a->out_normal || !a->out_margin0 || !a->out_margin1)
        return NT_ERR_INVALID_ARG;
    if (a->row_count == 0) return NT_OK;
    int blocks = (a->row_count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(contact_rows_write_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

nt_status nt_eval_body_contact_flat(const nt_flat_contact_forces* a, void* stream) {
    if (!a || a->contact_max < 0 || !a->body_q || !a->body_qd || !a->body_com || !a->shape_ke || !a->shape_kd || !a->shape_kf ||
        !a->shape_ka || !a->shape_mu || !a->shape_body || !a->point0 || !a->point1 || !a->normal || !a->shape0 || !a->shape1 ||
        !a->margin0 || !a->margin1 || !a->body_f)
        return NT_ERR_INVALID_ARG;
    if (a->contact_stiffness && (!a->contact_damping || !a->contact_friction_scale)) return NT_ERR_INVALID_ARG;
    if (a->contact_max == 0) return NT_OK;
    int blocks = (a->contact_max + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(eval_body_contact_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? NT_OK : NT_ERR_LAUNCH;
}

}  // extern "C"
