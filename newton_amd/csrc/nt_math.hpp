// nt_math.hpp -- fp32 vector / quaternion / transform arithmetic for the gfx950 kernels.
// Semantics follow the warp-lang builtins Newton's kernels call (quat xyzw, transform = (p, q),
// zero-safe normalize) -- see DESIGN.md "arithmetic conventions".  Compiled with -ffp-contract=off:
// every +,-,*,/ and sqrt is a single correctly-rounded IEEE operation, in source order.
//
// The file can be included a second time under another namespace (`#define NT_MATH_NS ntf` + `#pragma clang fp contract(fast)`
// around the include): clang attaches the contraction permission to every fmul / fadd where it is WRITTEN, so code that may fuse
// a * b + c into v_fma_f32 needs its own copy of these helpers -- nt_kernels.hip does that for the XPBD projection phases
// (namespace ntf), while everything that decides pair sets, contact counts and contact geometry keeps namespace nt.
#include <hip/hip_runtime.h>

#ifndef NT_DI
#define NT_DI __device__ __forceinline__
#endif
#ifndef NT_MATH_NS
#define NT_MATH_NS nt
#define NT_MATH_NS_IS_DEFAULT
#endif
#if (defined(NT_MATH_NS_IS_DEFAULT) && !defined(NT_MATH_HPP_DEFAULT_DONE)) || !defined(NT_MATH_NS_IS_DEFAULT)
#ifdef NT_MATH_NS_IS_DEFAULT
#define NT_MATH_HPP_DEFAULT_DONE
#endif

namespace NT_MATH_NS {

struct vec3 {
    float x, y, z;
    NT_DI vec3() : x(0.f), y(0.f), z(0.f) {}
    NT_DI vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    NT_DI explicit vec3(float s) : x(s), y(s), z(s) {}
};
NT_DI float vget(const vec3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
NT_DI void vset(vec3& v, int i, float s) {  // value selects, not an if-chain of stores (see vsel)
    v.x = i == 0 ? s : v.x;
    v.y = i == 1 ? s : v.y;
    v.z = (i != 0 && i != 1) ? s : v.z;
}
NT_DI vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
NT_DI vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
NT_DI vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
NT_DI vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
NT_DI vec3 operator*(float s, vec3 a) { return vec3(a.x * s, a.y * s, a.z * s); }
NT_DI vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
NT_DI vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
NT_DI vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
NT_DI vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
NT_DI float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
NT_DI vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
NT_DI float length_sq(vec3 a) { return dot(a, a); }
NT_DI float length(vec3 a) { return sqrtf(dot(a, a)); }
NT_DI vec3 normalize(vec3 a) {
    float l = length(a);
    if (l > 0.0f) return a / l;
    return vec3();
}
NT_DI vec3 cw_mul(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
NT_DI float fminw(float a, float b) { return a < b ? a : b; }
NT_DI float fmaxw(float a, float b) { return a > b ? a : b; }
// component-wise select.  `cond ? a : b` on two vec3 LVALUES compiles to a select between their ADDRESSES plus a copy, which
// pins both operands (and any struct they live in) in scratch memory; this form stays in registers
NT_DI float fsel(bool k, float a, float b) { return k ? a : b; }
NT_DI vec3 vsel(bool k, vec3 a, vec3 b) { return vec3(k ? a.x : b.x, k ? a.y : b.y, k ? a.z : b.z); }
NT_DI vec3 vmin(vec3 a, vec3 b) { return vec3(fminw(a.x, b.x), fminw(a.y, b.y), fminw(a.z, b.z)); }
NT_DI vec3 vmax(vec3 a, vec3 b) { return vec3(fmaxw(a.x, b.x), fmaxw(a.y, b.y), fmaxw(a.z, b.z)); }
NT_DI vec3 vabs(vec3 a) { return vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
NT_DI float clampf(float x, float lo, float hi) { return fminw(fmaxw(x, lo), hi); }
NT_DI float signf(float x) { return x < 0.0f ? -1.0f : 1.0f; }
NT_DI float nonzero(float x) { return x != 0.0f ? 1.0f : 0.0f; }

struct quat {
    float x, y, z, w;
    NT_DI quat() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    NT_DI quat(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    NT_DI quat(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
};
NT_DI quat quat_identity() { return quat(0.f, 0.f, 0.f, 1.f); }
NT_DI quat operator+(quat a, quat b) { return quat(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
NT_DI quat operator*(quat a, float s) { return quat(a.x * s, a.y * s, a.z * s, a.w * s); }
NT_DI quat operator*(float s, quat a) { return quat(a.x * s, a.y * s, a.z * s, a.w * s); }
NT_DI quat operator*(quat a, quat b) {
    return quat(a.w * b.x + b.w * a.x + a.y * b.z - b.y * a.z,
                a.w * b.y + b.w * a.y + a.z * b.x - b.z * a.x,
                a.w * b.z + b.w * a.z + a.x * b.y - b.x * a.y,
                a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
NT_DI float dot(quat a, quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
NT_DI float length(quat a) { return sqrtf(dot(a, a)); }
NT_DI quat normalize(quat q) {
    float l = length(q);
    if (l > 0.0f) {
        float inv = 1.0f / l;
        return q * inv;
    }
    return quat(0.f, 0.f, 0.f, 1.f);
}
NT_DI quat quat_inverse(quat q) { return quat(-q.x, -q.y, -q.z, q.w); }
// wp.quat_rotate / wp.quat_rotate_inv, literal operation order -- in BOTH copies of this file.  (Round 6 measured the two-cross-product
// form v + 2 (w c + q x c), c = q x v, in the XPBD copy: 18 operations instead of ~30, equally accurate against the exact rotation
// (1.2e-6 vs 0.9e-6 at |v| ~ 3) -- but a few ulps AWAY from the checker's literal form in every rotation, which the reference's ramp
// line-up (cubes resting on single MPR contacts, 1.5 - 2.5 m from the origin) amplifies to 2.3e-5 m per step against 1.2e-5 with the
// literal form, for +1.6 % on the headline.  Parity margin kept; profiles/r06F_ramp_rotation_ab.txt.)
NT_DI vec3 quat_rotate(quat q, vec3 v) {
    vec3 qv(q.x, q.y, q.z);
    return v * (2.0f * q.w * q.w - 1.0f) + cross(qv, v) * q.w * 2.0f + qv * dot(qv, v) * 2.0f;
}
NT_DI vec3 quat_rotate_inv(quat q, vec3 v) {
    vec3 qv(q.x, q.y, q.z);
    return v * (2.0f * q.w * q.w - 1.0f) - cross(qv, v) * q.w * 2.0f + qv * dot(qv, v) * 2.0f;
}
// quat_rotate(q, e_x / e_y / e_z) with the multiplications by the unit vector's literal zeros and ones carried out by hand: every
// non-zero term keeps the literal form's operation and rounding ((2 w) w - 1 on the diagonal, (a b) 2 for the products, the
// cross term added before the dot term), so the result is bit-identical to the general form for finite inputs, up to the sign of
// an exactly-zero component (the literal form adds a +-0 first).  ~12 operations instead of ~30.  For consumers that take
// absolute values, squares or products of the components (AABB extents, plane normals, cylinder axes).
NT_DI vec3 quat_rotate_ex(quat q) {
    const float s = 2.0f * q.w * q.w - 1.0f;
    return vec3(s + (q.x * q.x) * 2.0f, (q.z * q.w) * 2.0f + (q.y * q.x) * 2.0f, (q.z * q.x) * 2.0f - (q.y * q.w) * 2.0f);
}
NT_DI vec3 quat_rotate_ey(quat q) {
    const float s = 2.0f * q.w * q.w - 1.0f;
    return vec3((q.x * q.y) * 2.0f - (q.z * q.w) * 2.0f, s + (q.y * q.y) * 2.0f, (q.x * q.w) * 2.0f + (q.z * q.y) * 2.0f);
}
NT_DI vec3 quat_rotate_ez(quat q) {
    const float s = 2.0f * q.w * q.w - 1.0f;
    return vec3((q.y * q.w) * 2.0f + (q.x * q.z) * 2.0f, (q.y * q.z) * 2.0f - (q.x * q.w) * 2.0f, s + (q.z * q.z) * 2.0f);
}
NT_DI quat quat_from_axis_angle(vec3 axis, float angle) {
    float half = angle * 0.5f;
    float w = cosf(half);
    float s = sinf(half);
    vec3 v = axis * s;
    return quat(v.x, v.y, v.z, w);
}

struct mat33 {
    float m00, m01, m02, m10, m11, m12, m20, m21, m22;
    NT_DI mat33() : m00(0.f), m01(0.f), m02(0.f), m10(0.f), m11(0.f), m12(0.f), m20(0.f), m21(0.f), m22(0.f) {}
    NT_DI mat33(float a00, float a01, float a02, float a10, float a11, float a12, float a20, float a21, float a22)
        : m00(a00), m01(a01), m02(a02), m10(a10), m11(a11), m12(a12), m20(a20), m21(a21), m22(a22) {}
};
NT_DI vec3 operator*(const mat33& A, vec3 v) {
    return vec3(A.m00 * v.x + A.m01 * v.y + A.m02 * v.z, A.m10 * v.x + A.m11 * v.y + A.m12 * v.z,
                A.m20 * v.x + A.m21 * v.y + A.m22 * v.z);
}
NT_DI mat33 operator*(float s, const mat33& A) {
    return mat33(A.m00 * s, A.m01 * s, A.m02 * s, A.m10 * s, A.m11 * s, A.m12 * s, A.m20 * s, A.m21 * s, A.m22 * s);
}
NT_DI mat33 transpose(const mat33& A) { return mat33(A.m00, A.m10, A.m20, A.m01, A.m11, A.m21, A.m02, A.m12, A.m22); }
NT_DI mat33 matrix_from_cols(vec3 c0, vec3 c1, vec3 c2) {
    return mat33(c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z);
}
NT_DI vec3 mat_col(const mat33& A, int j) {
    return j == 0 ? vec3(A.m00, A.m10, A.m20) : (j == 1 ? vec3(A.m01, A.m11, A.m21) : vec3(A.m02, A.m12, A.m22));
}
#if defined(NT_MATH_NS_IS_DEFAULT) || defined(NT_XPBD_IEEE)
NT_DI mat33 quat_to_matrix(quat q) {
    vec3 c0 = quat_rotate(q, vec3(1.f, 0.f, 0.f));
    vec3 c1 = quat_rotate(q, vec3(0.f, 1.f, 0.f));
    vec3 c2 = quat_rotate(q, vec3(0.f, 0.f, 1.f));
    return matrix_from_cols(c0, c1, c2);
}
#else
// (namespace ntf) the matrix entries written out -- three rotations of unit vectors multiply by literal zeros the compiler may not
// fold under IEEE rules (~80 operations for 21); diagonal in the literal form's (2 w^2 - 1) + 2 x^2
NT_DI mat33 quat_to_matrix(quat q) {
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float s = q.w * (q.w + q.w) - 1.0f;
    const float xy = q.x * y2, xz = q.x * z2, yz = q.y * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    return mat33(s + q.x * x2, xy - wz, xz + wy, xy + wz, s + q.y * y2, yz - wx, xz - wy, yz + wx, s + q.z * z2);
}
#endif

// wp.quat_from_matrix (trace / largest-diagonal branches, normalised result)
NT_DI quat quat_from_matrix(const mat33& m) {
    const float tr = m.m00 + m.m11 + m.m22;
    float x, y, z, w, h;
    if (tr >= 0.0f) {
        h = sqrtf(tr + 1.0f);
        w = 0.5f * h;
        h = 0.5f / h;
        x = (m.m21 - m.m12) * h;
        y = (m.m02 - m.m20) * h;
        z = (m.m10 - m.m01) * h;
    } else {
        int max_diag = 0;
        if (m.m11 > m.m00) max_diag = 1;
        if (m.m22 > (max_diag == 0 ? m.m00 : m.m11)) max_diag = 2;
        if (max_diag == 0) {
            h = sqrtf((m.m00 - (m.m11 + m.m22)) + 1.0f);
            x = 0.5f * h;
            h = 0.5f / h;
            y = (m.m01 + m.m10) * h;
            z = (m.m20 + m.m02) * h;
            w = (m.m21 - m.m12) * h;
        } else if (max_diag == 1) {
            h = sqrtf((m.m11 - (m.m22 + m.m00)) + 1.0f);
            y = 0.5f * h;
            h = 0.5f / h;
            z = (m.m12 + m.m21) * h;
            x = (m.m01 + m.m10) * h;
            w = (m.m02 - m.m20) * h;
        } else {
            h = sqrtf((m.m22 - (m.m00 + m.m11)) + 1.0f);
            z = 0.5f * h;
            h = 0.5f / h;
            x = (m.m20 + m.m02) * h;
            y = (m.m12 + m.m21) * h;
            w = (m.m10 - m.m01) * h;
        }
    }
    return normalize(quat(x, y, z, w));
}

// newton.math.quat_decompose (math/spatial.py:150-176): wrapped XYZ Euler coordinates (a0, a1, a2) with
// q = Rx(a0) Ry(a1) Rz(a2), the intrinsic chain compute_3d_rotational_dofs composes (the reference: wp.quat_to_euler(q, 2, 1, 0))
NT_DI vec3 quat_decompose(quat q) {
    mat33 R = quat_to_matrix(q);
    float sb = clampf(R.m02, -1.0f, 1.0f);
    float a, b = asinf(sb), c;
    if (fabsf(sb) < 0.9999999f) {
        a = atan2f(-R.m12, R.m22);
        c = atan2f(-R.m01, R.m00);
    } else {
        a = atan2f(R.m21, R.m11);
        c = 0.0f;
    }
    const float pi = 3.14159265358979323846f;
    if (a >= pi) a -= 2.0f * pi;
    if (c >= pi) c -= 2.0f * pi;
    return vec3(a, b, c);
}

struct xform {
    vec3 p;
    quat q;
    NT_DI xform() : p(), q(0.f, 0.f, 0.f, 1.f) {}
    NT_DI xform(vec3 p_, quat q_) : p(p_), q(q_) {}
};
NT_DI xform operator*(const xform& a, const xform& b) { return xform(quat_rotate(a.q, b.p) + a.p, a.q * b.q); }
NT_DI xform xform_inverse(const xform& t) {
    quat qi = quat_inverse(t.q);
    return xform(-quat_rotate(qi, t.p), qi);
}
NT_DI vec3 xform_point(const xform& t, vec3 x) { return t.p + quat_rotate(t.q, x); }
NT_DI vec3 xform_vector(const xform& t, vec3 x) { return quat_rotate(t.q, x); }

struct spatial {
    vec3 top, bottom;  // (linear, angular)
    NT_DI spatial() {}
    NT_DI spatial(vec3 a, vec3 b) : top(a), bottom(b) {}
};
NT_DI spatial operator+(const spatial& a, const spatial& b) { return spatial(a.top + b.top, a.bottom + b.bottom); }
NT_DI spatial operator-(const spatial& a, const spatial& b) { return spatial(a.top - b.top, a.bottom - b.bottom); }
NT_DI spatial operator*(const spatial& a, float s) { return spatial(a.top * s, a.bottom * s); }
NT_DI vec3 velocity_at_point(const spatial& qd, vec3 r) { return cross(qd.bottom, r) + qd.top; }

}  // namespace NT_MATH_NS
#endif
#ifdef NT_MATH_NS_IS_DEFAULT
#undef NT_MATH_NS_IS_DEFAULT
#undef NT_MATH_NS
#endif
