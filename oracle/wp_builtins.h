// TEST INFRASTRUCTURE ONLY -- CPU oracle. Never linked or imported by the product path
// (newton_amd/). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// Restatement of the warp-lang builtins Newton's hot-path kernels call
// (warp-lang pin 1.17.0.dev20260807, /root/reference/uv.lock:7379-7381 -- the source of
// warp is NOT vendored under /root/reference, so the exact fp32 operation order of these
// builtins is restated from warp's published native headers (vec.h / quat.h / mat.h /
// spatial.h semantics).  PARITY UNPINNED at bit level for this file; pinned at tolerance
// level by the reference's own known-answer tests (see tests/test_oracle_known_answers.py).
//
// Call sites in the reference that fix the *meaning* of each builtin:
//   wp.quat_rotate / quat_rotate_inv      newton/_src/solvers/solver.py:88-98
//   wp.transform_multiply / inverse / point / vector   newton/_src/sim/collide.py:184-193,334
//   wp.quat_to_matrix                     newton/_src/geometry/narrow_phase.py:737
//   wp.quat_from_axis_angle               newton/_src/sim/articulation.py:290
//   wp.normalize (zero-safe)              newton/_src/solvers/xpbd/kernels.py:2329
//   wp.velocity_at_point (via Newton swizzle)   newton/_src/math/spatial.py:53-78
#pragma once
#include <cmath>
#include <cstdint>

namespace wp {

struct vec3 {
    float x, y, z;
    vec3() : x(0.f), y(0.f), z(0.f) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length_sq(vec3 a) { return dot(a, a); }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
// wp.normalize: returns the zero vector when length == 0 (kEps = 0 guard in warp's vec.h)
inline vec3 normalize(vec3 a) {
    float l = length(a);
    if (l > 0.0f) return a / l;
    return vec3();
}
inline vec3 cw_mul(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float fminw(float a, float b) { return a < b ? a : b; }  // wp.min
inline float fmaxw(float a, float b) { return a > b ? a : b; }  // wp.max
inline vec3 vmin(vec3 a, vec3 b) { return vec3(fminw(a.x, b.x), fminw(a.y, b.y), fminw(a.z, b.z)); }
inline vec3 vmax(vec3 a, vec3 b) { return vec3(fmaxw(a.x, b.x), fmaxw(a.y, b.y), fmaxw(a.z, b.z)); }
inline vec3 vabs(vec3 a) { return vec3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }

inline float clampf(float x, float lo, float hi) { return fminw(fmaxw(x, lo), hi); }  // wp.clamp
inline float signf(float x) { return x < 0.0f ? -1.0f : 1.0f; }  // wp.sign: -1 if x<0 else 1
inline float nonzero(float x) { return x != 0.0f ? 1.0f : 0.0f; }  // wp.nonzero

struct quat {
    float x, y, z, w;
    quat() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    quat(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    quat(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float operator[](int i) const { return (&x)[i]; }
};
inline quat quat_identity() { return quat(0.f, 0.f, 0.f, 1.f); }
inline quat operator+(quat a, quat b) { return quat(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline quat operator*(quat a, float s) { return quat(a.x * s, a.y * s, a.z * s, a.w * s); }
inline quat operator*(float s, quat a) { return quat(a.x * s, a.y * s, a.z * s, a.w * s); }
// Hamilton product (warp quat.h mul(quat, quat))
inline quat operator*(quat a, quat b) {
    return quat(a.w * b.x + b.w * a.x + a.y * b.z - b.y * a.z,
                a.w * b.y + b.w * a.y + a.z * b.x - b.z * a.x,
                a.w * b.z + b.w * a.z + a.x * b.y - b.x * a.y,
                a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
inline float dot(quat a, quat b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float length(quat a) { return std::sqrt(dot(a, a)); }
inline quat normalize(quat q) {
    float l = length(q);
    if (l > 0.0f) {
        float inv = 1.0f / l;
        return q * inv;
    }
    return quat(0.f, 0.f, 0.f, 1.f);
}
inline quat quat_inverse(quat q) { return quat(-q.x, -q.y, -q.z, q.w); }
inline vec3 quat_rotate(quat q, vec3 v) {
    vec3 qv(q.x, q.y, q.z);
    return v * (2.0f * q.w * q.w - 1.0f) + cross(qv, v) * q.w * 2.0f + qv * dot(qv, v) * 2.0f;
}
inline vec3 quat_rotate_inv(quat q, vec3 v) {
    vec3 qv(q.x, q.y, q.z);
    return v * (2.0f * q.w * q.w - 1.0f) - cross(qv, v) * q.w * 2.0f + qv * dot(qv, v) * 2.0f;
}
inline quat quat_from_axis_angle(vec3 axis, float angle) {
    float half = angle * 0.5f;
    float w = std::cos(half);
    float s = std::sin(half);
    vec3 v = axis * s;
    return quat(v.x, v.y, v.z, w);
}

struct mat33 {
    float m[3][3];
    mat33() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = 0.f; }
    mat33(float a00, float a01, float a02, float a10, float a11, float a12, float a20, float a21, float a22) {
        m[0][0] = a00; m[0][1] = a01; m[0][2] = a02;
        m[1][0] = a10; m[1][1] = a11; m[1][2] = a12;
        m[2][0] = a20; m[2][1] = a21; m[2][2] = a22;
    }
    float operator()(int i, int j) const { return m[i][j]; }
};
inline vec3 operator*(const mat33& A, vec3 v) {
    return vec3(A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
                A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
                A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z);
}
inline mat33 operator*(float s, const mat33& A) {
    mat33 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = A.m[i][j] * s;
    return r;
}
inline mat33 operator*(const mat33& A, float s) { return s * A; }
inline mat33 transpose(const mat33& A) {
    return mat33(A.m[0][0], A.m[1][0], A.m[2][0], A.m[0][1], A.m[1][1], A.m[2][1], A.m[0][2], A.m[1][2], A.m[2][2]);
}
// wp.skew / mat33 @ mat33 (newton/_src/solvers/featherstone/kernels.py:95-96)
inline mat33 skew(vec3 v) { return mat33(0.0f, -v.z, v.y, v.z, 0.0f, -v.x, -v.y, v.x, 0.0f); }
inline mat33 operator*(const mat33& A, const mat33& B) {
    mat33 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float sum = 0.0f;
            for (int k = 0; k < 3; ++k) sum += A(i, k) * B(k, j);
            C.m[i][j] = sum;
        }
    return C;
}
inline mat33 matrix_from_cols(vec3 c0, vec3 c1, vec3 c2) {
    return mat33(c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z);
}
// wp.quat_to_matrix: columns are the rotated basis vectors
inline mat33 quat_to_matrix(quat q) {
    vec3 c0 = quat_rotate(q, vec3(1.f, 0.f, 0.f));
    vec3 c1 = quat_rotate(q, vec3(0.f, 1.f, 0.f));
    vec3 c2 = quat_rotate(q, vec3(0.f, 0.f, 1.f));
    return matrix_from_cols(c0, c1, c2);
}
// newton.math.quat_decompose (math/spatial.py:150-176): wrapped XYZ Euler coordinates (a0, a1, a2) with
// q = Rx(a0) * Ry(a1) * Rz(a2), i.e. the intrinsic X-Y'-Z'' chain that compute_3d_rotational_dofs composes.  The reference gets
// them from wp.quat_to_euler(q, 2, 1, 0) (un-vendored Warp builtin, Bernardes & Viollet 2022); restated here through the
// rotation matrix, same angles up to rounding away from the gimbal lock |a1| = pi/2.
inline vec3 quat_decompose(quat q) {
    mat33 R = quat_to_matrix(q);
    float sb = clampf(R(0, 2), -1.0f, 1.0f);
    float a, b = std::asin(sb), c;
    if (std::fabs(sb) < 0.9999999f) {
        a = std::atan2(-R(1, 2), R(2, 2));
        c = std::atan2(-R(0, 1), R(0, 0));
    } else {
        a = std::atan2(R(2, 1), R(1, 1));
        c = 0.0f;
    }
    const float pi = 3.14159265358979323846f;
    if (a >= pi) a -= 2.0f * pi;
    if (c >= pi) c -= 2.0f * pi;
    return vec3(a, b, c);
}
// wp.quat_from_matrix (warp/native/quat.h, un-vendored): trace / largest-diagonal branches, normalised result
inline quat quat_from_matrix(const mat33& m) {
    const float tr = m(0, 0) + m(1, 1) + m(2, 2);
    float x, y, z, w, h = 0.0f;
    if (tr >= 0.0f) {
        h = std::sqrt(tr + 1.0f);
        w = 0.5f * h;
        h = 0.5f / h;
        x = (m(2, 1) - m(1, 2)) * h;
        y = (m(0, 2) - m(2, 0)) * h;
        z = (m(1, 0) - m(0, 1)) * h;
    } else {
        int max_diag = 0;
        if (m(1, 1) > m(0, 0)) max_diag = 1;
        if (m(2, 2) > m(max_diag, max_diag)) max_diag = 2;
        if (max_diag == 0) {
            h = std::sqrt((m(0, 0) - (m(1, 1) + m(2, 2))) + 1.0f);
            x = 0.5f * h;
            h = 0.5f / h;
            y = (m(0, 1) + m(1, 0)) * h;
            z = (m(2, 0) + m(0, 2)) * h;
            w = (m(2, 1) - m(1, 2)) * h;
        } else if (max_diag == 1) {
            h = std::sqrt((m(1, 1) - (m(2, 2) + m(0, 0))) + 1.0f);
            y = 0.5f * h;
            h = 0.5f / h;
            z = (m(1, 2) + m(2, 1)) * h;
            x = (m(0, 1) + m(1, 0)) * h;
            w = (m(0, 2) - m(2, 0)) * h;
        } else {
            h = std::sqrt((m(2, 2) - (m(0, 0) + m(1, 1))) + 1.0f);
            z = 0.5f * h;
            h = 0.5f / h;
            x = (m(2, 0) + m(0, 2)) * h;
            y = (m(1, 2) + m(2, 1)) * h;
            w = (m(1, 0) - m(0, 1)) * h;
        }
    }
    return normalize(quat(x, y, z, w));
}

struct transform {
    vec3 p;
    quat q;
    transform() : p(), q(0.f, 0.f, 0.f, 1.f) {}
    transform(vec3 p_, quat q_) : p(p_), q(q_) {}
};
inline transform transform_identity() { return transform(); }
inline transform operator*(const transform& a, const transform& b) {
    return transform(quat_rotate(a.q, b.p) + a.p, a.q * b.q);
}
inline transform transform_inverse(const transform& t) {
    quat qi = quat_inverse(t.q);
    return transform(-quat_rotate(qi, t.p), qi);
}
inline vec3 transform_point(const transform& t, vec3 x) { return t.p + quat_rotate(t.q, x); }
inline vec3 transform_vector(const transform& t, vec3 x) { return quat_rotate(t.q, x); }

// Newton (linear, angular) spatial vector
struct spatial {
    vec3 top;     // linear / force
    vec3 bottom;  // angular / torque
    spatial() {}
    spatial(vec3 a, vec3 b) : top(a), bottom(b) {}
};
inline spatial operator+(spatial a, spatial b) { return spatial(a.top + b.top, a.bottom + b.bottom); }
inline spatial operator-(spatial a, spatial b) { return spatial(a.top - b.top, a.bottom - b.bottom); }
inline spatial operator*(spatial a, float s) { return spatial(a.top * s, a.bottom * s); }
inline spatial operator-(spatial a) { return spatial(-a.top, -a.bottom); }

// newton/_src/math/spatial.py:53-78  v_p = v + w x r  (Newton layout)
inline vec3 velocity_at_point(const spatial& qd, vec3 r) { return cross(qd.bottom, r) + qd.top; }

}  // namespace wp
