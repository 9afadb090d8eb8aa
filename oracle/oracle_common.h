// TEST INFRASTRUCTURE ONLY -- CPU oracle helpers (AoS load/store, enums). See oracle.h.
#pragma once
#include "oracle.h"
#include "wp_builtins.h"

namespace orc {
using namespace wp;

// newton/_src/sim/enums.py:183-212
enum JointType { PRISMATIC = 0, REVOLUTE = 1, BALL = 2, FIXED = 3, FREE = 4, DISTANCE = 5, D6 = 6, ROD = 7 };
// newton/_src/sim/enums.py:136-142
enum BodyFlags { BODY_DYNAMIC = 1, BODY_KINEMATIC = 2 };
// newton/_src/geometry/types.py:78-111
enum GeoType { GEO_NONE = 0, GEO_PLANE = 1, GEO_HFIELD = 2, GEO_SPHERE = 3, GEO_CAPSULE = 4, GEO_ELLIPSOID = 5,
               GEO_CYLINDER = 6, GEO_BOX = 7, GEO_MESH = 8, GEO_CONE = 9, GEO_CONVEX_MESH = 10 };
// newton/_src/core/types.py:71-72
static const float MAXVAL = 1e10f;

inline vec3 ld3(const float* a, int i) { return vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
inline void st3(float* a, int i, vec3 v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }
inline transform ldx(const float* a, int i) {
    const float* p = a + 7 * i;
    return transform(vec3(p[0], p[1], p[2]), quat(p[3], p[4], p[5], p[6]));
}
inline void stx(float* a, int i, const transform& t) {
    float* p = a + 7 * i;
    p[0] = t.p.x; p[1] = t.p.y; p[2] = t.p.z; p[3] = t.q.x; p[4] = t.q.y; p[5] = t.q.z; p[6] = t.q.w;
}
inline spatial lds(const float* a, int i) {
    const float* p = a + 6 * i;
    return spatial(vec3(p[0], p[1], p[2]), vec3(p[3], p[4], p[5]));
}
inline void sts(float* a, int i, const spatial& s) {
    float* p = a + 6 * i;
    p[0] = s.top.x; p[1] = s.top.y; p[2] = s.top.z; p[3] = s.bottom.x; p[4] = s.bottom.y; p[5] = s.bottom.z;
}
inline void adds(float* a, int i, const spatial& s) {  // wp.atomic_add on spatial_vector (serial)
    float* p = a + 6 * i;
    p[0] += s.top.x; p[1] += s.top.y; p[2] += s.top.z; p[3] += s.bottom.x; p[4] += s.bottom.y; p[5] += s.bottom.z;
}
inline void subs(float* a, int i, const spatial& s) {  // wp.atomic_sub
    float* p = a + 6 * i;
    p[0] -= s.top.x; p[1] -= s.top.y; p[2] -= s.top.z; p[3] -= s.bottom.x; p[4] -= s.bottom.y; p[5] -= s.bottom.z;
}
inline mat33 ldm(const float* a, int i) {
    const float* p = a + 9 * i;
    return mat33(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]);
}
inline float wmin(float a, float b) { return a < b ? a : b; }
inline float wmax(float a, float b) { return a > b ? a : b; }

}  // namespace orc
