// TEST INFRASTRUCTURE ONLY -- CPU oracle: convex pair path (MPR -> GJK fallback -> multi-contact manifold).
// Literal restatement of
//   support_map / _support_map_box / create_shape_support_function(center_ties)   newton/_src/geometry/support_function.py:131-447
//   create_shape_center_function (primitive shapes: center = local origin)         support_function.py:537-598
//   create_support_map_function / solve_mpr_core                                   newton/_src/geometry/mpr.py:63-403
//   solve_closest_distance_core (GJK) + closest_segment/triangle/tetrahedron       newton/_src/geometry/simplex_solver.py:44-469
//   solve_convex_multi_contact                                                     newton/_src/geometry/collision_convex.py:110-232
//   build_manifold + polygon clipping helpers                                      newton/_src/geometry/multicontact.py:28-956
//   orthonormal_basis                                                              newton/_src/math/__init__.py:232-275
//   compute_gjk_mpr_contacts / find_contacts / post_process_axial_on_discrete_contact   newton/_src/geometry/collision_core.py:39-50,173-278,325-792
//   narrow_phase_kernel_gjk_mpr (pair set-up)                                      newton/_src/geometry/narrow_phase.py:1040-1216
//   write_contact (output_index = -1: gap test + append)                           newton/_src/sim/collide.py:206-254
// Scope: BOX, SPHERE, CAPSULE, ELLIPSOID, CYLINDER (straight), CONE; infinite-plane proxies, convex meshes and
// triangles are not restated (pairs with them produce no contacts here).
#include <algorithm>
#include <initializer_list>
#include <vector>

#include "oracle_common.h"

namespace orc {
void write_contact_at(const o_model* m, const float* body_q, o_contacts* ct, int index, int shape_a, int shape_b, vec3 center,
                      vec3 normal_in, float distance, float radius_eff_a, float radius_eff_b, float margin_a, float margin_b);
}
using namespace orc;

namespace {

constexpr int GEO_TRIANGLE = 1000;  // GeoTypeEx.TRIANGLE (support_function.py:58): vertex A at the origin, B - A in scale, C - A in aux
constexpr int GEO_TRIANGLE_PRISM = 1001;  // a heightfield cell's triangle, extruded 1 m along -Z of the heightfield frame (:59,193-200)
struct Geom {
    int type;
    vec3 scale;
    const float* points = nullptr;  // CONVEX_MESH: vertex slice [count][3] (unscaled)
    int count = 0;
    vec3 center;                    // interior point used to seed MPR / GJK (collision_core.py:690, narrow_phase.py:1102-1105)
    vec3 aux;                       // TRIANGLE: C - A (GenericShapeData.auxiliary)
};
struct vec2 {
    float x, y;
    vec2() : x(0.f), y(0.f) {}
    vec2(float a, float b) : x(a), y(b) {}
};
inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator*(float s, vec2 a) { return vec2(a.x * s, a.y * s); }
inline float length_sq2(vec2 a) { return a.x * a.x + a.y * a.y; }

struct Vert {
    vec3 B, BtoA;
};
inline vec3 vert_a(const Vert& v) { return v.B + v.BtoA; }

// support_function.py:44-53 (CPU branch)
inline float support_rsqrt_rn(float v) { return 1.0f / std::sqrt(v); }

// support_function.py:120-128
vec3 support_map_box(const Geom& g, vec3 d) {
    float ds = fmaxw(std::fabs(d.x), fmaxw(std::fabs(d.y), std::fabs(d.z)));
    float threshold = 1.0e-10f * ds;
    float sx = d.x >= -threshold ? 1.0f : -1.0f;
    float sy = d.y >= -threshold ? 1.0f : -1.0f;
    float sz = d.z >= -threshold ? 1.0f : -1.0f;
    return vec3(sx * g.scale.x, sy * g.scale.y, sz * g.scale.z);
}

// support_function.py:131-350
vec3 support_map(const Geom& g, vec3 direction) {
    const float eps = 1.0e-12f;
    vec3 result(0.0f);
    if (g.type == GEO_TRIANGLE || g.type == GEO_TRIANGLE_PRISM) {
        // support_function.py:174-200: the vertex furthest along the direction; ties prefer a, then b; the prism adds its 1 m depth
        vec3 tri_a(0.0f), tri_b = g.scale, tri_c = g.aux;
        float dot_a = dot(tri_a, direction), dot_b = dot(tri_b, direction), dot_c = dot(tri_c, direction);
        if (dot_a >= dot_b && dot_a >= dot_c) result = tri_a;
        else if (dot_b >= dot_c) result = tri_b;
        else result = tri_c;
        if (g.type == GEO_TRIANGLE_PRISM && direction.z < 0.0f) result = result + vec3(0.0f, 0.0f, -1.0f);
        return result;
    }
    if (g.type == GEO_PLANE) {
        // support_function.py:334-345: finite rectangle in XY (half-width scale.x, half-length scale.y), normal +Z
        float sx = direction[0] >= 0.0f ? 1.0f : -1.0f;
        float sy = direction[1] >= 0.0f ? 1.0f : -1.0f;
        return vec3(sx * g.scale.x, sy * g.scale.y, 0.0f);
    }
    if (g.type == GEO_CONVEX_MESH) {
        // support_function.py:152-171: furthest vertex; ties keep the first one
        vec3 scaled_dir = cw_mul(direction, g.scale);
        float max_dot = -1.0e10f;
        int best_idx = 0;
        for (int i = 0; i < g.count; ++i) {
            float dot_val = dot(ld3(g.points, i), scaled_dir);
            if (dot_val > max_dot) {
                max_dot = dot_val;
                best_idx = i;
            }
        }
        result = g.count > 0 ? cw_mul(ld3(g.points, best_idx), g.scale) : vec3(0.0f);
    } else if (g.type == GEO_BOX) {
        result = support_map_box(g, direction);
    } else if (g.type == GEO_SPHERE) {
        float radius = g.scale.x;
        float l2 = length_sq(direction);
        vec3 n = l2 > eps ? direction * support_rsqrt_rn(l2) : vec3(1.0f, 0.0f, 0.0f);
        result = n * radius;
    } else if (g.type == GEO_CAPSULE) {
        float radius = g.scale.x, half_height = g.scale.y;
        float l2 = length_sq(direction);
        vec3 n = l2 > eps ? direction * support_rsqrt_rn(l2) : vec3(1.0f, 0.0f, 0.0f);
        result = n * radius;
        if (direction.z >= 0.0f) result = result + vec3(0.0f, 0.0f, half_height);
        else result = result + vec3(0.0f, 0.0f, -half_height);
    } else if (g.type == GEO_ELLIPSOID) {
        float a = g.scale.x, b = g.scale.y, c = g.scale.z;
        float l2 = length_sq(direction);
        if (l2 > eps) {
            float adx = a * direction.x, bdy = b * direction.y, cdz = c * direction.z;
            float denom_sq = adx * adx + bdy * bdy + cdz * cdz;
            if (denom_sq > eps) {
                float inv_denom = support_rsqrt_rn(denom_sq);
                result = vec3((a * a) * direction.x * inv_denom, (b * b) * direction.y * inv_denom, (c * c) * direction.z * inv_denom);
            } else {
                result = vec3(a, 0.0f, 0.0f);
            }
        } else {
            result = vec3(a, 0.0f, 0.0f);
        }
    } else if (g.type == GEO_CYLINDER) {
        float radius = g.scale.x, half_height = g.scale.y, barrel_radius = g.scale.z;
        vec3 dir_xy(direction.x, direction.y, 0.0f);
        float l2 = length_sq(dir_xy);
        if (barrel_radius == 0.0f) {
            vec3 lateral;
            if (l2 > eps) {
                vec3 n_xy = dir_xy * support_rsqrt_rn(l2);
                lateral = vec3(n_xy.x * radius, n_xy.y * radius, 0.0f);
            } else {
                lateral = vec3(radius, 0.0f, 0.0f);
            }
            if (direction.z > 0.0f) result = vec3(lateral.x, lateral.y, half_height);
            else if (direction.z < 0.0f) result = vec3(lateral.x, lateral.y, -half_height);
            else result = lateral;
        } else {
            // barrel cylinder: the side profile is a circular arc of radius barrel_radius revolved about Z
            // (support_function.py:284-305)
            vec3 n_xy(1.0f, 0.0f, 0.0f);
            if (l2 > eps) {
                float dir_xy_len = std::sqrt(l2);
                n_xy = dir_xy / dir_xy_len;
            }
            float direction_len = std::sqrt(l2 + direction.z * direction.z);
            float support_z = 0.0f;
            if (direction_len > eps) support_z = clampf(barrel_radius * direction.z / direction_len, -half_height, half_height);
            float barrel_radius_sq = barrel_radius * barrel_radius;
            float half_height_sq = half_height * half_height;
            float support_z_sq = support_z * support_z;
            float end_offset = std::sqrt(barrel_radius_sq - half_height_sq);
            float support_offset = std::sqrt(fmaxw(barrel_radius_sq - support_z_sq, 0.0f));
            float offset_sum = support_offset + end_offset;
            float support_radius = radius;
            if (offset_sum > eps) support_radius += (half_height_sq - support_z_sq) / offset_sum;
            result = vec3(n_xy.x * support_radius, n_xy.y * support_radius, support_z);
        }
    } else if (g.type == GEO_CONE) {
        float radius = g.scale.x, half_height = g.scale.y;
        vec3 apex(0.0f, 0.0f, half_height);
        vec3 dir_xy(direction.x, direction.y, 0.0f);
        float dir_xy_len = length(dir_xy);
        float k = half_height > eps ? radius / (2.0f * half_height) : 0.0f;
        if (dir_xy_len <= eps) {
            if (direction.z >= 0.0f) result = apex;
            else result = vec3(radius, 0.0f, -half_height);
        } else {
            if (direction.z >= k * dir_xy_len) {
                result = apex;
            } else {
                vec3 n_xy = dir_xy / dir_xy_len;
                result = vec3(n_xy.x * radius, n_xy.y * radius, -half_height);
            }
        }
    }
    return result;
}

// create_shape_support_function(center_ties=True), support_function.py:399-431
vec3 shape_support_centered(const Geom& g, vec3 direction) {
    if (g.type == GEO_BOX) {
        vec3 ad = vabs(direction);
        vec3 result = support_map_box(g, direction);
        vec3 contribution = cw_mul(ad, g.scale);
        float threshold = 1.0e-6f * (contribution.x + contribution.y + contribution.z);
        if (contribution.x <= threshold) result.x = 0.0f;
        if (contribution.y <= threshold) result.y = 0.0f;
        if (contribution.z <= threshold) result.z = 0.0f;
        return result;
    }
    return support_map(g, direction);
}

// support_function.py:647-745
vec3 closest_point_on_triangle(vec3 p, vec3 tri_a, vec3 tri_b, vec3 tri_c) {
    vec3 ab = tri_b - tri_a, ac = tri_c - tri_a;
    float ab_sq = dot(ab, ab), ac_sq = dot(ac, ac);
    const float EPS2 = 1.0e-20f;
    vec3 triangle_normal = cross(ab, ac);
    if (dot(triangle_normal, triangle_normal) < EPS2) {
        vec3 bc = tri_c - tri_b;
        float bc_sq = dot(bc, bc);
        if (ab_sq >= ac_sq && ab_sq >= bc_sq) {
            if (ab_sq < EPS2) return tri_a;
            float t = clampf(dot(p - tri_a, ab) / ab_sq, 0.0f, 1.0f);
            return tri_a + t * ab;
        } else if (ac_sq >= bc_sq) {
            float t = clampf(dot(p - tri_a, ac) / ac_sq, 0.0f, 1.0f);
            return tri_a + t * ac;
        } else {
            float t = clampf(dot(p - tri_b, bc) / bc_sq, 0.0f, 1.0f);
            return tri_b + t * bc;
        }
    }
    vec3 ap = p - tri_a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) return tri_a;
    vec3 bp = p - tri_b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) return tri_b;
    vec3 cp = p - tri_c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) return tri_c;
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        float v = d1 / (d1 - d3);
        return tri_a + v * ab;
    }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        float w = d2 / (d2 - d6);
        return tri_a + w * ac;
    }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        return tri_b + w * (tri_c - tri_b);
    }
    float denom = 1.0f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    return tri_a + v * ab + w * ac;
}
// support_function.py:467-502: a triangle's Minkowski seed is the point of the triangle nearest B's centre, nudged to the centroid
vec3 adjust_minkowski_center(const Geom& ga, vec3 center_b_world, vec3 center_b_to_a) {
    if (ga.type != GEO_TRIANGLE && ga.type != GEO_TRIANGLE_PRISM) return center_b_to_a;
    vec3 tri_a(0.0f), tri_b = ga.scale, tri_c = ga.aux;
    vec3 face_normal = cross(tri_b - tri_a, tri_c - tri_a);
    float face_normal_length_sq = length_sq(face_normal);
    vec3 projection = closest_point_on_triangle(center_b_world, tri_a, tri_b, tri_c);
    if (face_normal_length_sq < 1.0e-20f) return projection - center_b_world;
    vec3 face_normal_unit = face_normal / std::sqrt(face_normal_length_sq);
    float signed_plane_distance = dot(center_b_world - tri_a, face_normal_unit);
    vec3 plane_projection = center_b_world - signed_plane_distance * face_normal_unit;
    bool inside_face = dot(cross(tri_b - tri_a, plane_projection - tri_a), face_normal) >= 0.0f &&
                       dot(cross(tri_c - tri_b, plane_projection - tri_b), face_normal) >= 0.0f &&
                       dot(cross(tri_a - tri_c, plane_projection - tri_c), face_normal) >= 0.0f;
    if (inside_face) {
        projection = plane_projection;
        center_b_to_a = -signed_plane_distance * face_normal_unit;
    } else {
        center_b_to_a = projection - center_b_world;
    }
    vec3 to_centroid = (tri_a + tri_b + tri_c) / 3.0f - projection;
    to_centroid = to_centroid - dot(to_centroid, face_normal_unit) * face_normal_unit;
    float distance_to_centroid = length(to_centroid);
    if (distance_to_centroid > 1.0e-12f) {
        float nudge_distance = 0.01f * fminw(distance_to_centroid, std::fabs(signed_plane_distance));
        center_b_to_a = center_b_to_a + to_centroid * (nudge_distance / distance_to_centroid);
    }
    return center_b_to_a;
}
// support_function.py:505-538
vec3 minkowski_center_fallback(const Geom& ga, vec3 center_b_world) {
    if (ga.type != GEO_TRIANGLE && ga.type != GEO_TRIANGLE_PRISM) return vec3(0.0f);
    vec3 tri_a(0.0f), tri_b = ga.scale, tri_c = ga.aux;
    vec3 face_normal = cross(tri_b - tri_a, tri_c - tri_a);
    float face_normal_length_sq = length_sq(face_normal);
    if (face_normal_length_sq < 1.0e-20f) return vec3(0.0f);
    face_normal = face_normal / std::sqrt(face_normal_length_sq);
    vec3 projection = closest_point_on_triangle(center_b_world, tri_a, tri_b, tri_c);
    vec3 to_centroid = (tri_a + tri_b + tri_c) / 3.0f - projection;
    to_centroid = to_centroid - dot(to_centroid, face_normal) * face_normal;
    float to_centroid_length_sq = length_sq(to_centroid);
    vec3 fallback_direction = -face_normal;
    if (dot(center_b_world - projection, face_normal) < 0.0f) fallback_direction = face_normal;
    if (to_centroid_length_sq > 1.0e-20f) fallback_direction = fallback_direction + 0.01f * to_centroid / std::sqrt(to_centroid_length_sq);
    return normalize(fallback_direction) * 1.0e-5f;
}

// mpr.py:100-160; CENTERED selects the tie-centred box support used by MPR's own support map
template <bool CENTERED>
Vert minkowski_support(const Geom& ga, const Geom& gb, vec3 direction, quat orientation_b, vec3 position_b, float extend) {
    Vert v;
    vec3 point_a = CENTERED ? shape_support_centered(ga, direction) : support_map(ga, direction);
    vec3 tmp_direction = -direction;
    vec3 tmp = quat_rotate_inv(orientation_b, tmp_direction);
    vec3 r = CENTERED ? shape_support_centered(gb, tmp) : support_map(gb, tmp);
    r = quat_rotate(orientation_b, r);
    v.B = r + position_b;
    if (extend != 0.0f) {
        vec3 d = normalize(direction) * extend * 0.5f;
        point_a = point_a + d;
        v.B = v.B - d;
    }
    v.BtoA = point_a - v.B;
    return v;
}

// mpr.py:186-403 (primitive shapes: geometric center of A and B is the local origin)
bool solve_mpr_core(const Geom& ga, const Geom& gb, quat orientation_b, vec3 position_b, float extend, vec3& point_a, vec3& point_b,
                    vec3& normal, float& penetration) {
    const int MAX_ITER = 30;
    const float COLLIDE_EPSILON = 1e-5f;
    const float NUMERIC_EPSILON = 1e-16f;
    penetration = 0.0f;
    point_a = vec3(0.0f);
    point_b = vec3(0.0f);
    Vert v0;  // create_shape_center_function(use_precomputed_center=True) (support_function.py:541-598)
    v0.B = position_b + quat_rotate(orientation_b, gb.center);
    v0.BtoA = adjust_minkowski_center(ga, v0.B, ga.center - v0.B);
    normal = v0.BtoA;
    if (length_sq(normal) < NUMERIC_EPSILON) {
        v0.BtoA = minkowski_center_fallback(ga, v0.B);  // zero for non-triangle shapes
        if (length_sq(v0.BtoA) < NUMERIC_EPSILON) {
            float best_dot = -1.0e30f;
            vec3 best_dir(1.0f, 0.0f, 0.0f);
            for (int axis_idx = 0; axis_idx < 3; ++axis_idx) {
                vec3 probe(0.0f);
                probe[axis_idx] = 1.0f;
                Vert sv = minkowski_support<true>(ga, gb, probe, orientation_b, position_b, extend);
                float d = dot(sv.BtoA, probe);
                if (d > best_dot) {
                    best_dot = d;
                    best_dir = probe;
                }
            }
            v0.BtoA = best_dir * 1e-05f;
        }
    }
    normal = -v0.BtoA;
    Vert v1 = minkowski_support<true>(ga, gb, normal, orientation_b, position_b, extend);
    point_a = vert_a(v1);
    point_b = v1.B;
    if (dot(v1.BtoA, normal) <= 0.0f) return false;
    normal = cross(v1.BtoA, v0.BtoA);
    if (length_sq(normal) < NUMERIC_EPSILON * NUMERIC_EPSILON) {
        normal = v1.BtoA - v0.BtoA;
        normal = normalize(normal);
        vec3 temp1 = v1.BtoA;
        penetration = dot(temp1, normal);
        return true;
    }
    Vert v2 = minkowski_support<true>(ga, gb, normal, orientation_b, position_b, extend);
    if (dot(v2.BtoA, normal) <= 0.0f) return false;
    vec3 temp1 = v1.BtoA - v0.BtoA;
    vec3 temp2 = v2.BtoA - v0.BtoA;
    normal = cross(temp1, temp2);
    float dist = dot(normal, v0.BtoA);
    if (dist > 0.0f) {
        std::swap(v1, v2);
        normal = -normal;
    }
    int phase1 = 0, phase2 = 0;
    bool hit = false;
    Vert v3;
    while (true) {
        if (phase1 > MAX_ITER) return false;
        phase1 += 1;
        v3 = minkowski_support<true>(ga, gb, normal, orientation_b, position_b, extend);
        if (dot(v3.BtoA, normal) <= 0.0f) return false;
        temp1 = cross(v1.BtoA, v3.BtoA);
        if (dot(temp1, v0.BtoA) < 0.0f) {
            v2 = v3;
            temp1 = v1.BtoA - v0.BtoA;
            temp2 = v3.BtoA - v0.BtoA;
            normal = cross(temp1, temp2);
            continue;
        }
        temp1 = cross(v3.BtoA, v2.BtoA);
        if (dot(temp1, v0.BtoA) < 0.0f) {
            v1 = v3;
            temp1 = v3.BtoA - v0.BtoA;
            temp2 = v2.BtoA - v0.BtoA;
            normal = cross(temp1, temp2);
            continue;
        }
        break;
    }
    Vert v4;
    while (true) {
        phase2 += 1;
        temp1 = v2.BtoA - v1.BtoA;
        temp2 = v3.BtoA - v1.BtoA;
        normal = cross(temp1, temp2);
        float normal_sq = length_sq(normal);
        if (normal_sq < NUMERIC_EPSILON * NUMERIC_EPSILON) return false;
        if (!hit) {
            float d = dot(normal, v1.BtoA);
            hit = d >= 0.0f;
        }
        v4 = minkowski_support<true>(ga, gb, normal, orientation_b, position_b, extend);
        vec3 temp3 = v4.BtoA - v3.BtoA;
        float delta = dot(temp3, normal);
        penetration = dot(v4.BtoA, normal);
        if (delta * delta <= COLLIDE_EPSILON * COLLIDE_EPSILON * normal_sq || penetration <= 0.0f || phase2 > MAX_ITER) {
            if (hit) {
                float inv_normal = 1.0f / std::sqrt(normal_sq);
                penetration *= inv_normal;
                normal = normal * inv_normal;
                temp3 = cross(v1.BtoA, temp1);
                float gamma = dot(temp3, normal) * inv_normal;
                temp3 = cross(temp2, v1.BtoA);
                float beta = dot(temp3, normal) * inv_normal;
                float alpha = 1.0f - gamma - beta;
                point_a = alpha * vert_a(v1) + beta * vert_a(v2) + gamma * vert_a(v3);
                point_b = alpha * v1.B + beta * v2.B + gamma * v3.B;
            }
            return hit;
        }
        temp1 = cross(v4.BtoA, v0.BtoA);
        float dt = dot(temp1, v1.BtoA);
        if (dt >= 0.0f) {
            dt = dot(temp1, v2.BtoA);
            if (dt >= 0.0f) v1 = v4;
            else v3 = v4;
        } else {
            dt = dot(temp1, v3.BtoA);
            if (dt >= 0.0f) v2 = v4;
            else v1 = v4;
        }
    }
}

// ---------------------------------------------------------------- GJK (simplex_solver.py)
const float GJK_EPSILON = 1e-8f;
struct Simplex {
    vec3 v[8];  // v[2i] = B, v[2i+1] = BtoA
};
struct vec4f4 {
    float c[4];
    vec4f4() { c[0] = c[1] = c[2] = c[3] = 0.0f; }
};

void closest_segment(const Simplex& s, int i0, int i1, vec3& closest, vec4f4& bc, uint32_t& mask) {
    vec3 a = s.v[2 * i0 + 1], b = s.v[2 * i1 + 1];
    vec3 edge = b - a;
    float vsq = length_sq(edge);
    bool degenerate = vsq < GJK_EPSILON;
    float denom = degenerate ? GJK_EPSILON : vsq;
    float t = -dot(a, edge) / denom;
    float lambda0 = 1.0f - t, lambda1 = t;
    mask = (1u << i0) | (1u << i1);
    bc = vec4f4();
    if (lambda0 < 0.0f || degenerate) {
        mask = 1u << i1;
        lambda0 = 0.0f;
        lambda1 = 1.0f;
    } else if (lambda1 < 0.0f) {
        mask = 1u << i0;
        lambda0 = 1.0f;
        lambda1 = 0.0f;
    }
    bc.c[i0] = lambda0;
    bc.c[i1] = lambda1;
    closest = lambda0 * a + lambda1 * b;
}

void closest_triangle(const Simplex& s, int i0, int i1, int i2, vec3& closest_out, vec4f4& bc_out, uint32_t& mask_out) {
    vec3 a = s.v[2 * i0 + 1], b = s.v[2 * i1 + 1], c = s.v[2 * i2 + 1];
    vec3 u = a - b, w = a - c;
    vec3 normal = cross(u, w);
    float t = length_sq(normal);
    bool degenerate = t < GJK_EPSILON;
    float denom = degenerate ? GJK_EPSILON : t;
    float it = 1.0f / denom;
    vec3 c1 = cross(u, a), c2 = cross(a, w);
    float lambda2 = dot(c1, normal) * it;
    float lambda1 = dot(c2, normal) * it;
    float lambda0 = 1.0f - lambda2 - lambda1;
    float best_distance = 1e30f;
    vec3 closest_pt(0.0f);
    vec4f4 bc;
    uint32_t mask = 0;
    vec3 cl;
    vec4f4 bt;
    uint32_t mm;
    if (lambda0 < 0.0f || degenerate) {
        closest_segment(s, i1, i2, cl, bt, mm);
        float dist = length_sq(cl);
        if (dist < best_distance) { bc = bt; mask = mm; best_distance = dist; closest_pt = cl; }
    }
    if (lambda1 < 0.0f || degenerate) {
        closest_segment(s, i0, i2, cl, bt, mm);
        float dist = length_sq(cl);
        if (dist < best_distance) { bc = bt; mask = mm; best_distance = dist; closest_pt = cl; }
    }
    if (lambda2 < 0.0f || degenerate) {
        closest_segment(s, i0, i1, cl, bt, mm);
        float dist = length_sq(cl);
        if (dist < best_distance) { bc = bt; mask = mm; closest_pt = cl; }
    }
    if (mask != 0) {
        closest_out = closest_pt; bc_out = bc; mask_out = mask;
        return;
    }
    bc.c[i0] = lambda0;
    bc.c[i1] = lambda1;
    bc.c[i2] = lambda2;
    mask_out = (1u << i0) | (1u << i1) | (1u << i2);
    bc_out = bc;
    closest_out = lambda0 * a + lambda1 * b + lambda2 * c;
}

inline float determinant(vec3 a, vec3 b, vec3 c, vec3 d) { return dot(b - a, cross(c - a, d - a)); }

void closest_tetrahedron(const Simplex& s, vec3& closest_out, vec4f4& bc_out, uint32_t& mask_out) {
    vec3 v0 = s.v[1], v1 = s.v[3], v2 = s.v[5], v3 = s.v[7];
    float det_t = determinant(v0, v1, v2, v3);
    bool degenerate = std::fabs(det_t) < GJK_EPSILON;
    float denom = degenerate ? GJK_EPSILON : det_t;
    float inverse_det_t = 1.0f / denom;
    vec3 zero(0.0f);
    float lambda0 = determinant(zero, v1, v2, v3) * inverse_det_t;
    float lambda1 = determinant(v0, zero, v2, v3) * inverse_det_t;
    float lambda2 = determinant(v0, v1, zero, v3) * inverse_det_t;
    float lambda3 = 1.0f - lambda0 - lambda1 - lambda2;
    float best_distance = 1e30f;
    vec3 closest_pt(0.0f);
    vec4f4 bc;
    uint32_t mask = 0;
    vec3 cl;
    vec4f4 bt;
    uint32_t mm;
    if (lambda0 < 0.0f || degenerate) {
        closest_triangle(s, 1, 2, 3, cl, bt, mm);
        float dist = length_sq(cl);
        if (dist < best_distance) { bc = bt; mask = mm; best_distance = dist; closest_pt = cl; }
    }
    if (lambda1 < 0.0f || degenerate) {
        closest_triangle(s, 0, 2, 3, cl, bt, mm);
        float dist = length_sq(cl);
        if (dist < best_distance) { bc = bt; mask = mm; best_distance = dist; closest_pt = cl; }
    }
    if (lambda2 < 0.0f || degenerate) {
        closest_triangle(s, 0, 1, 3, cl, bt, mm);
        float dist = length_sq(cl);
        if (dist < best_distance) { bc = bt; mask = mm; best_distance = dist; closest_pt = cl; }
    }
    if (lambda3 < 0.0f || degenerate) {
        closest_triangle(s, 0, 1, 2, cl, bt, mm);
        float dist = length_sq(cl);
        if (dist < best_distance) { bc = bt; mask = mm; closest_pt = cl; }
    }
    if (mask != 0) {
        closest_out = closest_pt; bc_out = bc; mask_out = mask;
        return;
    }
    bc.c[0] = lambda0; bc.c[1] = lambda1; bc.c[2] = lambda2; bc.c[3] = lambda3;
    bc_out = bc;
    mask_out = 15u;
    closest_out = zero;
}

void simplex_get_closest(const Simplex& s, const vec4f4& bc, uint32_t mask, vec3& point_a, vec3& point_b) {
    point_a = vec3(0.0f);
    point_b = vec3(0.0f);
    for (int i = 0; i < 4; ++i) {
        if ((mask & (1u << i)) == 0) continue;
        vec3 B = s.v[2 * i], BtoA = s.v[2 * i + 1];
        float w = bc.c[i];
        point_a = point_a + w * (B + BtoA);
        point_b = point_b + w * B;
    }
}

// simplex_solver.py:331-469; returns `separated`
bool solve_closest_distance_core(const Geom& ga, const Geom& gb, quat orientation_b, vec3 position_b, float extend, vec3& point_a,
                                 vec3& point_b, vec3& normal, float& distance) {
    const int MAX_ITER = 30;
    const float COLLIDE_EPSILON = 1e-4f;
    distance = 0.0f;
    point_a = vec3(0.0f);
    point_b = vec3(0.0f);
    normal = vec3(0.0f);
    Simplex simplex;
    vec4f4 bary;
    uint32_t usage = 0;
    int iter_count = MAX_ITER;
    vec3 center_b = position_b + quat_rotate(orientation_b, gb.center);
    vec3 v = adjust_minkowski_center(ga, center_b, ga.center - center_b);  // center.BtoA
    float dist_sq = length_sq(v);
    vec3 last_search_dir(1.0f, 0.0f, 0.0f);
    while (iter_count > 0) {
        iter_count -= 1;
        if (dist_sq < COLLIDE_EPSILON * COLLIDE_EPSILON) {
            distance = 0.0f;
            normal = vec3(0.0f);
            simplex_get_closest(simplex, bary, usage, point_a, point_b);
            return false;
        }
        vec3 search_dir = -v;
        last_search_dir = search_dir;
        Vert w = minkowski_support<false>(ga, gb, search_dir, orientation_b, position_b, extend);
        vec3 w_v = w.BtoA;
        float delta_dist = dot(v, v - w_v);
        if (delta_dist <= 0.0f || delta_dist * delta_dist < (COLLIDE_EPSILON * COLLIDE_EPSILON * dist_sq)) break;
        bool is_duplicate = false;
        for (int i = 0; i < 4; ++i)
            if ((usage & (1u << i)) != 0)
                if (length_sq(simplex.v[2 * i + 1] - w_v) < COLLIDE_EPSILON * COLLIDE_EPSILON) {
                    is_duplicate = true;
                    break;
                }
        if (is_duplicate) break;
        int use_count = 0, free_slot = 0;
        int indices[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            if ((usage & (1u << i)) != 0) {
                indices[use_count] = i;
                use_count += 1;
            } else {
                free_slot = i;
            }
        }
        indices[use_count] = free_slot;
        use_count += 1;
        simplex.v[2 * free_slot] = w.B;
        simplex.v[2 * free_slot + 1] = w.BtoA;
        vec3 closest(0.0f);
        bool success = true;
        if (use_count == 1) {
            int i0 = indices[0];
            closest = simplex.v[2 * i0 + 1];
            usage = 1u << i0;
            bary.c[i0] = 1.0f;
        } else if (use_count == 2) {
            closest_segment(simplex, indices[0], indices[1], closest, bary, usage);
        } else if (use_count == 3) {
            closest_triangle(simplex, indices[0], indices[1], indices[2], closest, bary, usage);
        } else if (use_count == 4) {
            closest_tetrahedron(simplex, closest, bary, usage);
            success = !(usage == 15u);
        } else {
            success = false;
        }
        if (!success) {
            distance = 0.0f;
            normal = vec3(0.0f);
            simplex_get_closest(simplex, bary, usage, point_a, point_b);
            return false;
        }
        v = closest;
        dist_sq = length_sq(v);
    }
    simplex_get_closest(simplex, bary, usage, point_a, point_b);
    vec3 delta = point_b - point_a;
    float delta_len_sq = length_sq(delta);
    if (delta_len_sq > GJK_EPSILON * GJK_EPSILON) {
        distance = std::sqrt(delta_len_sq);
        normal = delta * (1.0f / distance);
    } else {
        distance = std::sqrt(dist_sq);
        if (distance > COLLIDE_EPSILON) {
            normal = v * (-1.0f / distance);
        } else {
            float nsq = length_sq(last_search_dir);
            if (nsq > 0.0f) normal = last_search_dir * (1.0f / std::sqrt(nsq));
            else normal = vec3(1.0f, 0.0f, 0.0f);
        }
    }
    return true;
}

// ---------------------------------------------------------------- manifold (multicontact.py)
const float MC_EPS = 0.00001f;
const float SIN_TILT_ANGLE = 0.03489949670250097f;      // sin(2 deg)
const float COS_TILT_ANGLE = 0.9993908270190958f;       // cos(2 deg)
const float COS_DEEPEST_THRESHOLD = 0.9999984769132877f;  // cos(0.1 deg)

inline float signed_area(vec2 a, vec2 b, vec2 q) { return (b.x - a.x) * (q.y - a.y) - (b.y - a.y) * (q.x - a.x); }

vec3 ray_plane_intersection(vec3 ro, vec3 rd, float plane_d, vec3 plane_normal) {
    float denom = dot(rd, plane_normal);
    if (std::fabs(denom) < 1.0e-12f) return ro;
    float t = -(dot(ro, plane_normal) + plane_d) / denom;
    return ro + rd * t;
}
struct BodyProjector {
    float plane_d;
    vec3 normal;
    BodyProjector() : plane_d(0.0f) {}
};
struct PlaneTracker {
    vec3 reference_point, previous_point, normal;
    float largest_area_sq;
    PlaneTracker() : largest_area_sq(0.0f) {}
};
void update_tracker(PlaneTracker& t, vec3 p, int id) {
    if (id == 0) {
        t.reference_point = p;
        t.largest_area_sq = 0.0f;
    } else if (id == 1) {
        t.previous_point = p;
    } else {
        vec3 e1 = t.previous_point - t.reference_point;
        vec3 e2 = p - t.reference_point;
        vec3 c = cross(e1, e2);
        float area_sq = dot(c, c);
        if (area_sq > t.largest_area_sq) {
            t.largest_area_sq = area_sq;
            t.normal = c;
        }
        t.previous_point = p;
    }
}
vec3 line_segment_projector_normal(vec3 segment_dir, vec3 reference_normal) {
    vec3 right = cross(segment_dir, reference_normal);
    vec3 n = cross(right, segment_dir);
    float len = length(n);
    return len > 1.0e-12f ? n * (1.0f / len) : reference_normal;
}
void create_body_projectors(const PlaneTracker& ta, vec3 anchor_a, const PlaneTracker& tb, vec3 anchor_b, vec3 contact_normal,
                            BodyProjector& pa, BodyProjector& pb) {
    if (ta.largest_area_sq == 0.0f && tb.largest_area_sq == 0.0f) {
        vec3 dir_a = ta.previous_point - ta.reference_point;
        vec3 dir_b = tb.previous_point - tb.reference_point;
        vec3 pop_a = 0.5f * (ta.reference_point + ta.previous_point);
        pa.normal = line_segment_projector_normal(dir_a, contact_normal);
        pa.plane_d = -dot(pop_a, pa.normal);
        vec3 pop_b = 0.5f * (tb.reference_point + tb.previous_point);
        pb.normal = line_segment_projector_normal(dir_b, contact_normal);
        pb.plane_d = -dot(pop_b, pb.normal);
        return;
    }
    if (ta.largest_area_sq > 0.0f) {
        float inv = 1.0f / std::sqrt(fmaxw(1.0e-12f, ta.largest_area_sq));
        pa.normal = ta.normal * inv;
        pa.plane_d = -dot(anchor_a, pa.normal);
    }
    if (tb.largest_area_sq > 0.0f) {
        float inv = 1.0f / std::sqrt(fmaxw(1.0e-12f, tb.largest_area_sq));
        pb.normal = tb.normal * inv;
        pb.plane_d = -dot(anchor_b, pb.normal);
    }
    if (ta.largest_area_sq == 0.0f) {
        vec3 dir = ta.previous_point - ta.reference_point;
        vec3 pop = 0.5f * (ta.reference_point + ta.previous_point);
        pa.normal = line_segment_projector_normal(dir, pb.normal);
        pa.plane_d = -dot(pop, pa.normal);
    }
    if (tb.largest_area_sq == 0.0f) {
        vec3 dir = tb.previous_point - tb.reference_point;
        vec3 pop = 0.5f * (tb.reference_point + tb.previous_point);
        pb.normal = line_segment_projector_normal(dir, pa.normal);
        pb.plane_d = -dot(pop, pb.normal);
    }
}

vec2 intersection_point(vec2 s0, vec2 s1, vec2 a, vec2 b) {
    float sa = signed_area(s0, s1, a), sb = signed_area(s0, s1, b);
    float t = std::fabs(sa) / std::fabs(sa - sb);
    return (1.0f - t) * a + t * b;
}
void insert_vec2(vec2* arr, int arr_count, int index, vec2 element) {
    int i = arr_count;
    while (i > index) {
        arr[i] = arr[i - 1];
        i -= 1;
    }
    arr[index] = element;
}
// multicontact.py:311-381
int trim_in_place(vec2 s0, vec2 s1, vec2* loop, int loop_count) {
    if (loop_count < 3) return loop_count;
    vec2 intersection_a, intersection_b;
    int change_a = -1, change_b = -1;
    bool keep = false;
    bool prev_outside = signed_area(s0, s1, loop[0]) <= 0.0f;
    for (int i = 0; i < loop_count; ++i) {
        int next_idx = (i + 1) % loop_count;
        bool outside = signed_area(s0, s1, loop[next_idx]) <= 0.0f;
        if (outside != prev_outside) {
            vec2 ip = intersection_point(s0, s1, loop[i], loop[next_idx]);
            if (change_a < 0) {
                change_a = i;
                keep = !prev_outside;
                intersection_a = ip;
            } else {
                change_b = i;
                intersection_b = ip;
            }
        }
        prev_outside = outside;
    }
    int new_loop_count;
    if (change_a >= 0 && change_b >= 0) {
        int loop_indexer = -1;
        new_loop_count = loop_count;
        int i = 0;
        while (i < loop_count) {
            if (keep) {
                loop_indexer += 1;
                loop[loop_indexer] = loop[i];
            }
            if (i == change_a || i == change_b) {
                vec2 pt = i == change_a ? intersection_a : intersection_b;
                if (loop_indexer == i && !keep) {
                    loop_indexer += 1;
                    insert_vec2(loop, new_loop_count, loop_indexer, pt);
                    new_loop_count += 1;
                    i += 1;
                    change_b += 1;
                    loop_count += 1;
                } else {
                    loop_indexer += 1;
                    loop[loop_indexer] = pt;
                }
                keep = !keep;
            }
            i += 1;
        }
        new_loop_count = loop_indexer + 1;
    } else if (prev_outside) {
        new_loop_count = 0;
    } else {
        new_loop_count = loop_count;
    }
    return new_loop_count;
}
// multicontact.py:384-484; trim_poly aliases loop[5..9]
int trim_all_in_place(vec2* trim_poly, int trim_poly_count, vec2* loop, int loop_count) {
    if (trim_poly_count <= 1) return std::min(1, loop_count);
    const float move_distance = 1e-5f;
    if (trim_poly_count == 2) {
        vec2 p0 = trim_poly[0], p1 = trim_poly[1];
        float dx = p1.x - p0.x, dy = p1.y - p0.y;
        float dir_len = std::sqrt(dx * dx + dy * dy);
        if (dir_len > 1e-10f) {
            float inv = 1.0f / dir_len;
            float ox = -dy * inv * move_distance, oy = dx * inv * move_distance;
            trim_poly[0] = vec2(p0.x - ox, p0.y - oy);
            trim_poly[1] = vec2(p1.x - ox, p1.y - oy);
            trim_poly[2] = vec2(p1.x + ox, p1.y + oy);
            trim_poly[3] = vec2(p0.x + ox, p0.y + oy);
            trim_poly_count = 4;
        } else {
            return std::min(1, loop_count);
        }
    }
    if (loop_count == 2) {
        vec2 p0 = loop[0], p1 = loop[1];
        float dx = p1.x - p0.x, dy = p1.y - p0.y;
        float dir_len = std::sqrt(dx * dx + dy * dy);
        if (dir_len > 1e-10f) {
            float inv = 1.0f / dir_len;
            float ox = -dy * inv * move_distance, oy = dx * inv * move_distance;
            loop[0] = vec2(p0.x - ox, p0.y - oy);
            loop[1] = vec2(p1.x - ox, p1.y - oy);
            loop[2] = vec2(p1.x + ox, p1.y + oy);
            loop[3] = vec2(p0.x + ox, p0.y + oy);
            loop_count = 4;
        } else {
            return std::min(1, loop_count);
        }
    }
    int current = loop_count;
    vec2 trim_poly_0 = trim_poly[0];
    for (int i = 0; i < trim_poly_count; ++i) {
        vec2 s0 = trim_poly[i];
        vec2 s1 = i == trim_poly_count - 1 ? trim_poly_0 : trim_poly[i + 1];
        current = trim_in_place(s0, s1, loop, current);
    }
    return current;
}
// multicontact.py:487-580
void approx_max_quad(const vec2* hull, int n, int out[4]) {
    int p1 = 0, p3 = 1;
    vec2 diff = hull[p1] - hull[p3];
    float max_dist_sq = diff.x * diff.x + diff.y * diff.y;
    const float tie = 1.0e-3f;
    int j = 1;
    for (int i = 0; i < n; ++i) {
        vec2 hi = hull[i], hi1 = hull[(i + 1) % n];
        while (true) {
            float area_j1 = signed_area(hi, hi1, hull[(j + 1) % n]);
            float area_j = signed_area(hi, hi1, hull[j]);
            if (area_j1 > area_j) j = (j + 1) % n;
            else break;
        }
        vec2 hj = hull[j];
        vec2 d1 = hull[i] - hj;
        float ds1 = d1.x * d1.x + d1.y * d1.y;
        if (ds1 > max_dist_sq * (1.0f + tie)) {
            max_dist_sq = ds1;
            p1 = i;
            p3 = j;
        }
        vec2 d2 = hull[(i + 1) % n] - hj;
        float ds2 = d2.x * d2.x + d2.y * d2.y;
        if (ds2 > max_dist_sq * (1.0f + tie)) {
            max_dist_sq = ds2;
            p1 = (i + 1) % n;
            p3 = j;
        }
    }
    int p2 = 0, p4 = 0;
    float max_area_1 = 0.0f, max_area_2 = 0.0f;
    vec2 hp1 = hull[p1], hp3 = hull[p3];
    for (int i = 0; i < n; ++i) {
        float area = signed_area(hp1, hp3, hull[i]);
        if (area > max_area_1 * (1.0f + tie)) {
            max_area_1 = area;
            p2 = i;
        } else if (-area > max_area_2 * (1.0f + tie)) {
            max_area_2 = -area;
            p4 = i;
        }
    }
    out[0] = p1; out[1] = p2; out[2] = p3; out[3] = p4;
}
// multicontact.py:583-620
int remove_zero_length_edges(vec2* loop, int loop_count, float eps) {
    if (loop_count < 2) return 0;
    int write_idx = 0;
    for (int read_idx = 1; read_idx < loop_count; ++read_idx) {
        vec2 diff = loop[read_idx] - loop[write_idx];
        if (length_sq2(diff) > eps) {
            write_idx += 1;
            loop[write_idx] = loop[read_idx];
        }
    }
    int new_count;
    if (write_idx > 0) {
        vec2 diff = loop[write_idx] - loop[0];
        new_count = length_sq2(diff) < eps ? write_idx : write_idx + 1;
    } else {
        new_count = write_idx + 1;
    }
    if (new_count < 2) new_count = 0;
    return new_count;
}
bool add_avoid_duplicates(vec2* arr, int& count, vec2 v, float eps) {
    if (count > 0 && length_sq2(arr[0] - v) < eps) return false;
    if (count > 1 && length_sq2(arr[count - 1] - v) < eps) return false;
    arr[count] = v;
    count += 1;
    return true;
}
// newton/_src/math/__init__.py:232-275
void orthonormal_basis(vec3 n, vec3& b1, vec3& b2) {
    if (n.z < 0.0f) {
        float a = 1.0f / (1.0f - n.z);
        float b = n.x * n.y * a;
        b1 = vec3(1.0f - n.x * n.x * a, -b, n.x);
        b2 = vec3(b, n.y * n.y * a - 1.0f, -n.y);
    } else {
        float a = 1.0f / (1.0f + n.z);
        float b = -n.x * n.y * a;
        b1 = vec3(1.0f - n.x * n.x * a, b, -n.x);
        b2 = vec3(b, 1.0f - n.y * n.y * a, -n.y);
    }
}

// one generated contact before the writer's gap test
struct ContactOut {
    vec3 center, normal;
    float distance;
};

struct PairCtx {
    const o_model* m;
    const float* body_q;
    o_contacts* ct;
    int shape_a, shape_b;
    Geom ga, gb;  // geometry as seen by GJK/MPR (sphere/capsule radii shrunk to 1e-4)
    float radius_eff_a, radius_eff_b, margin_a, margin_b;
    int written;
    // write_contact_to_reducer instead of write_contact (the mesh-triangle leg): every generated contact is buffered as
    // (centre, normal, distance, fingerprint) with NO gap test; fingerprint = (sort_sub_key << 3) | emission index
    std::vector<float>* raw = nullptr;
    int sort_sub_key = 0;
};

// collision_core.py:173-278
ContactOut post_process_axial(ContactOut c, const PairCtx& P, vec3 pos_a, quat rot_a, vec3 pos_b, quat rot_b) {
    int type_a = P.ga.type, type_b = P.gb.type;
    vec3 normal = c.normal;
    if (type_a == GEO_SPHERE || type_a == GEO_CAPSULE) {
        c.center = c.center + normal * (P.radius_eff_a * 0.5f);
        c.distance = c.distance - P.radius_eff_a;
    }
    if (type_b == GEO_SPHERE || type_b == GEO_CAPSULE) {
        c.center = c.center - normal * (P.radius_eff_b * 0.5f);
        c.distance = c.distance - P.radius_eff_b;
    }
    auto discrete = [](int t) { return t == GEO_BOX || t == GEO_CONVEX_MESH || t == GEO_PLANE || t == GEO_TRIANGLE || t == GEO_TRIANGLE_PRISM; };
    bool is_discrete_a = discrete(type_a), is_discrete_b = discrete(type_b);
    bool is_axial_a = type_a == GEO_CYLINDER || type_a == GEO_CONE;
    bool is_axial_b = type_b == GEO_CYLINDER || type_b == GEO_CONE;
    if ((is_discrete_a && is_axial_b) || (is_discrete_b && is_axial_a)) {
        vec3 shape_axis, shape_pos, axial_normal;
        float shape_radius, shape_half_height;
        bool is_cone;
        if (is_discrete_a && is_axial_b) {
            shape_axis = quat_rotate(rot_b, vec3(0.0f, 0.0f, 1.0f));
            shape_radius = P.gb.scale.x;
            shape_half_height = P.gb.scale.y;
            is_cone = type_b == GEO_CONE;
            shape_pos = pos_b;
            axial_normal = normal;
        } else {
            shape_axis = quat_rotate(rot_a, vec3(0.0f, 0.0f, 1.0f));
            shape_radius = P.ga.scale.x;
            shape_half_height = P.ga.scale.y;
            is_cone = type_a == GEO_CONE;
            shape_pos = pos_a;
            axial_normal = -normal;
        }
        float axis_normal_dot = std::fabs(dot(shape_axis, axial_normal));
        bool is_rolling = false;
        if (is_cone) {
            float cone_half_angle = std::atan2(shape_radius, 2.0f * shape_half_height);
            const float tol = 2.0f * 3.14159265358979323846f / 180.0f;
            float lower = std::sin(cone_half_angle - tol), upper = std::sin(cone_half_angle + tol);
            if (axis_normal_dot >= lower && axis_normal_dot <= upper) is_rolling = true;
        } else {
            if (axis_normal_dot <= 0.03489949670250097f) is_rolling = true;
        }
        if (is_rolling) {
            vec3 pn = normalize(cross(shape_axis, axial_normal));
            // project_point_onto_plane (collision_core.py:51-67)
            vec3 to_point = c.center - shape_pos;
            float dist = dot(to_point, pn);
            c.center = c.center - pn * dist;
        }
    }
    return c;
}

// write_contact(output_index = -1) (collide.py:206-254)
void emit(PairCtx& P, ContactOut c, vec3 pos_a, quat rot_a, vec3 pos_b, quat rot_b) {
    if (P.ga.type == GEO_TRIANGLE_PRISM && c.distance < 0.0f) {
        // post_process_triangle_contact (collision_core.py:280-322): a penetrating contact of a heightfield prism moves to the
        // physical triangle face -- normal = the face normal (up), distance along it
        vec3 normal_local = cross(P.ga.scale, P.ga.aux);
        float normal_length_sq = length_sq(normal_local);
        if (normal_length_sq >= 1.0e-20f) {
            normal_local = normal_local / std::sqrt(normal_length_sq);
            if (normal_local.z < 0.0f) normal_local = -normal_local;
            vec3 normal_world = quat_rotate(rot_a, normal_local);
            vec3 point_b_world = c.center + 0.5f * c.distance * c.normal;
            vec3 point_b_local = quat_rotate_inv(rot_a, point_b_world - pos_a);
            vec3 projected_b = point_b_local - dot(point_b_local, normal_local) * normal_local;
            vec3 point_a = closest_point_on_triangle(projected_b, vec3(0.0f), P.ga.scale, P.ga.aux);
            float distance = 0.0f;
            if (length_sq(point_a - projected_b) < 1.0e-10f) distance = dot(point_b_local - point_a, normal_local);
            c.center = quat_rotate(rot_a, point_a) + pos_a + 0.5f * distance * normal_world;
            c.normal = normal_world;
            c.distance = distance;
        }
    }
    c = post_process_axial(c, P, pos_a, rot_a, pos_b, rot_b);
    if (P.raw) {
        const float key = (float)((P.sort_sub_key << 3) | P.written);  // (exact: keys of the test scenes stay below 2^24)
        for (float v : {c.center.x, c.center.y, c.center.z, c.normal.x, c.normal.y, c.normal.z, c.distance, key}) P.raw->push_back(v);
        P.written += 1;
        return;
    }
    float total_separation_needed = P.radius_eff_a + P.radius_eff_b + P.margin_a + P.margin_b;
    vec3 n = normalize(c.normal);
    vec3 a_world = c.center - n * (0.5f * c.distance + P.radius_eff_a);
    vec3 b_world = c.center + n * (0.5f * c.distance + P.radius_eff_b);
    float distance = dot(b_world - a_world, n);
    float d = distance - total_separation_needed;
    float contact_gap = P.m->shape_gap[P.shape_a] + P.m->shape_gap[P.shape_b];
    if (d > contact_gap) return;
    int index = P.ct->rigid_contact_count[0];
    P.ct->rigid_contact_count[0] += 1;
    write_contact_at(P.m, P.body_q, P.ct, index, P.shape_a, P.shape_b, c.center, c.normal, c.distance, P.radius_eff_a,
                     P.radius_eff_b, P.margin_a, P.margin_b);
    P.written += 1;
}

// multicontact.py:758-956 (+ extract_4_point_contact_manifolds :641-756)
int build_manifold(PairCtx& P, quat orientation_a, vec3 position_a_world, quat rel_q, vec3 rel_p, vec3 p_a, vec3 p_b, vec3 normal) {
    static const float PC[5] = {1.0f, 0.30901699437494745f, -0.8090169943749473f, -0.8090169943749476f, 0.30901699437494723f};
    static const float PS[5] = {0.0f, 0.9510565162951535f, 0.5877852522924732f, -0.587785252292473f, -0.9510565162951536f};
    int a_count = 0, b_count = 0;
    vec3 tangent_a, tangent_b;
    orthonormal_basis(normal, tangent_a, tangent_b);
    PlaneTracker tracker_a, tracker_b;
    vec3 center = 0.5f * (p_a + p_b);
    vec2 b_buffer[10];
    vec2* a_buffer = b_buffer + 5;
    vec3 local_normal_b = quat_rotate_inv(rel_q, -normal);
    vec3 local_ta_b = quat_rotate_inv(rel_q, -tangent_a);
    vec3 local_tb_b = quat_rotate_inv(rel_q, -tangent_b);
    for (int e = 0; e < 5; ++e) {
        float c = PC[e], s = PS[e];
        float cos_tilt = COS_TILT_ANGLE;
        float c_sin = c * SIN_TILT_ANGLE, s_sin = s * SIN_TILT_ANGLE;
        vec3 dir_a = normal * cos_tilt + c_sin * tangent_a + s_sin * tangent_b;
        vec3 pt_a_3d = support_map(P.ga, dir_a);
        vec3 projected_a = pt_a_3d - center;
        vec2 pt_a_2d(dot(tangent_a, projected_a), dot(tangent_b, projected_a));
        if (add_avoid_duplicates(a_buffer, a_count, pt_a_2d, MC_EPS)) update_tracker(tracker_a, pt_a_3d, a_count - 1);
        vec3 local_dir_b = local_normal_b * cos_tilt + c_sin * local_ta_b + s_sin * local_tb_b;
        vec3 pt_b_local = support_map(P.gb, local_dir_b);
        vec3 pt_b_3d = quat_rotate(rel_q, pt_b_local) + rel_p;
        vec3 projected_b = pt_b_3d - center;
        vec2 pt_b_2d(dot(tangent_a, projected_b), dot(tangent_b, projected_b));
        if (add_avoid_duplicates(b_buffer, b_count, pt_b_2d, MC_EPS)) update_tracker(tracker_b, pt_b_3d, b_count - 1);
    }
    vec3 normal_world = quat_rotate(orientation_a, normal);
    vec3 position_a_ws = position_a_world;
    vec3 position_b_ws = quat_rotate(orientation_a, rel_p) + position_a_world;
    quat quaternion_a_ws = orientation_a;
    quat quaternion_b_ws = orientation_a * rel_q;
    int count_out = 0;
    float normal_dot = 0.0f;
    if (!(a_count < 2 || b_count < 2)) {
        BodyProjector projector_a, projector_b;
        create_body_projectors(tracker_a, p_a, tracker_b, p_b, normal, projector_a, projector_b);
        bool dev_a = std::fabs(dot(normal, projector_a.normal)) < COS_TILT_ANGLE;
        bool dev_b = std::fabs(dot(normal, projector_b.normal)) < COS_TILT_ANGLE;
        if (!(dev_a || dev_b)) {
            normal_dot = std::fabs(dot(projector_a.normal, projector_b.normal));
            int loop_count = trim_all_in_place(a_buffer, a_count, b_buffer, b_count);
            loop_count = remove_zero_length_edges(b_buffer, loop_count, MC_EPS);
            if (loop_count > 1) {
                int result[4] = {0, 1, 2, 3};
                if (loop_count > 4) {
                    approx_max_quad(b_buffer, loop_count, result);
                    loop_count = 4;
                }
                for (int i = 0; i < loop_count; ++i) {
                    int ia = result[i];
                    vec3 p_local = b_buffer[ia].x * tangent_a + b_buffer[ia].y * tangent_b + center;
                    vec3 a = ray_plane_intersection(p_local, normal, projector_a.plane_d, projector_a.normal);
                    vec3 b = ray_plane_intersection(p_local, normal, projector_b.plane_d, projector_b.normal);
                    vec3 contact_point_local = 0.5f * (a + b);
                    ContactOut c;
                    c.distance = dot(b - a, normal);
                    c.center = quat_rotate(orientation_a, contact_point_local) + position_a_world;
                    c.normal = normal_world;
                    emit(P, c, position_a_ws, quaternion_a_ws, position_b_ws, quaternion_b_ws);
                }
            } else {
                normal_dot = 0.0f;
                loop_count = 0;
            }
            count_out = std::min(loop_count, 4);
        }
    }
    if (normal_dot < COS_DEEPEST_THRESHOLD || count_out == 0) {
        ContactOut c;
        vec3 deepest_center_local = 0.5f * (p_a + p_b);
        c.distance = dot(p_b - p_a, normal);
        c.center = quat_rotate(orientation_a, deepest_center_local) + position_a_world;
        c.normal = normal_world;
        emit(P, c, position_a_ws, quaternion_a_ws, position_b_ws, quaternion_b_ws);
        count_out += 1;
    }
    return count_out;
}

bool supported_type(int t) {
    return t == GEO_BOX || t == GEO_SPHERE || t == GEO_CAPSULE || t == GEO_ELLIPSOID || t == GEO_CYLINDER || t == GEO_CONE ||
           t == GEO_CONVEX_MESH;
}
void bind_mesh(const o_model* m, int shape, Geom& g) {
    if (g.type != GEO_CONVEX_MESH) return;
    g.points = m->mesh_points + 3 * m->shape_mesh_start[shape];
    g.count = m->shape_mesh_count[shape];
    g.center = 0.5f * (ld3(m->shape_collision_aabb_lower, shape) + ld3(m->shape_collision_aabb_upper, shape));
}

}  // namespace

namespace orc {
vec3 support_map_generic(int type, vec3 scale, vec3 direction) {
    Geom g;
    g.type = type;
    g.scale = scale;
    return support_map(g, direction);
}
int convex_pair_contacts(const o_model* m, int shape_a, int shape_b, const float* geom_data, const float* geom_xform,
                         const float* aabb_lower, const float* aabb_upper, const float* body_q, o_contacts* ct) {
    if (shape_a == shape_b || shape_a < 0 || shape_b < 0) return 0;
    PairCtx P;
    P.m = m; P.body_q = body_q; P.ct = ct; P.shape_a = shape_a; P.shape_b = shape_b; P.written = 0;
    P.ga.type = m->shape_type[shape_a];
    P.gb.type = m->shape_type[shape_b];
    P.ga.scale = vec3(geom_data[4 * shape_a], geom_data[4 * shape_a + 1], geom_data[4 * shape_a + 2]);
    P.gb.scale = vec3(geom_data[4 * shape_b], geom_data[4 * shape_b + 1], geom_data[4 * shape_b + 2]);
    // pairs arrive type-sorted, so an infinite plane (PLANE = 1) can only be shape A
    bool is_infinite_plane_a = P.ga.type == GEO_PLANE && P.ga.scale.x == 0.0f && P.ga.scale.y == 0.0f;
    bool is_infinite_plane_b = P.gb.type == GEO_PLANE && P.gb.scale.x == 0.0f && P.gb.scale.y == 0.0f;
    if (is_infinite_plane_a && is_infinite_plane_b) return 0;  // narrow_phase.py:1111-1112
    if (is_infinite_plane_b) return 0;                          // cannot happen after type sorting
    // finite planes are rectangles with their own support map (support_function.py:334-345); meshes etc. are not restated
    if (!(P.ga.type == GEO_PLANE || supported_type(P.ga.type)) || !supported_type(P.gb.type)) return 0;
    bind_mesh(m, shape_a, P.ga);
    bind_mesh(m, shape_b, P.gb);
    P.margin_a = geom_data[4 * shape_a + 3];
    P.margin_b = geom_data[4 * shape_b + 3];
    transform Xa = ldx(geom_xform, shape_a), Xb = ldx(geom_xform, shape_b);
    float rigid_gap = m->shape_gap[shape_a] + m->shape_gap[shape_b];
    if (is_infinite_plane_a) {
        // bounding-sphere half-space cull on the broad-phase AABB of the other shape (narrow_phase.py:1117-1194,
        // collision_core.py:549-560,628-683; external_aabb = True, speculative = False)
        vec3 lo = ld3(aabb_lower, shape_b), hi = ld3(aabb_upper, shape_b);
        vec3 bsphere_center_b = 0.5f * (lo + hi);
        float bsphere_radius_b = length(0.5f * (hi - lo));
        vec3 plane_normal = quat_rotate(Xa.q, vec3(0.0f, 0.0f, 1.0f));
        float center_dist = dot(bsphere_center_b - Xa.p, plane_normal);
        if (!(center_dist <= bsphere_radius_b)) return 0;
        if (ct->rigid_contact_count[0] >= ct->rigid_contact_max) return 0;
        // convert_infinite_plane_to_cube (collision_core.py:562-625)
        float other_radius = bsphere_radius_b + rigid_gap;
        float lateral_size = other_radius * 10.0f, depth = other_radius * 10.0f;
        P.ga.type = GEO_BOX;
        P.ga.scale = vec3(lateral_size, lateral_size, depth);
        P.ga.center = vec3(0.0f);
        vec3 to_other = Xb.p - Xa.p;
        float distance_along_normal = dot(to_other, plane_normal);
        vec3 plane_surface_point = Xb.p - plane_normal * distance_along_normal;
        Xa.p = plane_surface_point - plane_normal * depth;
    }
    if (ct->rigid_contact_count[0] >= ct->rigid_contact_max) return 0;  // find_contacts early-out

    // compute_gjk_mpr_contacts (collision_core.py:347-450)
    P.radius_eff_a = 0.0f;
    P.radius_eff_b = 0.0f;
    const float small_radius = 0.0001f;
    if (P.ga.type == GEO_SPHERE || P.ga.type == GEO_CAPSULE) {
        P.radius_eff_a = P.ga.scale.x;
        P.ga.scale.x = small_radius;
    }
    if (P.gb.type == GEO_SPHERE || P.gb.type == GEO_CAPSULE) {
        P.radius_eff_b = P.gb.scale.x;
        P.gb.scale.x = small_radius;
    }
    float contact_threshold = rigid_gap + P.radius_eff_a + P.radius_eff_b + P.margin_a + P.margin_b;
    bool skip_multi_contact =
        P.ga.type == GEO_SPHERE || P.gb.type == GEO_SPHERE || P.ga.type == GEO_ELLIPSOID || P.gb.type == GEO_ELLIPSOID;

    // solve_convex_multi_contact (collision_convex.py:131-230)
    quat orientation_a = Xa.q, orientation_b = Xb.q;
    vec3 position_a = Xa.p, position_b = Xb.p;
    quat rel_q = quat_inverse(orientation_a) * orientation_b;
    vec3 rel_p = quat_rotate_inv(orientation_a, position_b - position_a);
    float margin_sum = P.margin_a + P.margin_b;
    const float eps = 1.0e-4f;
    float enlarge = margin_sum <= 0.0f ? eps : (margin_sum < eps ? 2.0f * eps : 0.0f);
    vec3 point_a, point_b, normal;
    float penetration, signed_distance;
    bool collision = solve_mpr_core(P.ga, P.gb, rel_q, rel_p, enlarge, point_a, point_b, normal, penetration);
    if (collision) {
        signed_distance = -penetration + enlarge;
        float half_enlarge = enlarge * 0.5f;
        point_a = point_a - normal * half_enlarge;
        point_b = point_b + normal * half_enlarge;
    } else {
        solve_closest_distance_core(P.ga, P.gb, rel_q, rel_p, 0.0f, point_a, point_b, normal, signed_distance);
    }
    if (skip_multi_contact || signed_distance > contact_threshold) {
        ContactOut c;
        vec3 point = 0.5f * (point_a + point_b);
        c.center = quat_rotate(orientation_a, point) + position_a;
        c.normal = quat_rotate(orientation_a, normal);
        c.distance = signed_distance;
        emit(P, c, position_a, orientation_a, position_b, orientation_b);
        return P.written;
    }
    build_manifold(P, orientation_a, position_a, rel_q, rel_p, point_a, point_b, normal);
    return P.written;
}
}  // namespace orc

// ---------------------------------------------------------------------------------------------------------------------------
// The mesh-vs-convex leg (SURVEY.md section 8 rows a19 / a20 on triangle meshes):
//   midphase   collision_core.py:996-1180  _compute_mesh_vs_convex_query_aabb (compute_tight_aabb_from_support :452-548 in the scaled
//              mesh frame, aabb_to_unscaled :924-956, margin + gap per axis), the mesh query, _mesh_triangle_is_front_facing_local
//   contacts   contact_reduction_global.py:2299-2403 mesh_triangle_contacts_to_reducer_kernel (get_triangle_shape_from_mesh
//              collision_core.py:1218-1276, back-face culling, compute_gjk_mpr_contacts with the TRIANGLE support map and Minkowski
//              seed) written with write_contact_to_reducer (:2059-2096)
// Warp's BVH is native code: the query here tests every triangle's float32 bounds against the query box, ends inclusive (the set a
// BVH walk returns; triangle order does not matter to the consumers).  Partners: the primitives and CONVEX_MESH hulls.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
void tight_aabb_from_support(const Geom& g, quat orientation, vec3 center_pos, vec3& lo, vec3& hi) {
    mat33 rot_mat_t = transpose(quat_to_matrix(orientation));
    vec3 local_x(rot_mat_t(0, 0), rot_mat_t(1, 0), rot_mat_t(2, 0)), local_y(rot_mat_t(0, 1), rot_mat_t(1, 1), rot_mat_t(2, 1)),
        local_z(rot_mat_t(0, 2), rot_mat_t(1, 2), rot_mat_t(2, 2));
    if (g.type == GEO_CONVEX_MESH) {  // collision_core.py:491-523: one pass over the hull's vertices, axes pre-scaled
        vec3 scaled_x = cw_mul(local_x, g.scale), scaled_y = cw_mul(local_y, g.scale), scaled_z = cw_mul(local_z, g.scale);
        float min_x = 1.0e10f, max_x = -1.0e10f, min_y = 1.0e10f, max_y = -1.0e10f, min_z = 1.0e10f, max_z = -1.0e10f;
        for (int i = 0; i < g.count; ++i) {
            vec3 p = ld3(g.points, i);
            float vx = dot(p, scaled_x), vy = dot(p, scaled_y), vz = dot(p, scaled_z);
            min_x = fminw(min_x, vx); max_x = fmaxw(max_x, vx);
            min_y = fminw(min_y, vy); max_y = fmaxw(max_y, vy);
            min_z = fminw(min_z, vz); max_z = fmaxw(max_z, vz);
        }
        lo = vec3(min_x, min_y, min_z) + center_pos;
        hi = vec3(max_x, max_y, max_z) + center_pos;
        return;
    }
    float max_x = dot(local_x, support_map(g, local_x));
    float max_y = dot(local_y, support_map(g, local_y));
    float max_z = dot(local_z, support_map(g, local_z));
    float min_x = dot(local_x, support_map(g, -local_x));
    float min_y = dot(local_y, support_map(g, -local_y));
    float min_z = dot(local_z, support_map(g, -local_z));
    lo = vec3(min_x, min_y, min_z) + center_pos;
    hi = vec3(max_x, max_y, max_z) + center_pos;
}
}  // namespace

// -> number of buffered contacts (rows of `out` [cap][10]: mesh shape, convex shape, fingerprint, centre[3], normal[3], distance;
// counts past cap).  `tri_out` [tri_cap][3] receives the (mesh, convex, triangle) triples of the midphase, *n_tri their number.
extern "C" int o_mesh_triangle_contacts(int n_pairs, const int* pairs, const int* shape_type, const float* shape_transform,
                                        const float* shape_data, const float* shape_gap, const int* vertex_start,
                                        const int* tri_start, const int* tri_count, const float* vertices, const int* indices,
                                        const int* hull_start, const int* hull_count, const float* hull_points,
                                        const int* hf_index, const float* hf_table, const float* hf_elev, const float* aabb_lo,
                                        const float* aabb_hi, int* tri_out, int tri_cap, int* n_tri, float* out, int cap) {
    int nt = 0, nc = 0;
    // one (triangle | prism, convex) pair through compute_gjk_mpr_contacts, contacts appended to `out`
    auto gjk_mpr_contacts = [&](PairCtx& P, vec3 pos_a, quat quat_a, vec3 pos_b, quat quat_b, float rigid_gap, int sort_sub_key) {
        std::vector<float> raw;
        P.raw = &raw;
        P.sort_sub_key = sort_sub_key;
        P.radius_eff_a = 0.0f;
        P.radius_eff_b = 0.0f;
        const float small_radius = 0.0001f;
        if (P.gb.type == GEO_SPHERE || P.gb.type == GEO_CAPSULE) {
            P.radius_eff_b = P.gb.scale.x;
            P.gb.scale.x = small_radius;
        }
        float threshold = rigid_gap + P.radius_eff_a + P.radius_eff_b + P.margin_a + P.margin_b;
        bool skip_multi_contact = P.gb.type == GEO_SPHERE || P.gb.type == GEO_ELLIPSOID;
        quat rel_q = quat_inverse(quat_a) * quat_b;
        vec3 rel_p = quat_rotate_inv(quat_a, pos_b - pos_a);
        float margin_sum = P.margin_a + P.margin_b;
        const float e4 = 1.0e-4f;
        float enlarge = margin_sum <= 0.0f ? e4 : (margin_sum < e4 ? 2.0f * e4 : 0.0f);
        vec3 point_a, point_b, normal;
        float penetration, signed_distance;
        bool collision = solve_mpr_core(P.ga, P.gb, rel_q, rel_p, enlarge, point_a, point_b, normal, penetration);
        if (collision) {
            signed_distance = -penetration + enlarge;
            float half_enlarge = enlarge * 0.5f;
            point_a = point_a - normal * half_enlarge;
            point_b = point_b + normal * half_enlarge;
        } else {
            solve_closest_distance_core(P.ga, P.gb, rel_q, rel_p, 0.0f, point_a, point_b, normal, signed_distance);
        }
        if (skip_multi_contact || signed_distance > threshold) {
            ContactOut c;
            vec3 point = 0.5f * (point_a + point_b);
            c.center = quat_rotate(quat_a, point) + pos_a;
            c.normal = quat_rotate(quat_a, normal);
            c.distance = signed_distance;
            emit(P, c, pos_a, quat_a, pos_b, quat_b);
        } else {
            build_manifold(P, quat_a, pos_a, rel_q, rel_p, point_a, point_b, normal);
        }
        for (size_t r = 0; r + 8 <= raw.size(); r += 8) {
            if (nc < cap) {
                float* o = out + 10 * (size_t)nc;
                o[0] = (float)P.shape_a; o[1] = (float)P.shape_b; o[2] = raw[r + 7];
                for (int j = 0; j < 7; ++j) o[3 + j] = raw[r + j];
            }
            nc += 1;
        }
    };
    auto convex_geom = [&](int shape, Geom& g) {  // extract_shape_data (+ _shape_center of a hull); false: a hull without its table
        g.type = shape_type[shape];
        g.scale = vec3(shape_data[4 * shape], shape_data[4 * shape + 1], shape_data[4 * shape + 2]);
        if (g.type == GEO_CONVEX_MESH) {
            if (!hull_points || hull_count[shape] <= 0) return false;
            g.points = hull_points + 3 * hull_start[shape];
            g.count = hull_count[shape];
            vec3 first = cw_mul(ld3(g.points, 0), g.scale), lower = first, upper = first;
            for (int i = 1; i < g.count; ++i) {
                vec3 point = cw_mul(ld3(g.points, i), g.scale);
                lower = vmin(lower, point);
                upper = vmax(upper, point);
            }
            g.center = 0.5f * (lower + upper);
        }
        return true;
    };
    for (int k = 0; k < n_pairs; ++k) {
        int shape_a = pairs[2 * k], shape_b = pairs[2 * k + 1];
        if (hf_index && (shape_type[shape_a] == GEO_HFIELD || shape_type[shape_b] == GEO_HFIELD)) {
            // ---- heightfield vs convex (narrow_phase.py:553-583 routing; utils/heightfield.py:366-462 midphase, :280-363 cells)
            int hfield_shape = shape_type[shape_a] == GEO_HFIELD ? shape_a : shape_b;
            int other_shape = shape_type[shape_a] == GEO_HFIELD ? shape_b : shape_a;
            if (shape_type[other_shape] == GEO_HFIELD || shape_type[other_shape] == GEO_MESH || hf_index[hfield_shape] < 0) continue;
            const float* hd = hf_table + 7 * hf_index[hfield_shape];
            int data_offset = (int)hd[0], nrow = (int)hd[1], ncol = (int)hd[2];
            float hx = hd[3], hy = hd[4], min_z = hd[5], max_z = hd[6];
            transform X_hfield_ws = ldx(shape_transform, hfield_shape), X_other_ws = ldx(shape_transform, other_shape);
            transform X_other_in_hfield = transform_inverse(X_hfield_ws) * X_other_ws;
            vec3 other_pos = X_other_in_hfield.p;
            quat other_rot = X_other_in_hfield.q;
            vec3 local_lo = ld3(aabb_lo, other_shape), local_hi = ld3(aabb_hi, other_shape);
            vec3 local_center = 0.5f * (local_lo + local_hi), local_half = 0.5f * (local_hi - local_lo);
            vec3 center_in_hfield = quat_rotate(other_rot, local_center) + other_pos;
            vec3 r0 = quat_rotate(other_rot, vec3(1.0f, 0.0f, 0.0f)), r1 = quat_rotate(other_rot, vec3(0.0f, 1.0f, 0.0f)),
                 r2 = quat_rotate(other_rot, vec3(0.0f, 0.0f, 1.0f));
            vec3 half_in_hfield(std::fabs(r0.x) * local_half.x + std::fabs(r1.x) * local_half.y + std::fabs(r2.x) * local_half.z,
                                std::fabs(r0.y) * local_half.x + std::fabs(r1.y) * local_half.y + std::fabs(r2.y) * local_half.z,
                                std::fabs(r0.z) * local_half.x + std::fabs(r1.z) * local_half.y + std::fabs(r2.z) * local_half.z);
            float gap_sum = shape_gap[hfield_shape] + shape_gap[other_shape];
            float margin_sum = shape_data[4 * hfield_shape + 3] + shape_data[4 * other_shape + 3];
            float contact_threshold = gap_sum + margin_sum;
            vec3 threshold_vec(contact_threshold, contact_threshold, contact_threshold);
            vec3 q_lo = center_in_hfield - half_in_hfield - threshold_vec, q_hi = center_in_hfield + half_in_hfield + threshold_vec;
            float dx = 2.0f * hx / (float)(ncol - 1), dy = 2.0f * hy / (float)(nrow - 1);
            int col_min = std::max((int)std::floor((q_lo.x + hx) / dx), 0), col_max = std::min((int)std::floor((q_hi.x + hx) / dx), ncol - 2);
            int row_min = std::max((int)std::floor((q_lo.y + hy) / dy), 0), row_max = std::min((int)std::floor((q_hi.y + hy) / dy), nrow - 2);
            int cols = ncol - 1;
            Geom gb0;
            if (!convex_geom(other_shape, gb0)) continue;
            float z_range = max_z - min_z;
            for (int r = row_min; r <= row_max; ++r)
                for (int c = col_min; c <= col_max; ++c)
                    for (int tri_sub = 0; tri_sub < 2; ++tri_sub) {
                        int tri_idx = (r * cols + c) * 2 + tri_sub;
                        if (nt < tri_cap) { tri_out[3 * nt] = hfield_shape; tri_out[3 * nt + 1] = other_shape; tri_out[3 * nt + 2] = tri_idx; }
                        nt += 1;
                        // get_triangle_shape_from_heightfield
                        float x0 = -hx + (float)c * dx, x1 = x0 + dx, y0 = -hy + (float)r * dy, y1 = y0 + dy;
                        const float* e = hf_elev + data_offset;
                        float h00 = e[r * ncol + c], h10 = e[r * ncol + (c + 1)], h01 = e[(r + 1) * ncol + c], h11 = e[(r + 1) * ncol + (c + 1)];
                        float z00 = min_z + h00 * z_range, z10 = min_z + h10 * z_range, z01 = min_z + h01 * z_range, z11 = min_z + h11 * z_range;
                        vec3 p00(x0, y0, z00), p10(x1, y0, z10), p01(x0, y1, z01), p11(x1, y1, z11);
                        vec3 v0_local = p00, v1_local = tri_sub == 0 ? p10 : p11, v2_local = tri_sub == 0 ? p11 : p01;
                        PairCtx P;
                        P.m = nullptr; P.body_q = nullptr; P.ct = nullptr; P.shape_a = hfield_shape; P.shape_b = other_shape; P.written = 0;
                        P.ga.type = GEO_TRIANGLE_PRISM;
                        P.ga.scale = v1_local - v0_local;
                        P.ga.aux = v2_local - v0_local;
                        P.gb = gb0;
                        P.margin_a = shape_data[4 * hfield_shape + 3];
                        P.margin_b = shape_data[4 * other_shape + 3];
                        gjk_mpr_contacts(P, transform_point(X_hfield_ws, v0_local), X_hfield_ws.q, X_other_ws.p, X_other_ws.q, gap_sum, (tri_idx << 1) | 1);
                    }
            continue;
        }
        int mesh_shape, non_mesh_shape;
        if (shape_type[shape_a] == GEO_MESH && shape_type[shape_b] != GEO_MESH) { mesh_shape = shape_a; non_mesh_shape = shape_b; }
        else if (shape_type[shape_b] == GEO_MESH && shape_type[shape_a] != GEO_MESH) { mesh_shape = shape_b; non_mesh_shape = shape_a; }
        else continue;
        if (tri_count[mesh_shape] <= 0) continue;
        transform X_mesh_ws = ldx(shape_transform, mesh_shape), X_ws = ldx(shape_transform, non_mesh_shape);
        float gap_sum = shape_gap[non_mesh_shape] + shape_gap[mesh_shape];
        float margin_non_mesh = shape_data[4 * non_mesh_shape + 3], margin_mesh = shape_data[4 * mesh_shape + 3];
        float contact_threshold = gap_sum + margin_non_mesh + margin_mesh;
        // _compute_mesh_vs_convex_query_aabb
        transform X_mesh_sw = transform_inverse(X_mesh_ws);
        transform X_mesh_shape = X_mesh_sw * X_ws;
        vec3 pos_in_mesh = X_mesh_shape.p;
        Geom gq;
        gq.type = shape_type[non_mesh_shape];
        gq.scale = vec3(shape_data[4 * non_mesh_shape], shape_data[4 * non_mesh_shape + 1], shape_data[4 * non_mesh_shape + 2]);
        if (gq.type == GEO_CONVEX_MESH) {  // extract_shape_data: the hull's vertex table; _shape_center (support_function.py:448-464):
            if (!hull_points || hull_count[non_mesh_shape] <= 0) continue;  // the centre of its scaled bounds seeds MPR / GJK
            gq.points = hull_points + 3 * hull_start[non_mesh_shape];
            gq.count = hull_count[non_mesh_shape];
            vec3 first = cw_mul(ld3(gq.points, 0), gq.scale), lower = first, upper = first;
            for (int i = 1; i < gq.count; ++i) {
                vec3 point = cw_mul(ld3(gq.points, i), gq.scale);
                lower = vmin(lower, point);
                upper = vmax(upper, point);
            }
            gq.center = 0.5f * (lower + upper);
        }
        vec3 aabb_lower, aabb_upper;
        tight_aabb_from_support(gq, X_mesh_shape.q, pos_in_mesh, aabb_lower, aabb_upper);
        vec3 mesh_scale(shape_data[4 * mesh_shape], shape_data[4 * mesh_shape + 1], shape_data[4 * mesh_shape + 2]);
        const float eps = 1.0e-12f;
        auto guarded = [&](float s) { return std::fabs(s) > eps ? s : (s >= 0.0f ? eps : -eps); };
        vec3 inv_scale(1.0f / guarded(mesh_scale.x), 1.0f / guarded(mesh_scale.y), 1.0f / guarded(mesh_scale.z));
        vec3 l0 = cw_mul(aabb_lower, inv_scale), l1 = cw_mul(aabb_upper, inv_scale);
        vec3 q_lo = vmin(l0, l1), q_hi = vmax(l0, l1);
        vec3 margin_vec(contact_threshold / fmaxw(std::fabs(mesh_scale.x), 1.0e-12f), contact_threshold / fmaxw(std::fabs(mesh_scale.y), 1.0e-12f),
                        contact_threshold / fmaxw(std::fabs(mesh_scale.z), 1.0e-12f));
        q_lo = q_lo - margin_vec;
        q_hi = q_hi + margin_vec;
        vec3 center_in_bvh = cw_mul(pos_in_mesh, inv_scale);
        const float* pts = vertices + 3 * vertex_start[mesh_shape];
        const int* idx = indices + 3 * tri_start[mesh_shape];
        // shape B of every triangle pair (extract_shape_data)
        Geom gb0 = gq;
        float margin_offset_b = margin_non_mesh, margin_offset_a = margin_mesh;
        for (int t = 0; t < tri_count[mesh_shape]; ++t) {
            int idx0 = idx[3 * t], idx1 = idx[3 * t + 1], idx2 = idx[3 * t + 2];
            vec3 v0 = ld3(pts, idx0), v1 = ld3(pts, idx1), v2 = ld3(pts, idx2);
            vec3 tlo = vmin(v0, vmin(v1, v2)), thi = vmax(v0, vmax(v1, v2));
            if (tlo.x > q_hi.x || tlo.y > q_hi.y || tlo.z > q_hi.z || thi.x < q_lo.x || thi.y < q_lo.y || thi.z < q_lo.z) continue;
            // _mesh_triangle_is_front_facing_local (unscaled mesh frame, the stored winding)
            vec3 face_normal_l = cross(v1 - v0, v2 - v0);
            if (dot(face_normal_l, center_in_bvh - v0) < 0.0f) continue;
            if (nt < tri_cap) { tri_out[3 * nt] = mesh_shape; tri_out[3 * nt + 1] = non_mesh_shape; tri_out[3 * nt + 2] = t; }
            nt += 1;
            // get_triangle_shape_from_mesh
            if (mesh_scale.x * mesh_scale.y * mesh_scale.z < 0.0f) std::swap(idx1, idx2);
            vec3 v0_world = transform_point(X_mesh_ws, cw_mul(ld3(pts, idx0), mesh_scale));
            vec3 v1_world = transform_point(X_mesh_ws, cw_mul(ld3(pts, idx1), mesh_scale));
            vec3 v2_world = transform_point(X_mesh_ws, cw_mul(ld3(pts, idx2), mesh_scale));
            PairCtx P;
            P.m = nullptr; P.body_q = nullptr; P.ct = nullptr; P.shape_a = mesh_shape; P.shape_b = non_mesh_shape; P.written = 0;
            P.ga.type = GEO_TRIANGLE;
            P.ga.scale = v1_world - v0_world;
            P.ga.aux = v2_world - v0_world;
            P.gb = gb0;
            vec3 pos_a = v0_world, pos_b = X_ws.p;
            quat quat_a = quat_identity(), quat_b = X_ws.q;
            // back-face culling (contact_reduction_global.py:2368-2375)
            vec3 face_normal = cross(P.ga.scale, P.ga.aux);
            if (dot(face_normal, pos_b - pos_a) < 0.0f) continue;
            std::vector<float> raw;
            P.raw = &raw;
            P.sort_sub_key = (t << 1) | 1;
            P.margin_a = margin_offset_a;
            P.margin_b = margin_offset_b;
            // compute_gjk_mpr_contacts + solve_convex_multi_contact (as convex_pair_contacts above)
            P.radius_eff_a = 0.0f;
            P.radius_eff_b = 0.0f;
            const float small_radius = 0.0001f;
            if (P.gb.type == GEO_SPHERE || P.gb.type == GEO_CAPSULE) {
                P.radius_eff_b = P.gb.scale.x;
                P.gb.scale.x = small_radius;
            }
            float rigid_gap = shape_gap[mesh_shape] + shape_gap[non_mesh_shape];
            float threshold = rigid_gap + P.radius_eff_a + P.radius_eff_b + P.margin_a + P.margin_b;
            bool skip_multi_contact = P.gb.type == GEO_SPHERE || P.gb.type == GEO_ELLIPSOID;
            quat rel_q = quat_inverse(quat_a) * quat_b;
            vec3 rel_p = quat_rotate_inv(quat_a, pos_b - pos_a);
            float margin_sum = P.margin_a + P.margin_b;
            const float e4 = 1.0e-4f;
            float enlarge = margin_sum <= 0.0f ? e4 : (margin_sum < e4 ? 2.0f * e4 : 0.0f);
            vec3 point_a, point_b, normal;
            float penetration, signed_distance;
            bool collision = solve_mpr_core(P.ga, P.gb, rel_q, rel_p, enlarge, point_a, point_b, normal, penetration);
            if (collision) {
                signed_distance = -penetration + enlarge;
                float half_enlarge = enlarge * 0.5f;
                point_a = point_a - normal * half_enlarge;
                point_b = point_b + normal * half_enlarge;
            } else {
                solve_closest_distance_core(P.ga, P.gb, rel_q, rel_p, 0.0f, point_a, point_b, normal, signed_distance);
            }
            if (skip_multi_contact || signed_distance > threshold) {
                ContactOut c;
                vec3 point = 0.5f * (point_a + point_b);
                c.center = quat_rotate(quat_a, point) + pos_a;
                c.normal = quat_rotate(quat_a, normal);
                c.distance = signed_distance;
                emit(P, c, pos_a, quat_a, pos_b, quat_b);
            } else {
                build_manifold(P, quat_a, pos_a, rel_q, rel_p, point_a, point_b, normal);
            }
            for (size_t r = 0; r + 8 <= raw.size(); r += 8) {
                if (nc < cap) {
                    float* o = out + 10 * (size_t)nc;
                    o[0] = (float)mesh_shape; o[1] = (float)non_mesh_shape; o[2] = raw[r + 7];
                    for (int j = 0; j < 7; ++j) o[3 + j] = raw[r + j];
                }
                nc += 1;
            }
        }
    }
    *n_tri = nt;
    return nc;
}

// probes for the known-answer tests (reference: newton/tests/test_mpr.py, test_gjk.py)
extern "C" int o_probe_mpr(int type_a, int type_b, const float* xf_a, const float* xf_b, const float* scale_a, const float* scale_b,
                           float* out /* collision, signed_distance, point[3], normal[3] (world) */) {
    Geom ga, gb;
    ga.type = type_a; ga.scale = ld3(scale_a, 0);
    gb.type = type_b; gb.scale = ld3(scale_b, 0);
    transform Xa = ldx(xf_a, 0), Xb = ldx(xf_b, 0);
    quat rel_q = quat_inverse(Xa.q) * Xb.q;
    vec3 rel_p = quat_rotate_inv(Xa.q, Xb.p - Xa.p);
    vec3 pa, pb, n;
    float pen;
    bool col = solve_mpr_core(ga, gb, rel_q, rel_p, 0.0f, pa, pb, n, pen);
    vec3 point = quat_rotate(Xa.q, 0.5f * (pa + pb)) + Xa.p;
    vec3 nw = quat_rotate(Xa.q, n);
    out[0] = col ? 1.0f : 0.0f;
    out[1] = -pen;
    out[2] = point.x; out[3] = point.y; out[4] = point.z;
    out[5] = nw.x; out[6] = nw.y; out[7] = nw.z;
    return col ? 1 : 0;
}
extern "C" int o_probe_gjk(int type_a, int type_b, const float* xf_a, const float* xf_b, const float* scale_a, const float* scale_b,
                           float* out /* collision, distance, point[3], normal[3] (world) */) {
    Geom ga, gb;
    ga.type = type_a; ga.scale = ld3(scale_a, 0);
    gb.type = type_b; gb.scale = ld3(scale_b, 0);
    transform Xa = ldx(xf_a, 0), Xb = ldx(xf_b, 0);
    quat rel_q = quat_inverse(Xa.q) * Xb.q;
    vec3 rel_p = quat_rotate_inv(Xa.q, Xb.p - Xa.p);
    vec3 pa, pb, n;
    float dist;
    bool separated = solve_closest_distance_core(ga, gb, rel_q, rel_p, 0.0f, pa, pb, n, dist);
    vec3 point = quat_rotate(Xa.q, 0.5f * (pa + pb)) + Xa.p;
    vec3 nw = quat_rotate(Xa.q, n);
    out[0] = separated ? 0.0f : 1.0f;
    out[1] = dist;
    out[2] = point.x; out[3] = point.y; out[4] = point.z;
    out[5] = nw.x; out[6] = nw.y; out[7] = nw.z;
    return separated ? 0 : 1;
}
