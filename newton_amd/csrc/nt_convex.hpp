// Convex pair narrow phase for gfx950: MPR -> GJK fallback -> 5-direction manifold with polygon clipping.
// Device restatement of the reference's convex path (one lane works one candidate pair):
//   support_map / centred-tie box support            newton/_src/geometry/support_function.py:131-447
//   solve_mpr_core                                    newton/_src/geometry/mpr.py:186-403
//   solve_closest_distance_core (GJK)                 newton/_src/geometry/simplex_solver.py:44-469
//   solve_convex_multi_contact                        newton/_src/geometry/collision_convex.py:110-232
//   build_manifold + clipping helpers                 newton/_src/geometry/multicontact.py:28-956
//   post_process_axial_on_discrete_contact            newton/_src/geometry/collision_core.py:173-278
//   write_contact gap test                            newton/_src/sim/collide.py:206-254
// Scope: BOX, SPHERE, CAPSULE, ELLIPSOID, CYLINDER (straight and barrel), CONE, CONVEX_MESH, finite PLANE.
#pragma once
#include "nt_math.hpp"
#include "nt_primitives.hpp"

namespace nt {

#define NT_DEV __device__
#ifndef NT_HULL_BATCH
#define NT_HULL_BATCH 8  // hull vertices fetched per round of the CONVEX_MESH support scan (support_map)
#endif
NT_DI int imin(int a, int b) { return a < b ? a : b; }
// The mesh-triangle leg (nt_mesh_triangle.hip) defines NT_CONVEX_WITH_TRIANGLES = 1 before including this header: GeoTypeEx.TRIANGLE
// as shape A (support map, Minkowski seed, is_discrete_shape) and contacts buffered without the writer's gap test
// (write_contact_to_reducer).  Every other translation unit compiles the branches away.
#ifndef NT_CONVEX_WITH_TRIANGLES
#define NT_CONVEX_WITH_TRIANGLES 0
#endif
constexpr int GEO_TRIANGLE = 1000;  // support_function.py:58: vertex A at the origin, B - A in `scale`, C - A in `aux`
constexpr int GEO_TRIANGLE_PRISM = 1001;  // :59,193-200: a heightfield cell's triangle extruded 1 m along -Z of the heightfield frame

struct Geom {
    int type;
    vec3 scale;
    const float* points;  // CONVEX_MESH: vertex slice [count][3] (unscaled, env-uniform, global memory)
    int count;
    vec3 center;          // interior point that seeds MPR / GJK (collision_core.py:690, narrow_phase.py:1102-1105)
#if NT_CONVEX_WITH_TRIANGLES
    vec3 aux;             // TRIANGLE: C - A (GenericShapeData.auxiliary)
#endif
    NT_DI Geom() : type(0), points(nullptr), count(0) {}
};
struct vec2 {
    float x, y;
    NT_DI vec2() : x(0.f), y(0.f) {}
    NT_DI vec2(float a, float b) : x(a), y(b) {}
};
NT_DI vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
NT_DI vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
NT_DI vec2 operator*(float s, vec2 a) { return vec2(a.x * s, a.y * s); }
NT_DEV float length_sq2(vec2 a) { return a.x * a.x + a.y * a.y; }

// strided view of a 10 x vec2 polygon buffer in LDS: element i = (base[2 i stride], base[(2 i + 1) stride])
struct PolyRef {
    float* base;
    int stride;
    NT_DI vec2 get(int i) const { return vec2(base[(2 * i) * stride], base[(2 * i + 1) * stride]); }
    NT_DI void set(int i, vec2 v) const {
        base[(2 * i) * stride] = v.x;
        base[(2 * i + 1) * stride] = v.y;
    }
    NT_DI PolyRef operator+(int k) const { return PolyRef{base + 2 * k * stride, stride}; }
};

struct Vert {
    vec3 B, BtoA;
};
NT_DEV vec3 vert_a(const Vert& v) { return v.B + v.BtoA; }

// support_function.py:44-53 (CPU branch)
NT_DEV float support_rsqrt_rn(float v) { return 1.0f / sqrtf(v); }

// support_function.py:120-128
NT_DEV vec3 support_map_box(const Geom& g, vec3 d) {
    float ds = fmaxw(fabsf(d.x), fmaxw(fabsf(d.y), fabsf(d.z)));
    float threshold = 1.0e-10f * ds;
    float sx = d.x >= -threshold ? 1.0f : -1.0f;
    float sy = d.y >= -threshold ? 1.0f : -1.0f;
    float sz = d.z >= -threshold ? 1.0f : -1.0f;
    return vec3(sx * g.scale.x, sy * g.scale.y, sz * g.scale.z);
}

// support_function.py:334-345: finite plane = rectangle in XY (half-width scale.x, half-length scale.y), normal +Z
NT_DEV vec3 support_map_plane(vec3 half, vec3 direction) {
    float sx = direction.x >= 0.0f ? 1.0f : -1.0f;
    float sy = direction.y >= 0.0f ? 1.0f : -1.0f;
    return vec3(sx * half.x, sy * half.y, 0.0f);
}

// support_function.py:152-171: furthest vertex of a convex hull; ties keep the first one.
// NT_HULL_BATCH vertices per round: their loads are issued together (one memory round trip per round instead of one per vertex -- the
// scan is a chain of dependent global loads otherwise); the comparisons stay in ascending vertex order, so the first furthest vertex
// wins as in the serial loop.  Rounds past the end re-read the last vertex: its dot product cannot beat the maximum it already
// entered.  The winner's coordinates travel with the maximum (no reload by index at the end).
#ifdef NT_HULL_NOINLINE  // measurement builds: the scan as a call (its 24 vertex registers out of the MPR loops of box pairs)
__device__ __attribute__((noinline))
#else
NT_DEV
#endif
vec3 support_map_hull(const float* points, int count, vec3 scale, vec3 direction) {
    vec3 result(0.0f);
    vec3 scaled_dir = cw_mul(direction, scale);
    float max_dot = -1.0e10f;
    vec3 best;
    if (count > 0) best = vec3(points[0], points[1], points[2]);  // best_idx = 0 unless a vertex beats -1e10
    const int last = count - 1;
    for (int i0 = 0; i0 < count; i0 += NT_HULL_BATCH) {
        vec3 p[NT_HULL_BATCH];
#pragma unroll
        for (int q = 0; q < NT_HULL_BATCH; ++q) {
            const int i = imin(i0 + q, last);
            p[q] = vec3(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        }
#pragma unroll
        for (int q = 0; q < NT_HULL_BATCH; ++q) {
            float dot_val = dot(p[q], scaled_dir);
            if (dot_val > max_dot) {
                max_dot = dot_val;
                best = p[q];
            }
        }
    }
    if (count > 0) result = cw_mul(best, scale);
    return result;
}

// barrel cylinder (support_function.py:284-305): the side profile is a circular arc of radius barrel_radius about the axis; the
// support point sits where the arc's normal matches the direction, clamped to the end caps
// A CALL, not inlined: its three IEEE square roots and divisions inlined into every support query of the MPR / GJK / manifold loops
// cost the convex rollouts 10 % (592 instead of 256 bytes of scratch per lane; profiles/r05K_ab.txt) -- shapes that are not barrels
// never take the branch.
__device__ __attribute__((noinline)) vec3 support_map_barrel(float radius, float half_height, float barrel_radius, vec3 direction) {
    const float eps = 1.0e-12f;
    vec3 dir_xy(direction.x, direction.y, 0.0f);
    float l2 = length_sq(dir_xy);
    vec3 n_xy(1.0f, 0.0f, 0.0f);
    if (l2 > eps) n_xy = dir_xy / sqrtf(l2);
    float direction_len = sqrtf(l2 + direction.z * direction.z);
    float support_z = 0.0f;
    if (direction_len > eps) support_z = clampf(barrel_radius * direction.z / direction_len, -half_height, half_height);
    float barrel_radius_sq = barrel_radius * barrel_radius, half_height_sq = half_height * half_height;
    float support_z_sq = support_z * support_z;
    float end_offset = sqrtf(barrel_radius_sq - half_height_sq);
    float support_offset = sqrtf(fmaxw(barrel_radius_sq - support_z_sq, 0.0f));
    float offset_sum = support_offset + end_offset;
    float support_radius = radius;
    if (offset_sum > eps) support_radius += (half_height_sq - support_z_sq) / offset_sum;
    return vec3(n_xy.x * support_radius, n_xy.y * support_radius, support_z);
}

// the curved primitives of support_function.py:173-350 (sphere, capsule, ellipsoid, cylinder, cone)
#ifdef NT_SUPPORT_REST_NOINLINE  // measurement builds: a call (box pairs never take it)
__device__ __attribute__((noinline))
#else
NT_DEV
#endif
vec3 support_map_rest(int type, vec3 scale, vec3 direction) {
    const float eps = 1.0e-12f;
    vec3 result(0.0f);
    if (type == GEO_SPHERE) {
        float radius = scale.x;
        float l2 = length_sq(direction);
        vec3 n = l2 > eps ? direction * support_rsqrt_rn(l2) : vec3(1.0f, 0.0f, 0.0f);
        result = n * radius;
    } else if (type == GEO_CAPSULE) {
        float radius = scale.x, half_height = scale.y;
        float l2 = length_sq(direction);
        vec3 n = l2 > eps ? direction * support_rsqrt_rn(l2) : vec3(1.0f, 0.0f, 0.0f);
        result = n * radius;
        if (direction.z >= 0.0f) result = result + vec3(0.0f, 0.0f, half_height);
        else result = result + vec3(0.0f, 0.0f, -half_height);
    } else if (type == GEO_ELLIPSOID) {
        float a = scale.x, b = scale.y, c = scale.z;
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
    } else if (type == GEO_CYLINDER) {
        float radius = scale.x, half_height = scale.y, barrel_radius = scale.z;
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
            result = support_map_barrel(radius, half_height, barrel_radius, direction);
        }
    } else if (type == GEO_CONE) {
        float radius = scale.x, half_height = scale.y;
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

#if NT_CONVEX_WITH_TRIANGLES
// support_function.py:174-191: the vertex furthest along the direction; ties prefer a, then b
NT_DEV vec3 support_map_triangle(const Geom& g, vec3 direction) {
    const vec3 tri_a(0.0f), tri_b = g.scale, tri_c = g.aux;
    const float dot_a = dot(tri_a, direction), dot_b = dot(tri_b, direction), dot_c = dot(tri_c, direction);
    vec3 result = tri_c;
    if (dot_a >= dot_b && dot_a >= dot_c) result = tri_a;
    else if (dot_b >= dot_c) result = tri_b;
    if (g.type == GEO_TRIANGLE_PRISM && direction.z < 0.0f) result = result + vec3(0.0f, 0.0f, -1.0f);  // (:193-200)
    return result;
}
// support_function.py:647-745
NT_DEV vec3 closest_point_on_triangle(vec3 p, vec3 tri_a, vec3 tri_b, vec3 tri_c) {
    const vec3 ab = tri_b - tri_a, ac = tri_c - tri_a;
    const float ab_sq = dot(ab, ab), ac_sq = dot(ac, ac);
    const float EPS2 = 1.0e-20f;
    const vec3 triangle_normal = cross(ab, ac);
    if (dot(triangle_normal, triangle_normal) < EPS2) {
        const vec3 bc = tri_c - tri_b;
        const float bc_sq = dot(bc, bc);
        if (ab_sq >= ac_sq && ab_sq >= bc_sq) {
            if (ab_sq < EPS2) return tri_a;
            const float t = clampf(dot(p - tri_a, ab) / ab_sq, 0.0f, 1.0f);
            return tri_a + t * ab;
        } else if (ac_sq >= bc_sq) {
            const float t = clampf(dot(p - tri_a, ac) / ac_sq, 0.0f, 1.0f);
            return tri_a + t * ac;
        } else {
            const float t = clampf(dot(p - tri_b, bc) / bc_sq, 0.0f, 1.0f);
            return tri_b + t * bc;
        }
    }
    const vec3 ap = p - tri_a;
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) return tri_a;
    const vec3 bp = p - tri_b;
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) return tri_b;
    const vec3 cp = p - tri_c;
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) return tri_c;
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        const float v = d1 / (d1 - d3);
        return tri_a + v * ab;
    }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        const float w = d2 / (d2 - d6);
        return tri_a + w * ac;
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        return tri_b + w * (tri_c - tri_b);
    }
    const float denom = 1.0f / (va + vb + vc);
    const float v = vb * denom, w = vc * denom;
    return tri_a + v * ab + w * ac;
}
// support_function.py:467-502: a triangle's Minkowski seed is the point of the triangle nearest B's centre, nudged to the centroid
NT_DEV vec3 adjust_minkowski_center(const Geom& ga, vec3 center_b_world, vec3 center_b_to_a) {
    if (ga.type != GEO_TRIANGLE && ga.type != GEO_TRIANGLE_PRISM) return center_b_to_a;
    const vec3 tri_a(0.0f), tri_b = ga.scale, tri_c = ga.aux;
    const vec3 face_normal = cross(tri_b - tri_a, tri_c - tri_a);
    const float face_normal_length_sq = length_sq(face_normal);
    vec3 projection = closest_point_on_triangle(center_b_world, tri_a, tri_b, tri_c);
    if (face_normal_length_sq < 1.0e-20f) return projection - center_b_world;
    const vec3 face_normal_unit = face_normal / sqrtf(face_normal_length_sq);
    const float signed_plane_distance = dot(center_b_world - tri_a, face_normal_unit);
    const vec3 plane_projection = center_b_world - signed_plane_distance * face_normal_unit;
    const bool inside_face = dot(cross(tri_b - tri_a, plane_projection - tri_a), face_normal) >= 0.0f &&
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
    const float distance_to_centroid = length(to_centroid);
    if (distance_to_centroid > 1.0e-12f) {
        const float nudge_distance = 0.01f * fminw(distance_to_centroid, fabsf(signed_plane_distance));
        center_b_to_a = center_b_to_a + to_centroid * (nudge_distance / distance_to_centroid);
    }
    return center_b_to_a;
}
// support_function.py:505-538
NT_DEV vec3 minkowski_center_fallback(const Geom& ga, vec3 center_b_world) {
    if (ga.type != GEO_TRIANGLE && ga.type != GEO_TRIANGLE_PRISM) return vec3(0.0f);
    const vec3 tri_a(0.0f), tri_b = ga.scale, tri_c = ga.aux;
    vec3 face_normal = cross(tri_b - tri_a, tri_c - tri_a);
    const float face_normal_length_sq = length_sq(face_normal);
    if (face_normal_length_sq < 1.0e-20f) return vec3(0.0f);
    face_normal = face_normal / sqrtf(face_normal_length_sq);
    const vec3 projection = closest_point_on_triangle(center_b_world, tri_a, tri_b, tri_c);
    vec3 to_centroid = (tri_a + tri_b + tri_c) / 3.0f - projection;
    to_centroid = to_centroid - dot(to_centroid, face_normal) * face_normal;
    const float to_centroid_length_sq = length_sq(to_centroid);
    vec3 fallback_direction = -face_normal;
    if (dot(center_b_world - projection, face_normal) < 0.0f) fallback_direction = face_normal;
    if (to_centroid_length_sq > 1.0e-20f) fallback_direction = fallback_direction + 0.01f * to_centroid / sqrtf(to_centroid_length_sq);
    return normalize(fallback_direction) * 1.0e-5f;
}
#endif

// support_function.py:131-350
NT_DEV vec3 support_map(const Geom& g, vec3 direction) {
#if NT_CONVEX_WITH_TRIANGLES
    if (g.type == GEO_TRIANGLE || g.type == GEO_TRIANGLE_PRISM) return support_map_triangle(g, direction);
#endif
    if (g.type == GEO_PLANE) return support_map_plane(g.scale, direction);
    if (g.type == GEO_CONVEX_MESH) return support_map_hull(g.points, g.count, g.scale, direction);
    if (g.type == GEO_BOX) return support_map_box(g, direction);
    return support_map_rest(g.type, g.scale, direction);
}

// create_shape_support_function(center_ties=True), support_function.py:399-431
NT_DEV vec3 shape_support_centered(const Geom& g, vec3 direction) {
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

// mpr.py:100-160; CENTERED selects the tie-centred box support used by MPR's own support map
template <bool CENTERED>
NT_DEV Vert minkowski_support(const Geom& ga, const Geom& gb, vec3 direction, quat orientation_b, vec3 position_b, float extend) {
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
NT_DEV bool solve_mpr_core(const Geom& ga, const Geom& gb, quat orientation_b, vec3 position_b, float extend, vec3& point_a, vec3& point_b,
                    vec3& normal, float& penetration) {
    const int MAX_ITER = 30;
    const float COLLIDE_EPSILON = 1e-5f;
    const float NUMERIC_EPSILON = 1e-16f;
    penetration = 0.0f;
    point_a = vec3(0.0f);
    point_b = vec3(0.0f);
    Vert v0;  // create_shape_center_function(use_precomputed_center=True) (support_function.py:541-598)
    v0.B = position_b + quat_rotate(orientation_b, gb.center);
    v0.BtoA = ga.center - v0.B;
#if NT_CONVEX_WITH_TRIANGLES
    v0.BtoA = adjust_minkowski_center(ga, v0.B, v0.BtoA);
#endif
    normal = v0.BtoA;
    if (length_sq(normal) < NUMERIC_EPSILON) {
        v0.BtoA = vec3(0.0f);  // fallback() is zero for non-triangle shapes
#if NT_CONVEX_WITH_TRIANGLES
        v0.BtoA = minkowski_center_fallback(ga, v0.B);
#endif
        if (length_sq(v0.BtoA) < NUMERIC_EPSILON) {
            float best_dot = -1.0e30f;
            vec3 best_dir(1.0f, 0.0f, 0.0f);
            for (int axis_idx = 0; axis_idx < 3; ++axis_idx) {
                vec3 probe(0.0f);
                vset(probe, axis_idx, 1.0f);
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
        Vert vt = v1; v1 = v2; v2 = vt;
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
                float inv_normal = 1.0f / sqrtf(normal_sq);
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
// up to four Minkowski vertices (B and B->A of each).  Named fields with select-style access: the slots are addressed with
// run-time indices, and an indexed private array would live in scratch memory
struct Simplex {
    vec3 b0, a0, b1, a1, b2, a2, b3, a3;
    NT_DI vec3 B(int i) const { return vsel(i == 0, b0, vsel(i == 1, b1, vsel(i == 2, b2, b3))); }
    NT_DI vec3 BtoA(int i) const { return vsel(i == 0, a0, vsel(i == 1, a1, vsel(i == 2, a2, a3))); }
    NT_DI void set(int i, vec3 B_, vec3 BtoA_) {
        b0 = vsel(i == 0, B_, b0); a0 = vsel(i == 0, BtoA_, a0);
        b1 = vsel(i == 1, B_, b1); a1 = vsel(i == 1, BtoA_, a1);
        b2 = vsel(i == 2, B_, b2); a2 = vsel(i == 2, BtoA_, a2);
        b3 = vsel(i == 3, B_, b3); a3 = vsel(i == 3, BtoA_, a3);
    }
};
struct vec4f4 {
    float c0, c1, c2, c3;
    NT_DI vec4f4() : c0(0.0f), c1(0.0f), c2(0.0f), c3(0.0f) {}
    NT_DI float get(int i) const { return fsel(i == 0, c0, fsel(i == 1, c1, fsel(i == 2, c2, c3))); }
    NT_DI void set(int i, float v) {
        c0 = fsel(i == 0, v, c0); c1 = fsel(i == 1, v, c1); c2 = fsel(i == 2, v, c2); c3 = fsel(i == 3, v, c3);
    }
};

NT_DEV void closest_segment(const Simplex& s, int i0, int i1, vec3& closest, vec4f4& bc, unsigned& mask) {
    vec3 a = s.BtoA(i0), b = s.BtoA(i1);
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
    bc.set(i0, lambda0);
    bc.set(i1, lambda1);
    closest = lambda0 * a + lambda1 * b;
}

NT_DEV void closest_triangle(const Simplex& s, int i0, int i1, int i2, vec3& closest_out, vec4f4& bc_out, unsigned& mask_out) {
    vec3 a = s.BtoA(i0), b = s.BtoA(i1), c = s.BtoA(i2);
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
    unsigned mask = 0;
    vec3 cl;
    vec4f4 bt;
    unsigned mm;
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
    bc.set(i0, lambda0);
    bc.set(i1, lambda1);
    bc.set(i2, lambda2);
    mask_out = (1u << i0) | (1u << i1) | (1u << i2);
    bc_out = bc;
    closest_out = lambda0 * a + lambda1 * b + lambda2 * c;
}

NT_DEV float determinant(vec3 a, vec3 b, vec3 c, vec3 d) { return dot(b - a, cross(c - a, d - a)); }

NT_DEV void closest_tetrahedron(const Simplex& s, vec3& closest_out, vec4f4& bc_out, unsigned& mask_out) {
    vec3 v0 = s.a0, v1 = s.a1, v2 = s.a2, v3 = s.a3;
    float det_t = determinant(v0, v1, v2, v3);
    bool degenerate = fabsf(det_t) < GJK_EPSILON;
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
    unsigned mask = 0;
    vec3 cl;
    vec4f4 bt;
    unsigned mm;
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
    bc.c0 = lambda0; bc.c1 = lambda1; bc.c2 = lambda2; bc.c3 = lambda3;
    bc_out = bc;
    mask_out = 15u;
    closest_out = zero;
}

NT_DEV void simplex_get_closest(const Simplex& s, const vec4f4& bc, unsigned mask, vec3& point_a, vec3& point_b) {
    point_a = vec3(0.0f);
    point_b = vec3(0.0f);
    for (int i = 0; i < 4; ++i) {
        if ((mask & (1u << i)) == 0) continue;
        vec3 B = s.B(i), BtoA = s.BtoA(i);
        float w = bc.get(i);
        point_a = point_a + w * (B + BtoA);
        point_b = point_b + w * B;
    }
}

// simplex_solver.py:331-469; returns `separated`
NT_DEV bool solve_closest_distance_core(const Geom& ga, const Geom& gb, quat orientation_b, vec3 position_b, float extend, vec3& point_a,
                                 vec3& point_b, vec3& normal, float& distance) {
    const int MAX_ITER = 30;
    const float COLLIDE_EPSILON = 1e-4f;
    distance = 0.0f;
    point_a = vec3(0.0f);
    point_b = vec3(0.0f);
    normal = vec3(0.0f);
    Simplex simplex;
    vec4f4 bary;
    unsigned usage = 0;
    int iter_count = MAX_ITER;
    vec3 v = ga.center - (position_b + quat_rotate(orientation_b, gb.center));  // center.BtoA
#if NT_CONVEX_WITH_TRIANGLES
    v = adjust_minkowski_center(ga, position_b + quat_rotate(orientation_b, gb.center), v);
#endif
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
                if (length_sq(simplex.BtoA(i) - w_v) < COLLIDE_EPSILON * COLLIDE_EPSILON) {
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
        simplex.set(free_slot, w.B, w.BtoA);
        vec3 closest(0.0f);
        bool success = true;
        if (use_count == 1) {
            int i0 = indices[0];
            closest = simplex.BtoA(i0);
            usage = 1u << i0;
            bary.set(i0, 1.0f);
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
        distance = sqrtf(delta_len_sq);
        normal = delta * (1.0f / distance);
    } else {
        distance = sqrtf(dist_sq);
        if (distance > COLLIDE_EPSILON) {
            normal = v * (-1.0f / distance);
        } else {
            float nsq = length_sq(last_search_dir);
            if (nsq > 0.0f) normal = last_search_dir * (1.0f / sqrtf(nsq));
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

NT_DEV float signed_area(vec2 a, vec2 b, vec2 q) { return (b.x - a.x) * (q.y - a.y) - (b.y - a.y) * (q.x - a.x); }

NT_DEV vec3 ray_plane_intersection(vec3 ro, vec3 rd, float plane_d, vec3 plane_normal) {
    float denom = dot(rd, plane_normal);
    if (fabsf(denom) < 1.0e-12f) return ro;
    float t = -(dot(ro, plane_normal) + plane_d) / denom;
    return ro + rd * t;
}
struct BodyProjector {
    float plane_d;
    vec3 normal;
    NT_DI BodyProjector() : plane_d(0.0f) {}
};
struct PlaneTracker {
    vec3 reference_point, previous_point, normal;
    float largest_area_sq;
    NT_DI PlaneTracker() : largest_area_sq(0.0f) {}
};
NT_DEV void update_tracker(PlaneTracker& t, vec3 p, int id) {
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
NT_DEV vec3 line_segment_projector_normal(vec3 segment_dir, vec3 reference_normal) {
    vec3 right = cross(segment_dir, reference_normal);
    vec3 n = cross(right, segment_dir);
    float len = length(n);
    return len > 1.0e-12f ? n * (1.0f / len) : reference_normal;
}
NT_DEV void create_body_projectors(const PlaneTracker& ta, vec3 anchor_a, const PlaneTracker& tb, vec3 anchor_b, vec3 contact_normal,
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
        float inv = 1.0f / sqrtf(fmaxw(1.0e-12f, ta.largest_area_sq));
        pa.normal = ta.normal * inv;
        pa.plane_d = -dot(anchor_a, pa.normal);
    }
    if (tb.largest_area_sq > 0.0f) {
        float inv = 1.0f / sqrtf(fmaxw(1.0e-12f, tb.largest_area_sq));
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

NT_DEV vec2 intersection_point(vec2 s0, vec2 s1, vec2 a, vec2 b) {
    float sa = signed_area(s0, s1, a), sb = signed_area(s0, s1, b);
    float t = fabsf(sa) / fabsf(sa - sb);
    return (1.0f - t) * a + t * b;
}
NT_DEV void insert_vec2(PolyRef arr, int arr_count, int index, vec2 element) {
    int i = arr_count;
    while (i > index) {
        arr.set(i, arr.get(i - 1));
        i -= 1;
    }
    arr.set(index, element);
}
// multicontact.py:311-381
NT_DEV int trim_in_place(vec2 s0, vec2 s1, PolyRef loop, int loop_count) {
    if (loop_count < 3) return loop_count;
    vec2 intersection_a, intersection_b;
    int change_a = -1, change_b = -1;
    bool keep = false;
    bool prev_outside = signed_area(s0, s1, loop.get(0)) <= 0.0f;
    for (int i = 0; i < loop_count; ++i) {
        int next_idx = (i + 1) % loop_count;
        bool outside = signed_area(s0, s1, loop.get(next_idx)) <= 0.0f;
        if (outside != prev_outside) {
            vec2 ip = intersection_point(s0, s1, loop.get(i), loop.get(next_idx));
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
                loop.set(loop_indexer, loop.get(i));
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
                    loop.set(loop_indexer, pt);
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
// multicontact.py:384-484; trim_poly aliases loop.get(5..9)
NT_DEV int trim_all_in_place(PolyRef trim_poly, int trim_poly_count, PolyRef loop, int loop_count) {
    if (trim_poly_count <= 1) return imin(1, loop_count);
    const float move_distance = 1e-5f;
    if (trim_poly_count == 2) {
        vec2 p0 = trim_poly.get(0), p1 = trim_poly.get(1);
        float dx = p1.x - p0.x, dy = p1.y - p0.y;
        float dir_len = sqrtf(dx * dx + dy * dy);
        if (dir_len > 1e-10f) {
            float inv = 1.0f / dir_len;
            float ox = -dy * inv * move_distance, oy = dx * inv * move_distance;
            trim_poly.set(0, vec2(p0.x - ox, p0.y - oy));
            trim_poly.set(1, vec2(p1.x - ox, p1.y - oy));
            trim_poly.set(2, vec2(p1.x + ox, p1.y + oy));
            trim_poly.set(3, vec2(p0.x + ox, p0.y + oy));
            trim_poly_count = 4;
        } else {
            return imin(1, loop_count);
        }
    }
    if (loop_count == 2) {
        vec2 p0 = loop.get(0), p1 = loop.get(1);
        float dx = p1.x - p0.x, dy = p1.y - p0.y;
        float dir_len = sqrtf(dx * dx + dy * dy);
        if (dir_len > 1e-10f) {
            float inv = 1.0f / dir_len;
            float ox = -dy * inv * move_distance, oy = dx * inv * move_distance;
            loop.set(0, vec2(p0.x - ox, p0.y - oy));
            loop.set(1, vec2(p1.x - ox, p1.y - oy));
            loop.set(2, vec2(p1.x + ox, p1.y + oy));
            loop.set(3, vec2(p0.x + ox, p0.y + oy));
            loop_count = 4;
        } else {
            return imin(1, loop_count);
        }
    }
    int current = loop_count;
    vec2 trim_poly_0 = trim_poly.get(0);
    for (int i = 0; i < trim_poly_count; ++i) {
        vec2 s0 = trim_poly.get(i);
        vec2 s1 = i == trim_poly_count - 1 ? trim_poly_0 : trim_poly.get(i + 1);
        current = trim_in_place(s0, s1, loop, current);
    }
    return current;
}
// multicontact.py:487-580
NT_DEV void approx_max_quad(PolyRef hull, int n, int out[4]) {
    int p1 = 0, p3 = 1;
    vec2 diff = hull.get(p1) - hull.get(p3);
    float max_dist_sq = diff.x * diff.x + diff.y * diff.y;
    const float tie = 1.0e-3f;
    int j = 1;
    for (int i = 0; i < n; ++i) {
        vec2 hi = hull.get(i), hi1 = hull.get((i + 1) % n);
        while (true) {
            float area_j1 = signed_area(hi, hi1, hull.get((j + 1) % n));
            float area_j = signed_area(hi, hi1, hull.get(j));
            if (area_j1 > area_j) j = (j + 1) % n;
            else break;
        }
        vec2 hj = hull.get(j);
        vec2 d1 = hull.get(i) - hj;
        float ds1 = d1.x * d1.x + d1.y * d1.y;
        if (ds1 > max_dist_sq * (1.0f + tie)) {
            max_dist_sq = ds1;
            p1 = i;
            p3 = j;
        }
        vec2 d2 = hull.get((i + 1) % n) - hj;
        float ds2 = d2.x * d2.x + d2.y * d2.y;
        if (ds2 > max_dist_sq * (1.0f + tie)) {
            max_dist_sq = ds2;
            p1 = (i + 1) % n;
            p3 = j;
        }
    }
    int p2 = 0, p4 = 0;
    float max_area_1 = 0.0f, max_area_2 = 0.0f;
    vec2 hp1 = hull.get(p1), hp3 = hull.get(p3);
    for (int i = 0; i < n; ++i) {
        float area = signed_area(hp1, hp3, hull.get(i));
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
NT_DEV int remove_zero_length_edges(PolyRef loop, int loop_count, float eps) {
    if (loop_count < 2) return 0;
    int write_idx = 0;
    for (int read_idx = 1; read_idx < loop_count; ++read_idx) {
        vec2 diff = loop.get(read_idx) - loop.get(write_idx);
        if (length_sq2(diff) > eps) {
            write_idx += 1;
            loop.set(write_idx, loop.get(read_idx));
        }
    }
    int new_count;
    if (write_idx > 0) {
        vec2 diff = loop.get(write_idx) - loop.get(0);
        new_count = length_sq2(diff) < eps ? write_idx : write_idx + 1;
    } else {
        new_count = write_idx + 1;
    }
    if (new_count < 2) new_count = 0;
    return new_count;
}
NT_DEV bool add_avoid_duplicates(PolyRef arr, int& count, vec2 v, float eps) {
    if (count > 0 && length_sq2(arr.get(0) - v) < eps) return false;
    if (count > 1 && length_sq2(arr.get(count - 1) - v) < eps) return false;
    arr.set(count, v);
    count += 1;
    return true;
}
// newton/_src/math/__init__.py:232-275
NT_DEV void orthonormal_basis(vec3 n, vec3& b1, vec3& b2) {
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

// contacts admitted for one pair (world frame, before the world->body conversion of the writer); named fields with
// select-style access keep the record in registers
struct ConvexContacts {
    vec3 c0, c1, c2, c3, c4;
    float d0, d1, d2, d3, d4;
    vec3 normal;  // shared by every contact of the pair (world frame, as generated)
#if NT_CONVEX_WITH_TRIANGLES
    vec3 n0, n1, n2, n3, n4;  // ... except in the triangle leg: a heightfield prism's penetrating contacts take the face normal
    NT_DI void set_normal(vec3 n) {  // (of the contact the next push() appends)
        const int k = count;
        n0 = vsel(k == 0, n, n0); n1 = vsel(k == 1, n, n1); n2 = vsel(k == 2, n, n2); n3 = vsel(k == 3, n, n3); n4 = vsel(k == 4, n, n4);
    }
    NT_DI vec3 normal_of(int i) const { return vsel(i == 0, n0, vsel(i == 1, n1, vsel(i == 2, n2, vsel(i == 3, n3, n4)))); }
#endif
    int count;
    // value selects on every field (see vsel in nt_math.hpp: an if-chain of stores or a ternary on lvalues turns into an
    // address select and pins the record in scratch memory)
    NT_DI void push(vec3 c, float d) {
        const int k = count;
        c0 = vsel(k == 0, c, c0); d0 = fsel(k == 0, d, d0);
        c1 = vsel(k == 1, c, c1); d1 = fsel(k == 1, d, d1);
        c2 = vsel(k == 2, c, c2); d2 = fsel(k == 2, d, d2);
        c3 = vsel(k == 3, c, c3); d3 = fsel(k == 3, d, d3);
        c4 = vsel(k == 4, c, c4); d4 = fsel(k == 4, d, d4);
        count = k + 1;
    }
    NT_DI vec3 center(int i) const { return vsel(i == 0, c0, vsel(i == 1, c1, vsel(i == 2, c2, vsel(i == 3, c3, c4)))); }
    NT_DI float distance(int i) const { return fsel(i == 0, d0, fsel(i == 1, d1, fsel(i == 2, d2, fsel(i == 3, d3, d4)))); }
};

struct PairCtx {
    Geom ga, gb;  // geometry as seen by GJK/MPR (sphere/capsule radii shrunk to 1e-4)
    float radius_eff_a, radius_eff_b, margin_a, margin_b, contact_gap;
    ConvexContacts* out;
    PolyRef poly;  // LDS scratch for the manifold clipper
#if NT_CONVEX_WITH_TRIANGLES
    bool raw;      // write_contact_to_reducer: every generated contact is kept, no gap test (contact_reduction_global.py:2059-2096)
#endif
};

// collision_core.py:173-278
NT_DEV ContactOut post_process_axial(ContactOut c, const PairCtx& P, vec3 pos_a, quat rot_a, vec3 pos_b, quat rot_b) {
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
    // is_discrete_shape (collision_core.py:39-48): shapes with flat polygon faces
    bool is_discrete_a = type_a == GEO_BOX || type_a == GEO_CONVEX_MESH || type_a == GEO_PLANE ||
                         (NT_CONVEX_WITH_TRIANGLES && (type_a == GEO_TRIANGLE || type_a == GEO_TRIANGLE_PRISM));
    bool is_discrete_b = type_b == GEO_BOX || type_b == GEO_CONVEX_MESH || type_b == GEO_PLANE;
    bool is_axial_a = type_a == GEO_CYLINDER || type_a == GEO_CONE;
    bool is_axial_b = type_b == GEO_CYLINDER || type_b == GEO_CONE;
    if ((is_discrete_a && is_axial_b) || (is_discrete_b && is_axial_a)) {
        vec3 shape_axis, shape_pos, axial_normal;
        float shape_radius, shape_half_height;
        bool is_cone;
        // (both shapes' sizes read up front and selected by value: a load in each branch is merged into one load through a
        // selected address, which keeps the shape records in scratch memory)
        const bool b_axial = is_discrete_a && is_axial_b;
        const float ra_x = P.ga.scale.x, ra_y = P.ga.scale.y, rb_x = P.gb.scale.x, rb_y = P.gb.scale.y;
        shape_radius = fsel(b_axial, rb_x, ra_x);
        shape_half_height = fsel(b_axial, rb_y, ra_y);
        if (b_axial) {
            shape_axis = quat_rotate(rot_b, vec3(0.0f, 0.0f, 1.0f));
            is_cone = type_b == GEO_CONE;
            shape_pos = pos_b;
            axial_normal = normal;
        } else {
            shape_axis = quat_rotate(rot_a, vec3(0.0f, 0.0f, 1.0f));
            is_cone = type_a == GEO_CONE;
            shape_pos = pos_a;
            axial_normal = -normal;
        }
        float axis_normal_dot = fabsf(dot(shape_axis, axial_normal));
        bool is_rolling = false;
        if (is_cone) {
            float cone_half_angle = atan2f(shape_radius, 2.0f * shape_half_height);
            const float tol = 2.0f * 3.14159265358979323846f / 180.0f;
            float lower = sinf(cone_half_angle - tol), upper = sinf(cone_half_angle + tol);
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
NT_DEV void emit(PairCtx& P, ContactOut c, vec3 pos_a, quat rot_a, vec3 pos_b, quat rot_b) {
#if NT_CONVEX_WITH_TRIANGLES
    if (P.ga.type == GEO_TRIANGLE_PRISM && c.distance < 0.0f) {
        // post_process_triangle_contact (collision_core.py:280-322): a penetrating contact of a heightfield prism moves to the
        // physical triangle face -- normal = the face normal (up), distance along it
        vec3 normal_local = cross(P.ga.scale, P.ga.aux);
        const float normal_length_sq = length_sq(normal_local);
        if (normal_length_sq >= 1.0e-20f) {
            normal_local = normal_local / sqrtf(normal_length_sq);
            if (normal_local.z < 0.0f) normal_local = -normal_local;
            const vec3 normal_world = quat_rotate(rot_a, normal_local);
            const vec3 point_b_world = c.center + 0.5f * c.distance * c.normal;
            const vec3 point_b_local = quat_rotate_inv(rot_a, point_b_world - pos_a);
            const vec3 projected_b = point_b_local - dot(point_b_local, normal_local) * normal_local;
            const vec3 point_a = closest_point_on_triangle(projected_b, vec3(0.0f), P.ga.scale, P.ga.aux);
            float distance = 0.0f;
            if (length_sq(point_a - projected_b) < 1.0e-10f) distance = dot(point_b_local - point_a, normal_local);
            c.center = quat_rotate(rot_a, point_a) + pos_a + 0.5f * distance * normal_world;
            c.normal = normal_world;
            c.distance = distance;
        }
    }
#endif
    c = post_process_axial(c, P, pos_a, rot_a, pos_b, rot_b);
    float total_separation_needed = P.radius_eff_a + P.radius_eff_b + P.margin_a + P.margin_b;
    vec3 n = normalize(c.normal);
    vec3 a_world = c.center - n * (0.5f * c.distance + P.radius_eff_a);
    vec3 b_world = c.center + n * (0.5f * c.distance + P.radius_eff_b);
    float distance = dot(b_world - a_world, n);
    float d = distance - total_separation_needed;
#if NT_CONVEX_WITH_TRIANGLES
    if (!P.raw)
#endif
    if (d > P.contact_gap) return;
    ConvexContacts& o = *P.out;
    o.normal = c.normal;
#if NT_CONVEX_WITH_TRIANGLES
    o.set_normal(c.normal);
#endif
    o.push(c.center, c.distance);
}

// multicontact.py:758-956 (+ extract_4_point_contact_manifolds :641-756)
NT_DEV int build_manifold(PairCtx& P, quat orientation_a, vec3 position_a_world, quat rel_q, vec3 rel_p, vec3 p_a, vec3 p_b, vec3 normal) {
    const float PC[5] = {1.0f, 0.30901699437494745f, -0.8090169943749473f, -0.8090169943749476f, 0.30901699437494723f};
    const float PS[5] = {0.0f, 0.9510565162951535f, 0.5877852522924732f, -0.587785252292473f, -0.9510565162951536f};
    int a_count = 0, b_count = 0;
    vec3 tangent_a, tangent_b;
    orthonormal_basis(normal, tangent_a, tangent_b);
    PlaneTracker tracker_a, tracker_b;
    vec3 center = 0.5f * (p_a + p_b);
    // 10-vertex polygon scratch (a aliases b[5..9], multicontact.py:841-843) lives in LDS: a dynamically indexed private
    // array would be placed in scratch (HBM-backed) memory
    PolyRef b_buffer = P.poly;
    PolyRef a_buffer = b_buffer + 5;
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
        bool dev_a = fabsf(dot(normal, projector_a.normal)) < COS_TILT_ANGLE;
        bool dev_b = fabsf(dot(normal, projector_b.normal)) < COS_TILT_ANGLE;
        if (!(dev_a || dev_b)) {
            normal_dot = fabsf(dot(projector_a.normal, projector_b.normal));
            int loop_count = trim_all_in_place(a_buffer, a_count, b_buffer, b_count);
            loop_count = remove_zero_length_edges(b_buffer, loop_count, MC_EPS);
            if (loop_count > 1) {
                int result[4] = {0, 1, 2, 3};
                if (loop_count > 4) {
                    approx_max_quad(b_buffer, loop_count, result);
                    loop_count = 4;
                }
                const int r0 = result[0], r1 = result[1], r2 = result[2], r3 = result[3];
                for (int i = 0; i < loop_count; ++i) {
                    int ia = i == 0 ? r0 : (i == 1 ? r1 : (i == 2 ? r2 : r3));
                    vec3 p_local = b_buffer.get(ia).x * tangent_a + b_buffer.get(ia).y * tangent_b + center;
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
            count_out = imin(loop_count, 4);
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


// compute_gjk_mpr_contacts + solve_convex_multi_contact (collision_core.py:325-452, collision_convex.py:110-232).
// Shapes arrive type-sorted (type_a <= type_b) with world transforms; contacts come back in the reference's emission order.
NT_DI void convex_pair(const Geom& geom_a, const Geom& geom_b, xform Xa, const xform& Xb, float margin_a, float margin_b,
                        float rigid_gap, vec3 aabb_lower_b, vec3 aabb_upper_b, PolyRef poly, ConvexContacts& out) {
    out.count = 0;
    PairCtx P;
    P.out = &out;
    P.poly = poly;
    P.ga = geom_a;
    P.gb = geom_b;
    P.margin_a = margin_a; P.margin_b = margin_b;
    P.contact_gap = rigid_gap;
    // pairs arrive type-sorted, so a plane (PLANE = 1) can only be shape A; a finite plane (non-zero half extents) is an
    // ordinary convex shape, the rectangle of support_map
    if (P.ga.type == GEO_PLANE && P.ga.scale.x == 0.0f && P.ga.scale.y == 0.0f) {
        // bounding-sphere half-space cull on the other shape's broad-phase AABB (narrow_phase.py:1117-1194,
        // collision_core.py:549-560,628-683), then the box proxy of convert_infinite_plane_to_cube (collision_core.py:562-625)
        vec3 bsphere_center_b = 0.5f * (aabb_lower_b + aabb_upper_b);
        float bsphere_radius_b = length(0.5f * (aabb_upper_b - aabb_lower_b));
        vec3 plane_normal = quat_rotate(Xa.q, vec3(0.0f, 0.0f, 1.0f));
        float center_dist = dot(bsphere_center_b - Xa.p, plane_normal);
        if (!(center_dist <= bsphere_radius_b)) return;
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
    const int type_a = P.ga.type, type_b = P.gb.type;
    const vec3 scale_a = P.ga.scale, scale_b = P.gb.scale;
    P.radius_eff_a = 0.0f;
    P.radius_eff_b = 0.0f;
    const float small_radius = 0.0001f;
    if (type_a == GEO_SPHERE || type_a == GEO_CAPSULE) {
        P.radius_eff_a = scale_a.x;
        P.ga.scale.x = small_radius;
    }
    if (type_b == GEO_SPHERE || type_b == GEO_CAPSULE) {
        P.radius_eff_b = scale_b.x;
        P.gb.scale.x = small_radius;
    }
    float contact_threshold = rigid_gap + P.radius_eff_a + P.radius_eff_b + margin_a + margin_b;
    bool skip_multi_contact = type_a == GEO_SPHERE || type_b == GEO_SPHERE || type_a == GEO_ELLIPSOID || type_b == GEO_ELLIPSOID;

    quat orientation_a = Xa.q, orientation_b = Xb.q;
    vec3 position_a = Xa.p, position_b = Xb.p;
    quat rel_q = quat_inverse(orientation_a) * orientation_b;
    vec3 rel_p = quat_rotate_inv(orientation_a, position_b - position_a);
    float margin_sum = margin_a + margin_b;
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
        return;
    }
    build_manifold(P, orientation_a, position_a, rel_q, rel_p, point_a, point_b, normal);
}

#if NT_CONVEX_WITH_TRIANGLES
// mesh_triangle_contacts_to_reducer_kernel's call of compute_gjk_mpr_contacts (contact_reduction_global.py:2385-2403): shape A is a
// world-space triangle (v0 at pos_a, identity rotation) or a heightfield cell's prism (edges in the heightfield frame, its rotation),
// shape B a convex shape; contacts come back in emission order, UNFILTERED (their index is the low three bits of the fingerprint).
// Post-processing as post_process_triangle_contact: the TRIANGLE_PRISM edit in emit(), then the axial projection (both are discrete).
NT_DI void triangle_pair(int type_a, vec3 edge_ab, vec3 edge_ac, vec3 pos_a, quat rot_a, const Geom& geom_b, const xform& Xb, float margin_a,
                         float margin_b, float rigid_gap, PolyRef poly, ConvexContacts& out) {
    out.count = 0;
    PairCtx P;
    P.out = &out;
    P.poly = poly;
    P.raw = true;
    P.ga.type = type_a;  // TRIANGLE (world-space edges, identity rotation) or TRIANGLE_PRISM (heightfield-local edges, its rotation)
    P.ga.scale = edge_ab;
    P.ga.aux = edge_ac;
    P.ga.center = vec3(0.0f);
    P.gb = geom_b;
    P.margin_a = margin_a; P.margin_b = margin_b;
    P.contact_gap = rigid_gap;
    const int type_b = P.gb.type;
    P.radius_eff_a = 0.0f;
    P.radius_eff_b = 0.0f;
    const float small_radius = 0.0001f;
    if (type_b == GEO_SPHERE || type_b == GEO_CAPSULE) {
        P.radius_eff_b = P.gb.scale.x;
        P.gb.scale.x = small_radius;
    }
    const float contact_threshold = rigid_gap + P.radius_eff_a + P.radius_eff_b + margin_a + margin_b;
    const bool skip_multi_contact = type_b == GEO_SPHERE || type_b == GEO_ELLIPSOID;
    const quat orientation_a = rot_a, orientation_b = Xb.q;
    const vec3 position_a = pos_a, position_b = Xb.p;
    const quat rel_q = quat_inverse(orientation_a) * orientation_b;
    const vec3 rel_p = quat_rotate_inv(orientation_a, position_b - position_a);
    const float margin_sum = margin_a + margin_b;
    const float eps = 1.0e-4f;
    const float enlarge = margin_sum <= 0.0f ? eps : (margin_sum < eps ? 2.0f * eps : 0.0f);
    vec3 point_a, point_b, normal;
    float penetration, signed_distance;
    const bool collision = solve_mpr_core(P.ga, P.gb, rel_q, rel_p, enlarge, point_a, point_b, normal, penetration);
    if (collision) {
        signed_distance = -penetration + enlarge;
        const float half_enlarge = enlarge * 0.5f;
        point_a = point_a - normal * half_enlarge;
        point_b = point_b + normal * half_enlarge;
    } else {
        solve_closest_distance_core(P.ga, P.gb, rel_q, rel_p, 0.0f, point_a, point_b, normal, signed_distance);
    }
    if (skip_multi_contact || signed_distance > contact_threshold) {
        ContactOut c;
        const vec3 point = 0.5f * (point_a + point_b);
        c.center = quat_rotate(orientation_a, point) + position_a;
        c.normal = quat_rotate(orientation_a, normal);
        c.distance = signed_distance;
        emit(P, c, position_a, orientation_a, position_b, orientation_b);
        return;
    }
    build_manifold(P, orientation_a, position_a, rel_q, rel_p, point_a, point_b, normal);
}
#endif

}  // namespace nt
