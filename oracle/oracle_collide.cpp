// TEST INFRASTRUCTURE ONLY -- CPU oracle: CollisionPipeline.collide (rigid, primitive path).
// Literal restatement (ascending-tid serial execution) of
//   compute_shape_aabbs                       newton/_src/sim/collide.py:283-472
//   check_aabb_overlap                        newton/_src/geometry/broad_phase_common.py:20-38
//   test_group_pair / test_world_and_group_pair   newton/_src/geometry/broad_phase_common.py:220-268
//   _nxn_broadphase_precomputed_pairs (EXPLICIT)  newton/_src/geometry/broad_phase_nxn.py:29-69
//   _nxn_broadphase_kernel (NXN) + precompute_world_map   broad_phase_nxn.py:72-218, broad_phase_common.py:271-388
//   narrow_phase_primitive_kernel             newton/_src/geometry/narrow_phase.py:458-1014
//   collide_plane_sphere/.../collide_sphere_box   newton/_src/geometry/collision_primitive.py:48-683,1176-1232
//   _contact_passes_gap_check_precomputed     newton/_src/geometry/contact_data.py:139-157
//   write_contact / _write_contact_at_index   newton/_src/sim/collide.py:166-254
// Pairs that the reference routes to GJK/MPR (box-box, ...) are handled by oracle_convex.cpp.
#include <algorithm>
#include <vector>

#include "oracle_common.h"

using namespace orc;

namespace orc {
// implemented in oracle_convex.cpp (MPR/GJK/manifold); returns number of contacts written
int convex_pair_contacts(const o_model* m, int shape_a, int shape_b, const float* geom_data /*[S][4]*/,
                         const float* geom_xform /*[S][7]*/, const float* aabb_lower, const float* aabb_upper,
                         const float* body_q, o_contacts* ct);
vec3 support_map_generic(int type, vec3 scale, vec3 direction);  // support_function.py:131-350
}  // namespace orc

struct vec4f {
    float v[4];
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
};
struct mat43f {
    vec3 r[4];
};

static const float MINVAL = 1e-15f;

// ---------------------------------------------------------------- collision_primitive.py:48-98
static vec3 closest_segment_point(vec3 a, vec3 b, vec3 pt) {
    vec3 ab = b - a;
    float t = dot(pt - a, ab) / (dot(ab, ab) + 1e-6f);
    return a + clampf(t, 0.0f, 1.0f) * ab;
}

// collision_primitive.py:100-108
static void collide_plane_sphere(vec3 plane_normal, vec3 plane_pos, vec3 sphere_pos, float sphere_radius, float& dist,
                                 vec3& pos) {
    dist = dot(sphere_pos - plane_pos, plane_normal) - sphere_radius;
    pos = sphere_pos - plane_normal * (sphere_radius + 0.5f * dist);
}

// collision_primitive.py:110-140
static void collide_sphere_sphere(vec3 pos1, float radius1, vec3 pos2, float radius2, float& dist, vec3& pos, vec3& n) {
    vec3 dir = pos2 - pos1;
    dist = length(dir);
    if (dist == 0.0f)
        n = vec3(1.0f, 0.0f, 0.0f);
    else
        n = dir / dist;
    dist = dist - (radius1 + radius2);
    pos = pos1 + n * (radius1 + 0.5f * dist);
}

// collision_primitive.py:143-177
static void collide_sphere_capsule(vec3 sphere_pos, float sphere_radius, vec3 capsule_pos, vec3 capsule_axis,
                                   float capsule_radius, float capsule_half_length, float& dist, vec3& pos, vec3& n) {
    vec3 segment = capsule_axis * capsule_half_length;
    vec3 pt = closest_segment_point(capsule_pos - segment, capsule_pos + segment, sphere_pos);
    collide_sphere_sphere(sphere_pos, sphere_radius, pt, capsule_radius, dist, pos, n);
}

// collision_primitive.py:180-275
static void collide_capsule_capsule(vec3 cap1_pos, vec3 cap1_axis, float cap1_radius, float cap1_half_length, vec3 cap2_pos,
                                    vec3 cap2_axis, float cap2_radius, float cap2_half_length, float contact_dist[2],
                                    vec3 contact_pos[2], vec3& contact_normal) {
    contact_dist[0] = MAXVAL;
    contact_dist[1] = MAXVAL;
    contact_normal = vec3();

    vec3 axis1 = cap1_axis * cap1_half_length;
    vec3 axis2 = cap2_axis * cap2_half_length;
    vec3 dif = cap1_pos - cap2_pos;

    float ma = dot(axis1, axis1);
    float mb = -dot(axis1, axis2);
    float mc = dot(axis2, axis2);
    float u = -dot(axis1, dif);
    float v = dot(axis2, dif);
    float det = ma * mc - mb * mb;

    if (std::fabs(det) >= MINVAL) {
        float inv_det = 1.0f / det;
        float x1 = (mc * u - mb * v) * inv_det;
        float x2 = (ma * v - mb * u) * inv_det;

        if (x1 > 1.0f) {
            x1 = 1.0f;
            x2 = (v - mb) / mc;
        } else if (x1 < -1.0f) {
            x1 = -1.0f;
            x2 = (v + mb) / mc;
        }
        if (x2 > 1.0f) {
            x2 = 1.0f;
            x1 = clampf((u - mb) / ma, -1.0f, 1.0f);
        } else if (x2 < -1.0f) {
            x2 = -1.0f;
            x1 = clampf((u + mb) / ma, -1.0f, 1.0f);
        }
        vec3 vec1 = cap1_pos + axis1 * x1;
        vec3 vec2 = cap2_pos + axis2 * x2;
        collide_sphere_sphere(vec1, cap1_radius, vec2, cap2_radius, contact_dist[0], contact_pos[0], contact_normal);
    } else {
        vec3 vec1 = cap1_pos + axis1;
        float x2 = clampf((v - mb) / mc, -1.0f, 1.0f);
        vec3 vec2 = cap2_pos + axis2 * x2;
        collide_sphere_sphere(vec1, cap1_radius, vec2, cap2_radius, contact_dist[0], contact_pos[0], contact_normal);

        vec1 = cap1_pos - axis1;
        x2 = clampf((v + mb) / mc, -1.0f, 1.0f);
        vec2 = cap2_pos + axis2 * x2;
        vec3 n_unused;
        collide_sphere_sphere(vec1, cap1_radius, vec2, cap2_radius, contact_dist[1], contact_pos[1], n_unused);
    }
}

// collision_primitive.py:352-381
static void collide_plane_ellipsoid(vec3 plane_normal, vec3 plane_pos, vec3 ellipsoid_pos, const mat33& ellipsoid_rot,
                                    vec3 ellipsoid_size, float& dist, vec3& pos, vec3& normal) {
    vec3 sphere_support = -normalize(cw_mul(transpose(ellipsoid_rot) * plane_normal, ellipsoid_size));
    pos = ellipsoid_pos + ellipsoid_rot * cw_mul(sphere_support, ellipsoid_size);
    dist = dot(plane_normal, pos - plane_pos);
    pos = pos - plane_normal * dist * 0.5f;
    normal = plane_normal;
}

// collision_primitive.py:384-458
static void collide_plane_box(vec3 plane_normal, vec3 plane_pos, vec3 box_pos, const mat33& box_rot, vec3 box_size,
                              float margin, float dist[4], vec3 pos[4], vec3& normal) {
    vec3 corner;
    float center_dist = dot(box_pos - plane_pos, plane_normal);
    for (int i = 0; i < 4; ++i) {
        dist[i] = MAXVAL;
        pos[i] = vec3();
    }
    int ncontact = 0;
    int worst_idx = 0;
    for (int i = 0; i < 8; ++i) {
        corner.x = (i & 1) != 0 ? box_size.x : -box_size.x;
        corner.y = (i & 2) != 0 ? box_size.y : -box_size.y;
        corner.z = (i & 4) != 0 ? box_size.z : -box_size.z;
        corner = box_rot * corner;

        float ldist = dot(plane_normal, corner);
        float cdist = center_dist + ldist;
        if (cdist > margin) continue;

        vec3 cpos = corner + box_pos - 0.5f * plane_normal * cdist;

        if (ncontact < 4) {
            dist[ncontact] = cdist;
            pos[ncontact] = cpos;
            if (ncontact == 0 || cdist > dist[worst_idx]) worst_idx = ncontact;
            ncontact += 1;
        } else {
            if (cdist < dist[worst_idx]) {
                dist[worst_idx] = cdist;
                pos[worst_idx] = cpos;
                worst_idx = 0;
                if (dist[1] > dist[worst_idx]) worst_idx = 1;
                if (dist[2] > dist[worst_idx]) worst_idx = 2;
                if (dist[3] > dist[worst_idx]) worst_idx = 3;
            }
        }
    }
    normal = plane_normal;
}

// newton/_src/math/__init__.py:282-293
static float safe_div(float x, float y) { return x / (y != 0.0f ? y : 1e-15f); }

// collision_primitive.py:461-531
static void collide_sphere_cylinder(vec3 sphere_pos, float sphere_radius, vec3 cylinder_pos, vec3 cylinder_axis,
                                    float cylinder_radius, float cylinder_half_height, float& dist, vec3& pos, vec3& n) {
    vec3 vec = sphere_pos - cylinder_pos;
    float x = dot(vec, cylinder_axis);

    vec3 a_proj = cylinder_axis * x;
    vec3 p_proj = vec - a_proj;
    float p_proj_sqr = dot(p_proj, p_proj);

    bool collide_side = std::fabs(x) < cylinder_half_height;
    bool collide_cap = p_proj_sqr < (cylinder_radius * cylinder_radius);

    if (collide_side && collide_cap) {
        float dist_cap = cylinder_half_height - std::fabs(x);
        float dist_radius = cylinder_radius - std::sqrt(p_proj_sqr);
        if (dist_cap < dist_radius)
            collide_side = false;
        else
            collide_cap = false;
    }

    if (collide_side) {
        vec3 pos_target = cylinder_pos + a_proj;
        collide_sphere_sphere(sphere_pos, sphere_radius, pos_target, cylinder_radius, dist, pos, n);
    } else if (collide_cap) {
        vec3 pos_cap, plane_normal;
        if (x > 0.0f) {
            pos_cap = cylinder_pos + cylinder_axis * cylinder_half_height;
            plane_normal = cylinder_axis;
        } else {
            pos_cap = cylinder_pos - cylinder_axis * cylinder_half_height;
            plane_normal = -cylinder_axis;
        }
        collide_plane_sphere(plane_normal, pos_cap, sphere_pos, sphere_radius, dist, pos);
        n = -plane_normal;
    } else {
        float inv_len = safe_div(1.0f, std::sqrt(p_proj_sqr));
        p_proj = p_proj * (cylinder_radius * inv_len);
        vec3 cap_offset = cylinder_axis * (signf(x) * cylinder_half_height);
        vec3 pos_corner = cylinder_pos + cap_offset + p_proj;
        collide_sphere_sphere(sphere_pos, sphere_radius, pos_corner, 0.0f, dist, pos, n);
    }
}

// collision_primitive.py:42-45 : cos(22.5 deg)
static const float CYLINDER_FLAT_MODE_COS = 0.9238795325112867f;

// collision_primitive.py:534-683
static void collide_plane_cylinder(vec3 plane_normal, vec3 plane_pos, vec3 cylinder_pos, vec3 cylinder_axis,
                                   float cylinder_radius, float cylinder_half_height, float contact_dist[4],
                                   vec3 contact_pos[4], vec3& normal) {
    for (int i = 0; i < 4; ++i) {
        contact_dist[i] = MAXVAL;
        contact_pos[i] = vec3();
    }
    vec3 n = plane_normal;
    vec3 axis = cylinder_axis;

    float dot_na = dot(n, axis);
    if (dot_na > 0.0f) {
        axis = -axis;
        dot_na = -dot_na;
    }
    vec3 cap_center = cylinder_pos + axis * cylinder_half_height;

    vec3 perp_align = -n + axis * dot_na;
    float perp_align_len_sq = dot(perp_align, perp_align);
    bool has_align = perp_align_len_sq > 1e-10f;
    if (has_align) perp_align = perp_align * (1.0f / std::sqrt(perp_align_len_sq));

    float abs_dot = -dot_na;
    bool in_flat_surface_mode = abs_dot >= CYLINDER_FLAT_MODE_COS;

    vec3 perp_fixed;
    if (in_flat_surface_mode || !has_align) {
        vec3 ref(1.0f, 0.0f, 0.0f);
        if (std::fabs(dot(axis, ref)) > 0.9f) ref = vec3(0.0f, 1.0f, 0.0f);
        perp_fixed = ref - axis * dot(axis, ref);
        perp_fixed = normalize(perp_fixed);
    }

    vec3 deepest_perp = has_align ? perp_align : perp_fixed;
    vec3 deepest_pt = cap_center + deepest_perp * cylinder_radius;
    float deepest_d = dot(deepest_pt - plane_pos, n);
    vec3 deepest_pos = deepest_pt - n * (deepest_d * 0.5f);

    contact_dist[0] = deepest_d;
    contact_pos[0] = deepest_pos;
    int ncontact = 1;
    float merge_threshold = 0.01f * wmax(cylinder_radius, cylinder_half_height);
    float merge_threshold_sq = merge_threshold * merge_threshold;

    if (in_flat_surface_mode) {
        vec3 u_fixed = perp_fixed * cylinder_radius;
        vec3 v_fixed = cross(axis, perp_fixed) * cylinder_radius;
        const float c120 = -0.5f;
        const float s120 = 0.8660254f;

        vec3 pt0 = cap_center + u_fixed;
        float d0 = dot(pt0 - plane_pos, n);
        vec3 pos0 = pt0 - n * (d0 * 0.5f);
        if (ncontact < 4 && length_sq(pos0 - deepest_pos) > merge_threshold_sq) {
            contact_dist[ncontact] = d0;
            contact_pos[ncontact] = pos0;
            ncontact += 1;
        }
        vec3 pt1 = cap_center + c120 * u_fixed + s120 * v_fixed;
        float d1 = dot(pt1 - plane_pos, n);
        vec3 pos1 = pt1 - n * (d1 * 0.5f);
        if (ncontact < 4 && length_sq(pos1 - deepest_pos) > merge_threshold_sq) {
            contact_dist[ncontact] = d1;
            contact_pos[ncontact] = pos1;
            ncontact += 1;
        }
        vec3 pt2 = cap_center + c120 * u_fixed - s120 * v_fixed;
        float d2 = dot(pt2 - plane_pos, n);
        vec3 pos2 = pt2 - n * (d2 * 0.5f);
        if (ncontact < 4 && length_sq(pos2 - deepest_pos) > merge_threshold_sq) {
            contact_dist[ncontact] = d2;
            contact_pos[ncontact] = pos2;
            ncontact += 1;
        }
    } else {
        vec3 perp_roll = has_align ? perp_align : perp_fixed;
        vec3 u = perp_roll * cylinder_radius;
        vec3 v = cross(axis, perp_roll) * cylinder_radius;

        vec3 pt = cylinder_pos - axis * cylinder_half_height + u;
        float d = dot(pt - plane_pos, n);
        vec3 pos = pt - n * (d * 0.5f);
        if (ncontact < 4 && length_sq(pos - deepest_pos) > merge_threshold_sq) {
            contact_dist[ncontact] = d;
            contact_pos[ncontact] = pos;
            ncontact += 1;
        }

        vec3 pt_pos_v = cap_center + v;
        float d_pos_v = dot(pt_pos_v - plane_pos, n);
        vec3 pt_neg_v = cap_center - v;
        float d_neg_v = dot(pt_neg_v - plane_pos, n);
        bool use_pos_v = d_pos_v <= d_neg_v;
        pt = use_pos_v ? pt_pos_v : pt_neg_v;
        d = use_pos_v ? d_pos_v : d_neg_v;
        pos = pt - n * (d * 0.5f);
        if (ncontact < 4 && length_sq(pos - deepest_pos) > merge_threshold_sq) {
            contact_dist[ncontact] = d;
            contact_pos[ncontact] = pos;
            ncontact += 1;
        }
    }
    normal = n;
}

// collision_primitive.py:1176-1232
static void collide_sphere_box(vec3 sphere_pos, float sphere_radius, vec3 box_pos, const mat33& box_rot, vec3 box_size,
                               float& contact_distance, vec3& contact_position, vec3& contact_normal) {
    vec3 center = transpose(box_rot) * (sphere_pos - box_pos);
    vec3 clamped = vmax(-box_size, vmin(box_size, center));
    vec3 diff = clamped - center;
    float dist = length(diff);
    vec3 clamped_dir = dist == 0.0f ? diff : diff / dist;  // normalize_with_norm

    vec3 pos;
    if (dist <= 1e-6f) {
        float closest = 2.0f * (box_size[0] + box_size[1] + box_size[2]);
        int k = 0;
        for (int i = 0; i < 6; ++i) {
            float face_dist = std::fabs(((i % 2) ? 1.0f : -1.0f) * box_size[i / 2] - center[i / 2]);
            if (closest > face_dist) {
                closest = face_dist;
                k = i;
            }
        }
        vec3 nearest(0.0f);
        nearest[k / 2] = (k % 2) ? -1.0f : 1.0f;
        pos = center + nearest * (sphere_radius - closest) / 2.0f;
        contact_normal = box_rot * nearest;
        contact_distance = -closest - sphere_radius;
    } else {
        vec3 deepest = center + clamped_dir * sphere_radius;
        pos = 0.5f * (clamped + deepest);
        contact_normal = box_rot * clamped_dir;
        contact_distance = dist - sphere_radius;
    }
    contact_position = box_pos + box_rot * pos;
}

// ---------------------------------------------------------------- collide.py:283-472
static void compute_shape_aabbs(const o_model* m, const float* body_q, float* aabb_lower, float* aabb_upper,
                                float* geom_data /*[S][4]*/, float* geom_xform /*[S][7]*/) {
    for (int shape_id = 0; shape_id < m->shape_count; ++shape_id) {
        int rigid_id = m->shape_body[shape_id];
        int geo_type = m->shape_type[shape_id];
        transform X_ws;
        if (rigid_id == -1)
            X_ws = ldx(m->shape_transform, shape_id);
        else
            X_ws = ldx(body_q, rigid_id) * ldx(m->shape_transform, shape_id);
        vec3 pos = X_ws.p;
        quat orientation = X_ws.q;

        float margin = m->shape_margin[shape_id];
        float effective_gap = margin + m->shape_gap[shape_id];
        vec3 margin_vec(effective_gap, effective_gap, effective_gap);

        vec3 scale = ld3(m->shape_scale, shape_id);
        bool is_infinite_plane = (geo_type == GEO_PLANE) && (scale[0] == 0.0f && scale[1] == 0.0f);
        vec3 geom_scale = scale;
        vec3 lo, hi;

        if (is_infinite_plane) {
            vec3 normal = quat_rotate(orientation, vec3(0.0f, 0.0f, 1.0f));
            const float HALF_SPACE_EXTENT = 1.0e6f;
            vec3 half_extents(HALF_SPACE_EXTENT, HALF_SPACE_EXTENT, HALF_SPACE_EXTENT);
            lo = pos - half_extents - margin_vec;
            hi = pos + half_extents + margin_vec;
            for (int i = 0; i < 3; ++i) {
                float n_i = normal[i];
                if (std::fabs(n_i) > 0.5f) {
                    float lateral = std::fabs(normal[(i + 1) % 3]) + std::fabs(normal[(i + 2) % 3]);
                    float rise = lateral * HALF_SPACE_EXTENT / std::fabs(n_i);
                    if (n_i > 0.0f)
                        hi[i] = wmin(hi[i], pos[i] + rise + effective_gap);
                    else
                        lo[i] = wmax(lo[i], pos[i] - rise - effective_gap);
                }
            }
        } else if (geo_type == GEO_SPHERE) {
            float radius = scale[0];
            vec3 half_extents(radius, radius, radius);
            lo = pos - half_extents - margin_vec;
            hi = pos + half_extents + margin_vec;
        } else if (geo_type == GEO_BOX) {
            vec3 r0 = quat_rotate(orientation, vec3(1.0f, 0.0f, 0.0f));
            vec3 r1 = quat_rotate(orientation, vec3(0.0f, 1.0f, 0.0f));
            vec3 r2 = quat_rotate(orientation, vec3(0.0f, 0.0f, 1.0f));
            vec3 half_extents(std::fabs(r0[0]) * scale[0] + std::fabs(r1[0]) * scale[1] + std::fabs(r2[0]) * scale[2],
                              std::fabs(r0[1]) * scale[0] + std::fabs(r1[1]) * scale[1] + std::fabs(r2[1]) * scale[2],
                              std::fabs(r0[2]) * scale[0] + std::fabs(r1[2]) * scale[1] + std::fabs(r2[2]) * scale[2]);
            lo = pos - half_extents - margin_vec;
            hi = pos + half_extents + margin_vec;
        } else if (geo_type == GEO_CAPSULE) {
            float radius = scale[0];
            float half_height = scale[1];
            vec3 axis = quat_rotate(orientation, vec3(0.0f, 0.0f, 1.0f));
            vec3 half_extents = vec3(radius, radius, radius) + vabs(axis) * half_height;
            lo = pos - half_extents - margin_vec;
            hi = pos + half_extents + margin_vec;
        } else if (geo_type == GEO_CYLINDER) {
            float radius = scale[0];
            float half_height = scale[1];
            float barrel_radius = scale[2];
            if (barrel_radius >= half_height && barrel_radius > 0.0f)
                radius += (half_height * half_height) /
                          (barrel_radius + std::sqrt(barrel_radius * barrel_radius - half_height * half_height));
            vec3 r0 = quat_rotate(orientation, vec3(1.0f, 0.0f, 0.0f));
            vec3 r1 = quat_rotate(orientation, vec3(0.0f, 1.0f, 0.0f));
            vec3 r2 = quat_rotate(orientation, vec3(0.0f, 0.0f, 1.0f));
            vec3 half_extents(radius * std::sqrt(r0[0] * r0[0] + r1[0] * r1[0]) + half_height * std::fabs(r2[0]),
                              radius * std::sqrt(r0[1] * r0[1] + r1[1] * r1[1]) + half_height * std::fabs(r2[1]),
                              radius * std::sqrt(r0[2] * r0[2] + r1[2] * r1[2]) + half_height * std::fabs(r2[2]));
            lo = pos - half_extents - margin_vec;
            hi = pos + half_extents + margin_vec;
        } else if (geo_type == GEO_CONVEX_MESH) {
            // pre-computed local AABB (scale baked in) rotated to the world frame (collide.py:421-445)
            vec3 local_lo = ld3(m->shape_collision_aabb_lower, shape_id), local_hi = ld3(m->shape_collision_aabb_upper, shape_id);
            vec3 center = (local_lo + local_hi) * 0.5f;
            vec3 half = (local_hi - local_lo) * 0.5f;
            vec3 world_center = quat_rotate(orientation, center) + pos;
            vec3 r0 = quat_rotate(orientation, vec3(1.0f, 0.0f, 0.0f));
            vec3 r1 = quat_rotate(orientation, vec3(0.0f, 1.0f, 0.0f));
            vec3 r2 = quat_rotate(orientation, vec3(0.0f, 0.0f, 1.0f));
            vec3 world_half(std::fabs(r0[0]) * half[0] + std::fabs(r1[0]) * half[1] + std::fabs(r2[0]) * half[2],
                            std::fabs(r0[1]) * half[0] + std::fabs(r1[1]) * half[1] + std::fabs(r2[1]) * half[2],
                            std::fabs(r0[2]) * half[0] + std::fabs(r1[2]) * half[1] + std::fabs(r2[2]) * half[2]);
            lo = world_center - world_half - margin_vec;
            hi = world_center + world_half + margin_vec;
        } else if (geo_type == GEO_ELLIPSOID || geo_type == GEO_CONE || geo_type == GEO_PLANE) {
            // finite plane: a rectangle with half extents scale / 2 (collide.py:452-453)
            if (geo_type == GEO_PLANE) geom_scale = vec3(scale[0] * 0.5f, scale[1] * 0.5f, 0.0f);
            // compute_tight_aabb_from_support (collision_core.py:454-547): six support evaluations in local space
            mat33 rot_mat_t = transpose(quat_to_matrix(orientation));
            vec3 local_x(rot_mat_t(0, 0), rot_mat_t(1, 0), rot_mat_t(2, 0));
            vec3 local_y(rot_mat_t(0, 1), rot_mat_t(1, 1), rot_mat_t(2, 1));
            vec3 local_z(rot_mat_t(0, 2), rot_mat_t(1, 2), rot_mat_t(2, 2));
            float max_x = dot(local_x, support_map_generic(geo_type, geom_scale, local_x));
            float max_y = dot(local_y, support_map_generic(geo_type, geom_scale, local_y));
            float max_z = dot(local_z, support_map_generic(geo_type, geom_scale, local_z));
            float min_x = dot(local_x, support_map_generic(geo_type, geom_scale, -local_x));
            float min_y = dot(local_y, support_map_generic(geo_type, geom_scale, -local_y));
            float min_z = dot(local_z, support_map_generic(geo_type, geom_scale, -local_z));
            lo = vec3(min_x, min_y, min_z) + pos - margin_vec;
            hi = vec3(max_x, max_y, max_z) + pos + margin_vec;
        } else {
            // triangle meshes: not restated (conservative bounding sphere; rejected by the product host)
            float r = m->shape_collision_radius[shape_id];
            vec3 half_extents(r, r, r);
            lo = pos - half_extents - margin_vec;
            hi = pos + half_extents + margin_vec;
        }
        st3(aabb_lower, shape_id, lo);
        st3(aabb_upper, shape_id, hi);
        geom_data[4 * shape_id + 0] = geom_scale[0];
        geom_data[4 * shape_id + 1] = geom_scale[1];
        geom_data[4 * shape_id + 2] = geom_scale[2];
        geom_data[4 * shape_id + 3] = margin;
        stx(geom_xform, shape_id, X_ws);
    }
}

// broad_phase_common.py:20-38 with cutoff 0 (AABBs are pre-expanded: collide.py:1911)
static bool check_aabb_overlap(vec3 l1, vec3 u1, vec3 l2, vec3 u2) {
    const float c = 0.0f + 0.0f;
    return l1[0] <= u2[0] + c && u1[0] >= l2[0] - c && l1[1] <= u2[1] + c && u1[1] >= l2[1] - c && l1[2] <= u2[2] + c &&
           u1[2] >= l2[2] - c;
}

static bool test_group_pair(int a, int b) {
    if (a == 0 || b == 0) return false;
    if (a > 0) return a == b || b < 0;
    return a != b;
}
static bool test_world_and_group_pair(int wa, int wb, int ga, int gb) {
    if (wa != -1 && wb != -1 && wa != wb) return false;
    return test_group_pair(ga, gb);
}

// Candidate pairs in the order a serial Warp-CPU launch would append them.
static void broadphase_explicit(const o_model* m, const float* lo, const float* hi, std::vector<int>& pairs) {
    for (int e = 0; e < m->pair_count; ++e) {
        int s1 = m->shape_contact_pairs[2 * e], s2 = m->shape_contact_pairs[2 * e + 1];
        if (check_aabb_overlap(ld3(lo, s1), ld3(hi, s1), ld3(lo, s2), ld3(hi, s2))) {
            pairs.push_back(s1);
            pairs.push_back(s2);
        }
    }
}

// broad_phase_nxn.py:132-218 driven by precompute_world_map (broad_phase_common.py:271-388).
// is_pair_excluded: binary search in the sorted exclusion list (broad_phase_common.py:132-162)
static bool is_pair_excluded(const o_model* m, int s1, int s2) {
    int low = 0, high = m->filter_pair_count - 1;
    while (low <= high) {
        int mid = (low + high) >> 1;
        int a = m->shape_collision_filter_pairs[2 * mid], b = m->shape_collision_filter_pairs[2 * mid + 1];
        if (a == s1 && b == s2) return true;
        if (s1 < a || (s1 == a && s2 < b)) high = mid - 1;
        else low = mid + 1;
    }
    return false;
}
static void broadphase_nxn(const o_model* m, const float* lo, const float* hi, std::vector<int>& pairs) {
    const int S = m->shape_count;
    std::vector<int> shared, worlds;
    for (int i = 0; i < S; ++i) {
        if (!(m->shape_flags[i] & 2)) continue;  // COLLIDE_SHAPES
        if (m->shape_world[i] == -1) shared.push_back(i);
    }
    int max_world = -1;
    for (int i = 0; i < S; ++i) max_world = std::max(max_world, m->shape_world[i]);
    std::vector<std::vector<int>> segs;
    for (int w = 0; w <= max_world; ++w) {
        std::vector<int> seg;
        for (int i = 0; i < S; ++i)
            if ((m->shape_flags[i] & 2) && m->shape_world[i] == w) seg.push_back(i);
        if (seg.empty()) continue;
        seg.insert(seg.end(), shared.begin(), shared.end());
        segs.push_back(seg);
    }
    int num_regular = (int)segs.size();
    segs.push_back(shared);
    for (int wid = 0; wid < (int)segs.size(); ++wid) {
        const std::vector<int>& seg = segs[wid];
        int n = (int)seg.size();
        // thread order: local_id ascending == (r, c) lexicographic over the upper triangle
        for (int r = 0; r < n; ++r)
            for (int c = r + 1; c < n; ++c) {
                int s1 = std::min(seg[r], seg[c]), s2 = std::max(seg[r], seg[c]);
                int w1 = m->shape_world[s1], w2 = m->shape_world[s2];
                bool dedicated = wid >= num_regular;
                if (w1 == -1 && w2 == -1 && !dedicated) continue;
                if (!test_world_and_group_pair(w1, w2, m->shape_collision_group[s1], m->shape_collision_group[s2])) continue;
                if (check_aabb_overlap(ld3(lo, s1), ld3(hi, s1), ld3(lo, s2), ld3(hi, s2))) {
                    if (m->filter_pair_count > 0 && is_pair_excluded(m, s1, s2)) continue;
                    pairs.push_back(s1);
                    pairs.push_back(s2);
                }
            }
    }
}

// ---------------------------------------------------------------- contact writer (collide.py:166-254)
struct ContactData {
    vec3 contact_point_center, contact_normal_a_to_b;
    float contact_distance, radius_eff_a, radius_eff_b, margin_a, margin_b;
    int shape_a, shape_b;
    float gap_sum;
};

static bool contact_passes_gap_check_precomputed(const ContactData& cd, vec3 n, float total_separation_needed) {
    vec3 a = cd.contact_point_center - n * (0.5f * cd.contact_distance + cd.radius_eff_a);
    vec3 b = cd.contact_point_center + n * (0.5f * cd.contact_distance + cd.radius_eff_b);
    vec3 diff = b - a;
    float distance = dot(diff, n);
    float d = distance - total_separation_needed;
    return d <= cd.gap_sum;
}

namespace orc {
// _write_contact_at_index + write_contact(output_index >= 0)
void write_contact_at(const o_model* m, const float* body_q, o_contacts* ct, int index, int shape_a, int shape_b,
                      vec3 center, vec3 normal_in, float distance, float radius_eff_a, float radius_eff_b, float margin_a,
                      float margin_b) {
    vec3 n = normalize(normal_in);
    vec3 a_world = center - n * (0.5f * distance + radius_eff_a);
    vec3 b_world = center + n * (0.5f * distance + radius_eff_b);
    if (index >= ct->rigid_contact_max) return;
    ct->shape0[index] = shape_a;
    ct->shape1[index] = shape_b;
    int body0 = m->shape_body[shape_a];
    int body1 = m->shape_body[shape_b];
    transform X_bw_a = body0 == -1 ? transform_identity() : transform_inverse(ldx(body_q, body0));
    transform X_bw_b = body1 == -1 ? transform_identity() : transform_inverse(ldx(body_q, body1));
    st3(ct->point0, index, transform_point(X_bw_a, a_world));
    st3(ct->point1, index, transform_point(X_bw_b, b_world));
    float offset_mag_a = radius_eff_a + margin_a;
    float offset_mag_b = radius_eff_b + margin_b;
    st3(ct->offset0, index, transform_vector(X_bw_a, offset_mag_a * n));
    st3(ct->offset1, index, transform_vector(X_bw_b, -offset_mag_b * n));
    st3(ct->normal, index, n);
    ct->margin0[index] = offset_mag_a;
    ct->margin1[index] = offset_mag_b;
    ct->tids[index] = 0;
}
}  // namespace orc

// analytic contacts for one (type-sorted) pair; returns false if the pair has no analytic path
static bool primitive_pair(int type_a, int type_b, const transform& X_a, const transform& X_b, vec3 scale_a, vec3 scale_b,
                           float margin_for_box, float dist[4], vec3 pos[4], vec3& contact_normal, bool& handled_empty) {
    for (int i = 0; i < 4; ++i) {
        dist[i] = MAXVAL;
        pos[i] = vec3();
    }
    contact_normal = vec3();
    vec3 pos_a = X_a.p, pos_b = X_b.p;
    quat quat_a = X_a.q, quat_b = X_b.q;

    bool is_plane_a = type_a == GEO_PLANE;
    bool is_sphere_a = type_a == GEO_SPHERE, is_sphere_b = type_b == GEO_SPHERE;
    bool is_capsule_a = type_a == GEO_CAPSULE, is_capsule_b = type_b == GEO_CAPSULE;
    bool is_ellipsoid_b = type_b == GEO_ELLIPSOID;
    bool is_cylinder_b = type_b == GEO_CYLINDER;
    bool is_box_b = type_b == GEO_BOX;

    bool use_plane_cylinder = is_plane_a && is_cylinder_b;
    if (use_plane_cylinder && scale_b[2] > 0.0f) {
        vec3 plane_normal = quat_rotate(quat_a, vec3(0.0f, 0.0f, 1.0f));
        vec3 cylinder_axis = quat_rotate(quat_b, vec3(0.0f, 0.0f, 1.0f));
        use_plane_cylinder = std::fabs(dot(plane_normal, cylinder_axis)) * scale_b[2] >= scale_b[1];
    }

    if (is_plane_a && is_sphere_b) {
        vec3 plane_normal = quat_rotate(quat_a, vec3(0.0f, 0.0f, 1.0f));
        collide_plane_sphere(plane_normal, pos_a, pos_b, scale_b[0], dist[0], pos[0]);
        contact_normal = plane_normal;
    } else if (is_plane_a && is_ellipsoid_b) {
        vec3 plane_normal = quat_rotate(quat_a, vec3(0.0f, 0.0f, 1.0f));
        mat33 rot = quat_to_matrix(quat_b);
        collide_plane_ellipsoid(plane_normal, pos_a, pos_b, rot, scale_b, dist[0], pos[0], contact_normal);
    } else if (is_plane_a && is_box_b) {
        vec3 plane_normal = quat_rotate(quat_a, vec3(0.0f, 0.0f, 1.0f));
        mat33 rot = quat_to_matrix(quat_b);
        collide_plane_box(plane_normal, pos_a, pos_b, rot, scale_b, margin_for_box, dist, pos, contact_normal);
    } else if (is_sphere_a && is_sphere_b) {
        collide_sphere_sphere(pos_a, scale_a[0], pos_b, scale_b[0], dist[0], pos[0], contact_normal);
    } else if (is_plane_a && is_capsule_b) {
        vec3 plane_normal = quat_rotate(quat_a, vec3(0.0f, 0.0f, 1.0f));
        vec3 capsule_axis = quat_rotate(quat_b, vec3(0.0f, 0.0f, 1.0f));
        vec3 segment = capsule_axis * scale_b[1];
        collide_plane_sphere(plane_normal, pos_a, pos_b + segment, scale_b[0], dist[0], pos[0]);
        collide_plane_sphere(plane_normal, pos_a, pos_b - segment, scale_b[0], dist[1], pos[1]);
        contact_normal = plane_normal;
    } else if (use_plane_cylinder) {
        vec3 plane_normal = quat_rotate(quat_a, vec3(0.0f, 0.0f, 1.0f));
        vec3 cylinder_axis = quat_rotate(quat_b, vec3(0.0f, 0.0f, 1.0f));
        collide_plane_cylinder(plane_normal, pos_a, pos_b, cylinder_axis, scale_b[0], scale_b[1], dist, pos, contact_normal);
    } else if (is_sphere_a && is_capsule_b) {
        vec3 capsule_axis = quat_rotate(quat_b, vec3(0.0f, 0.0f, 1.0f));
        collide_sphere_capsule(pos_a, scale_a[0], pos_b, capsule_axis, scale_b[0], scale_b[1], dist[0], pos[0], contact_normal);
    } else if (is_capsule_a && is_capsule_b) {
        vec3 axis_a = quat_rotate(quat_a, vec3(0.0f, 0.0f, 1.0f));
        vec3 axis_b = quat_rotate(quat_b, vec3(0.0f, 0.0f, 1.0f));
        float d2[2];
        vec3 p2[2];
        collide_capsule_capsule(pos_a, axis_a, scale_a[0], scale_a[1], pos_b, axis_b, scale_b[0], scale_b[1], d2, p2,
                                contact_normal);
        dist[0] = d2[0];
        pos[0] = p2[0];
        dist[1] = d2[1];
        pos[1] = p2[1];
    } else if (is_sphere_a && is_cylinder_b && scale_b[2] == 0.0f) {
        vec3 cylinder_axis = quat_rotate(quat_b, vec3(0.0f, 0.0f, 1.0f));
        collide_sphere_cylinder(pos_a, scale_a[0], pos_b, cylinder_axis, scale_b[0], scale_b[1], dist[0], pos[0],
                                contact_normal);
    } else if (is_sphere_a && is_box_b) {
        mat33 rot = quat_to_matrix(quat_b);
        collide_sphere_box(pos_a, scale_a[0], pos_b, rot, scale_b, dist[0], pos[0], contact_normal);
    }

    handled_empty = (is_plane_a && (is_sphere_b || is_capsule_b || is_ellipsoid_b || use_plane_cylinder || is_box_b)) ||
                    (is_sphere_a && (is_sphere_b || is_capsule_b || (is_cylinder_b && scale_b[2] == 0.0f) || is_box_b)) ||
                    (is_capsule_a && is_capsule_b);
    return handled_empty;
}

// ---------------------------------------------------------------- narrow_phase.py:458-1014 (rigid primitive subset)
static void narrow_phase(const o_model* m, const float* body_q, const std::vector<int>& pairs, const float* geom_data,
                         const float* geom_xform, const float* aabb_lower, const float* aabb_upper, o_contacts* ct) {
    std::vector<int> gjk_pairs;
    int npairs = (int)pairs.size() / 2;
    for (int t = 0; t < npairs; ++t) {
        int shape_a = pairs[2 * t], shape_b = pairs[2 * t + 1];
        if (shape_a == shape_b || shape_a < 0 || shape_b < 0) continue;
        int type_a = m->shape_type[shape_a], type_b = m->shape_type[shape_b];
        if (type_a > type_b) {
            std::swap(shape_a, shape_b);
            std::swap(type_a, type_b);
        }
        vec3 scale_a(geom_data[4 * shape_a], geom_data[4 * shape_a + 1], geom_data[4 * shape_a + 2]);
        vec3 scale_b(geom_data[4 * shape_b], geom_data[4 * shape_b + 1], geom_data[4 * shape_b + 2]);

        // mesh / heightfield routing (narrow_phase.py:552-640) is out of the rigid primitive scope
        if (type_a == GEO_HFIELD || type_b == GEO_HFIELD || type_a == GEO_MESH || type_b == GEO_MESH) continue;

        if (type_a >= GEO_ELLIPSOID || type_b == GEO_CONE || (type_a == GEO_CAPSULE && type_b > GEO_CAPSULE)) {
            gjk_pairs.push_back(shape_a);
            gjk_pairs.push_back(shape_b);
            continue;
        }
        float margin_offset_a = geom_data[4 * shape_a + 3];
        float margin_offset_b = geom_data[4 * shape_b + 3];
        transform X_a = ldx(geom_xform, shape_a), X_b = ldx(geom_xform, shape_b);
        float gap_sum = m->shape_gap[shape_a] + m->shape_gap[shape_b];

        float radius_eff_a = 0.0f, radius_eff_b = 0.0f;
        if (type_a == GEO_SPHERE || type_a == GEO_CAPSULE) radius_eff_a = scale_a[0];
        if (type_b == GEO_SPHERE || type_b == GEO_CAPSULE) radius_eff_b = scale_b[0];

        float dist[4];
        vec3 pos[4];
        vec3 contact_normal;
        bool handled;
        primitive_pair(type_a, type_b, X_a, X_b, scale_a, scale_b, gap_sum + margin_offset_a + margin_offset_b, dist, pos,
                       contact_normal, handled);

        int num_contacts = int(dist[0] < MAXVAL) + int(dist[1] < MAXVAL) + int(dist[2] < MAXVAL) + int(dist[3] < MAXVAL);
        if (num_contacts > 0) {
            ContactData cd;
            cd.contact_normal_a_to_b = contact_normal;
            cd.radius_eff_a = radius_eff_a;
            cd.radius_eff_b = radius_eff_b;
            cd.margin_a = margin_offset_a;
            cd.margin_b = margin_offset_b;
            cd.shape_a = shape_a;
            cd.shape_b = shape_b;
            cd.gap_sum = gap_sum;
            float total_separation_needed = radius_eff_a + radius_eff_b + margin_offset_a + margin_offset_b;
            vec3 nn = normalize(contact_normal);
            bool valid[4] = {false, false, false, false};
            int num_valid = 0;
            for (int k = 0; k < 4; ++k) {
                if (dist[k] < MAXVAL) {
                    cd.contact_point_center = pos[k];
                    cd.contact_distance = dist[k];
                    valid[k] = contact_passes_gap_check_precomputed(cd, nn, total_separation_needed);
                }
                num_valid += int(valid[k]);
            }
            if (num_valid > 0) {
                int base_index = ct->rigid_contact_count[0];
                ct->rigid_contact_count[0] += num_valid;
                if (base_index + num_valid > ct->rigid_contact_max) continue;
                for (int k = 0; k < 4; ++k) {
                    if (!valid[k]) continue;
                    write_contact_at(m, body_q, ct, base_index, shape_a, shape_b, pos[k], contact_normal, dist[k],
                                     radius_eff_a, radius_eff_b, margin_offset_a, margin_offset_b);
                    base_index += 1;
                }
            }
            continue;
        }
        if (handled) continue;
        gjk_pairs.push_back(shape_a);
        gjk_pairs.push_back(shape_b);
    }
    // second kernel: narrow_phase_kernel_gjk_mpr over the compacted GJK queue (narrow_phase.py:1040-1216)
    for (size_t t = 0; t < gjk_pairs.size() / 2; ++t)
        convex_pair_contacts(m, gjk_pairs[2 * t], gjk_pairs[2 * t + 1], geom_data, geom_xform, aabb_lower, aabb_upper, body_q,
                             ct);
}

extern "C" int o_collide(const o_model* m, const float* body_q, int broad_phase, o_contacts* ct, int32_t* out_pairs,
                         int out_pairs_cap, float* out_aabb_lower, float* out_aabb_upper) {
    const int S = m->shape_count;
    std::vector<float> lo(3 * S), hi(3 * S), gd(4 * S), gx(7 * S);
    ct->rigid_contact_count[0] = 0;  // thread 0 of compute_shape_aabbs zeroes the counters (collide.py:325-335)
    compute_shape_aabbs(m, body_q, lo.data(), hi.data(), gd.data(), gx.data());
    std::vector<int> pairs;
    if (broad_phase == O_BP_EXPLICIT)
        broadphase_explicit(m, lo.data(), hi.data(), pairs);
    else
        broadphase_nxn(m, lo.data(), hi.data(), pairs);  // SAP emits the same *set* (test_broad_phase.py:91-145)
    narrow_phase(m, body_q, pairs, gd.data(), gx.data(), lo.data(), hi.data(), ct);
    int np = (int)pairs.size() / 2;
    if (out_pairs)
        for (int i = 0; i < np && i < out_pairs_cap; ++i) {
            out_pairs[2 * i] = pairs[2 * i];
            out_pairs[2 * i + 1] = pairs[2 * i + 1];
        }
    if (out_aabb_lower) std::copy(lo.begin(), lo.end(), out_aabb_lower);
    if (out_aabb_upper) std::copy(hi.begin(), hi.end(), out_aabb_upper);
    return np;
}

// probe for known-answer tests: runs the analytic function for a type-sorted pair
extern "C" int o_probe_primitive(int type_a, int type_b, const float* xf_a, const float* xf_b, const float* scale_a,
                                 const float* scale_b, float margin, float* out_dist4, float* out_pos12, float* out_normal3) {
    float dist[4];
    vec3 pos[4];
    vec3 n;
    bool handled;
    primitive_pair(type_a, type_b, ldx(xf_a, 0), ldx(xf_b, 0), ld3(scale_a, 0), ld3(scale_b, 0), margin, dist, pos, n, handled);
    int cnt = 0;
    for (int k = 0; k < 4; ++k) {
        out_dist4[k] = dist[k];
        st3(out_pos12, k, pos[k]);
        cnt += dist[k] < MAXVAL;
    }
    st3(out_normal3, 0, n);
    return handled ? cnt : -1;
}
