/* newton_hip_mesh.h -- mesh legs of CollisionPipeline.collide that work on mesh VERTICES (extension of newton_hip.h).
 *
 * Reference interface replaced (paths relative to /root/reference/newton/_src/geometry):
 *   pair routing        narrow_phase.py:618-631   a MESH against an INFINITE plane (scale x = y = 0) leaves the primitive / GJK
 *                                                 path for `shape_pairs_mesh_plane`, stored as (mesh, plane)
 *   contact generation  narrow_phase.py:1744-1992 narrow_phase_process_mesh_plane_contacts(_reduce)_kernel: one lane per mesh
 *                                                 vertex -- world position, projection on the plane through the plane's frame,
 *                                                 distance = (v - proj) . n, admitted when distance < gap sum + margin sum,
 *                                                 contact centre = midpoint, normal = -n (mesh -> plane), sort_sub_key = vertex
 *   reduction           contact_reduction_global.py:2059-2096 write_contact_to_reducer, :1246-1346 reduce_contact_in_hashtable
 *                                                 (the BUFFERED variant: uncentred projection, directional slots only for
 *                                                 depth < 1e-4 |aabb(mesh)|, max-depth and voxel slots for every contact),
 *                                                 :2098-2290 export_reduced_contacts_kernel
 * Output = ContactData rows in the form nt_mesh_sdf_collide_reduced emits them (newton_hip.h): one contiguous block per pair,
 * rows in ascending vertex order -- what nt_sdf_rows_finalize / nt_contact_rows_write turn into Newton's contact arrays.
 * Same conventions as newton_hip.h: device pointers owned by the caller, work enqueued on `stream`, no allocation. */
#ifndef NEWTON_HIP_MESH_H
#define NEWTON_HIP_MESH_H

#include "newton_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t* pairs;                    /* [pair_count][2] shape ids, or the per-world candidate regions [worlds * pairs_per_world][2]
                                          of nt_sdf_candidate_pairs; rewritten in place as (mesh, plane) for the pairs processed */
    int32_t pair_count;                /* plain list: number of pairs (pair_world_prefix == NULL) */
    const int32_t* pair_world_prefix;  /* [worlds + 1] exclusive prefix of the live pairs per world, or NULL for a plain list */
    int32_t worlds, pairs_per_world;
    const uint8_t* pair_kind;          /* [pairs] or NULL: when given only pairs of kind NT_PAIR_KIND_MESH_PLANE are processed */
    const int32_t* shape_type;         /* [S] GeoType (PLANE = 1, MESH = 8): which shape of a pair is the plane */
    const float* shape_transform;      /* [S][7] world transforms */
    const float* shape_data;           /* [S][4] scale xyz, margin */
    const float* shape_gap;            /* [S] */
    const int32_t* shape_vertex_range; /* [S][2] (first vertex, vertex count) of a mesh shape in `vertices` */
    const float* vertices;             /* [V][3] mesh-local, unscaled (wp.Mesh.points) */
    const float* shape_aabb_lower;     /* [S][3] Model.shape_collision_aabb_lower / _upper: the mesh's scaled local AABB */
    const float* shape_aabb_upper;
    const int32_t* shape_voxel_res;    /* [S][3] Model.shape_voxel_resolution */
    int32_t reduce;                    /* 1: the global contact reduction (CollisionPipeline default); 0: every admitted vertex */
    int32_t* out_count;                /* [1] rows appended so far (the caller sets the start; keeps counting past capacity) */
    int32_t* out_pair;                 /* [capacity] pair position of the row */
    int32_t* out_key;                  /* [capacity] vertex index (ContactData.sort_sub_key) */
    float* out_data;                   /* [capacity][9] centre, normal mesh -> plane, distance, margin mesh, margin plane */
    int32_t capacity;
    int32_t* out_blk;                  /* [pairs][2] (first row, row count) of the pair's block; written for every processed pair */
} nt_mesh_plane_args;
#define NT_PAIR_KIND_MESH_PLANE 2      /* nt_sdf_scene.template_kind / world_pair_kind: 0 mesh-SDF edges, 1 hydroelastic */
nt_status nt_mesh_plane_pairs(const nt_mesh_plane_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * The triangle leg: a MESH that does not take the SDF route against a convex primitive (sphere, capsule, ellipsoid, cylinder, box,
 * cone) or a convex hull (CONVEX_MESH).  Reference interface replaced (paths relative to /root/reference/newton/_src/geometry):
 *   pair routing        narrow_phase.py:633-638    `shape_pairs_mesh`: a mesh against a non-mesh shape that took no earlier route
 *   midphase            narrow_phase.py:1455-1568 narrow_phase_find_mesh_triangle_overlaps_kernel -> collision_core.py:996-1180
 *                                                 (query AABB from the convex shape's support function in the unscaled mesh frame,
 *                                                 widened by (margin + gap) / |scale|; mesh BVH query; front-face test)
 *   contact generation  contact_reduction_global.py:2299-2403 mesh_triangle_contacts_to_reducer_kernel (reduce_contacts=True) and
 *                                                 narrow_phase.py:1571-1665 (reduce_contacts=False): world-space triangle, back-face
 *                                                 culling, GJK / MPR + manifold with GeoTypeEx.TRIANGLE as shape A,
 *                                                 sort_sub_key = (((triangle << 1) | 1) << 3) | manifold index
 *   reduction           contact_reduction_global.py:2059-2096, :1246-1346, :2098-2290 (the buffered variant, as nt_mesh_plane_pairs)
 * The BVH is replaced by a scan over the mesh's triangles (same candidate set: every triangle whose float32 bounds touch the query
 * box).  Output = ContactData rows like nt_mesh_plane_pairs: one contiguous block per pair, rows in ascending sort_sub_key order;
 * under `reduce` they are the reducer's survivors BEFORE the writer's gap test (nt_sdf_rows_finalize / nt_contact_rows_write apply
 * it, with the effective radii of `out_radius`), without it every generated contact.
 * --------------------------------------------------------------------------------------------------------------------------- */
/* HeightfieldData (newton/_src/utils/heightfield.py:141-156): a grid of nrow x ncol normalised elevations in [0, 1] at
 * `elevations[data_offset + row * ncol + col]`, spanning [-hx, hx] x [-hy, hy], world z = min_z + h (max_z - min_z). */
typedef struct {
    int32_t data_offset, nrow, ncol;
    float hx, hy, min_z, max_z;
} nt_heightfield;
typedef struct {
    int32_t* pairs;                    /* as nt_mesh_plane_args.pairs; rewritten in place as (mesh, convex) for the pairs processed */
    int32_t pair_count;
    const int32_t* pair_world_prefix;  /* [worlds + 1] or NULL for a plain list */
    int32_t worlds, pairs_per_world;
    const uint8_t* pair_kind;          /* [pairs] or NULL: when given only pairs of kind NT_PAIR_KIND_MESH_TRIANGLE are processed */
    const int32_t* shape_type;         /* [S] GeoType (MESH = 8; the partner: SPHERE 3, CAPSULE 4, ELLIPSOID 5, CYLINDER 6, BOX 7, CONE 9,
                                          CONVEX_MESH 10 with `hull_points`) */
    const float* shape_transform;      /* [S][7] world transforms */
    const float* shape_data;           /* [S][4] scale xyz, margin */
    const float* shape_gap;            /* [S] */
    const int32_t* shape_vertex_range; /* [S][2] (first vertex, vertex count) of a mesh shape in `vertices` */
    const int32_t* shape_triangle_range; /* [S][2] (first triangle, triangle count < 2^18) of a mesh shape in `indices` */
    const float* vertices;             /* [V][3] mesh-local, unscaled (wp.Mesh.points) */
    const int32_t* indices;            /* [T][3] vertex ids relative to the shape's first vertex (wp.Mesh.indices) */
    const float* shape_aabb_lower;     /* [S][3] Model.shape_collision_aabb_lower / _upper (reduce: the mesh's scaled local AABB) */
    const float* shape_aabb_upper;
    const int32_t* shape_voxel_res;    /* [S][3] Model.shape_voxel_resolution (reduce) */
    int32_t reduce;                    /* 1: the global contact reduction (CollisionPipeline default); 0: every generated contact */
    int32_t* out_count;                /* [1] rows appended so far (the caller sets the start; keeps counting past capacity) */
    int32_t* out_pair;                 /* [capacity] pair position of the row */
    int32_t* out_key;                  /* [capacity] sort_sub_key of the contact */
    float* out_data;                   /* [capacity][9] centre, normal mesh -> convex, distance, margin mesh, margin convex */
    float* out_radius;                 /* [capacity][2] or NULL: effective radii (0, sphere / capsule radius) of the export */
    int32_t capacity;
    int32_t* out_blk;                  /* [pairs][2] (first row, row count) of the pair's block; written for every processed pair */
    const float* block_bounds;         /* [blocks][6] or NULL: (lower xyz, upper xyz) of the unscaled vertices of every block of
                                          NT_MESH_TRIANGLE_BLOCK consecutive triangles of a mesh (the last block of a mesh is short);
                                          the scan skips blocks that miss the query box -- same candidate set, fewer rounds */
    const int32_t* shape_block_start;  /* [S] or NULL (both or neither): first block of a mesh shape in `block_bounds` */
    const float* hull_points;          /* [H][3] or NULL: vertex tables of the CONVEX_MESH partners (wp.Mesh.points of a hull, unscaled;
                                          Model.mesh_points); without it pairs with a CONVEX_MESH are skipped */
    const int32_t* shape_hull_range;   /* [S][2] or NULL (both or neither): (first vertex, vertex count) of a CONVEX_MESH shape */
    /* heightfields (GeoType.HFIELD = 2) as the mesh-like shape of a pair: narrow_phase.py:553-583 routes (heightfield, convex) pairs to
     * the same triangle kernels -- heightfield_vs_convex_midphase (utils/heightfield.py:366-462: the partner's LOCAL AABB
     * `shape_aabb_lower / _upper` as an oriented box in the heightfield frame -> a cell range, two triangles per cell) and
     * get_triangle_shape_from_heightfield (:280-363: GeoTypeEx.TRIANGLE_PRISM, the triangle extruded 1 m along the field's -Z, MPR /
     * GJK in the heightfield frame; penetrating contacts move to the physical face, collision_core.py:280-322).
     * sort_sub_key = ((((row * (ncol - 1) + col) * 2 + tri_sub) << 1 | 1) << 3) | manifold index; 2 (nrow - 1)(ncol - 1) < 2^18.
     * All three NULL: no heightfield pairs (they are skipped); the mesh tables may be NULL when only heightfields collide. */
    const int32_t* shape_heightfield_index; /* [S] index into `heightfields`, -1 for other shapes (Model.shape_heightfield_index) */
    const nt_heightfield* heightfields;     /* [H] */
    const float* elevations;                /* concatenated normalised elevation grids */
} nt_mesh_triangle_args;
#define NT_MESH_TRIANGLE_BLOCK 64
#define NT_PAIR_KIND_MESH_TRIANGLE 3
nt_status nt_mesh_triangle_pairs(const nt_mesh_triangle_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif
