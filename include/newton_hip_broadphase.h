/* newton_hip_broadphase.h -- swept-AABB variants of the standalone broad phases of libnewton_hip.so (extension of newton_hip.h).
 *
 * Reference interface replaced (paths relative to /root/reference): the optional `shape_displacement` /
 * `sort_axis_displacement_limit` arguments of
 *   BroadPhaseAllPairs.launch   newton/_src/geometry/broad_phase_nxn.py:309-427   (kernel :132-218)
 *   BroadPhaseExplicit.launch   newton/_src/geometry/broad_phase_nxn.py:441-535   (kernel :29-69)
 *   BroadPhaseSAP.launch        newton/_src/geometry/broad_phase_sap.py:631-848   (projection :44-79, pair test :273-316)
 * i.e. the broad phase of the speculative-contact mode: CollisionPipeline.collide(dt=...) hands every shape's world-space
 * displacement over the collision-update interval (collide.py:487-543,1877-1960) and a pair becomes a candidate when the
 * two gap-widened boxes overlap at ONE common time of that interval -- check_aabb_overlap_moving
 * (broad_phase_common.py:41-85: slab clipping of the relative displacement), not when their swept unions overlap.
 *
 * Kept in its own header so that the translation unit of the stepping kernels (which includes only newton_hip.h) is
 * byte-identical to the one the round-4 measurements were taken on; same conventions as newton_hip.h: device pointers owned
 * by the caller, work enqueued on `stream`, no allocation, count[0] added to and counting past `cap`. */
#ifndef NEWTON_HIP_BROADPHASE_H
#define NEWTON_HIP_BROADPHASE_H

#include "newton_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const float* displacement;          /* [n][3] world-space displacement of every shape over the interval [m]; required */
    float sort_axis_displacement_limit; /* SAP only: cap on |displacement . sort axis| when the projected interval is extended
                                         * (CollisionPipeline passes max_speculative_extension); < 0 = uncapped */
} nt_broadphase_motion;

/* nt_broadphase_nxn with the swept pair test; arguments as there */
nt_status nt_broadphase_nxn_swept(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* index_map,
                                  const int32_t* slice_ends, int32_t segments, int32_t num_regular_worlds, int32_t map_len,
                                  int32_t* pairs /*[cap][2]*/, int32_t* count, int32_t cap, void* stream);
/* nt_broadphase_sap_device with the projected interval of a shape extended by its (capped) displacement along the sort axis
 * and the swept pair test: like the reference, a pair is only tested when the extended intervals overlap */
nt_status nt_broadphase_sap_device_swept(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* index_map,
                                         const int32_t* slice_ends, int32_t segments, int32_t num_regular_worlds, int32_t map_len,
                                         int32_t max_segment, int32_t* sorted_map, float* projections, int32_t* pairs,
                                         int32_t* count, int32_t cap, void* stream);
/* nt_broadphase_explicit with the swept pair test */
nt_status nt_broadphase_explicit_swept(const nt_broadphase_in* in, const nt_broadphase_motion* motion, const int32_t* pair_list,
                                       int32_t n_pairs, int32_t* pairs, int32_t* count, int32_t cap, void* stream);

#ifdef __cplusplus
}
#endif
#endif
