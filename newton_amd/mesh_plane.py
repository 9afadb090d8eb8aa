"""MESH vs infinite plane, stand-alone (newton/_src/geometry/narrow_phase.py:618-631 routing, :1744-1992 the vertex kernels,
contact_reduction_global.py:1246-1346,2059-2290 the buffered reduction + export): ``mesh_plane_contacts`` runs
``nt_mesh_plane_pairs`` (csrc/nt_mesh_plane.hip, include/newton_hip_mesh.h) on torch CUDA tensors in Newton's flat layout and
returns the ContactData rows per pair -- the rows ``nt_contact_rows_write`` / ``nt_sdf_rows_finalize`` turn into Newton's contact
arrays.  Inside ``CollisionPipeline.collide`` the same entry point runs as pair kind 2 of the SDF leg
(``newton_amd/sdf_pipeline.py``); this module is the stand-alone form on caller-provided arrays.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

from . import _lib


def mesh_plane_contacts(pairs, shape_type, shape_transform, shape_data, shape_gap, shape_vertex_range, vertices,
                        shape_collision_aabb_lower, shape_collision_aabb_upper, shape_voxel_resolution, reduce_contacts: bool = True,
                        capacity: int | None = None):
    """pairs: [P, 2] int32 shape ids in any order (rewritten in place as (mesh, plane)); the other arguments are the Model arrays
    named like them (float32 / int32 CUDA tensors, contiguous).  -> dict(count, blk [P, 2] (first row, rows) per pair, pair [n],
    vertex [n], data [n, 9] = centre, normal mesh -> plane, distance, margin mesh, margin plane); rows of a pair are contiguous and
    in ascending vertex order; ``count`` may exceed ``capacity`` (the blocks are clamped)."""
    import torch  # noqa: PLC0415

    lib = _lib.load()
    i32, f32 = torch.int32, torch.float32
    tensors = dict(pairs=(pairs, i32), shape_type=(shape_type, i32), shape_transform=(shape_transform, f32), shape_data=(shape_data, f32),
                   shape_gap=(shape_gap, f32), shape_vertex_range=(shape_vertex_range, i32), vertices=(vertices, f32),
                   shape_collision_aabb_lower=(shape_collision_aabb_lower, f32), shape_collision_aabb_upper=(shape_collision_aabb_upper, f32),
                   shape_voxel_resolution=(shape_voxel_resolution, i32))
    for name, (t, dt) in tensors.items():
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != dt or not t.is_contiguous():
            raise TypeError(f"{name} must be a contiguous CUDA tensor of dtype {dt}")
    dev = pairs.device
    P = int(pairs.shape[0])
    if reduce_contacts and int(shape_vertex_range[:, 1].max().item()) >= (1 << 22):
        # the reduction's packed values carry the vertex index in 22 bits (contact_reduction_global.py: FINGERPRINT bits): larger
        # meshes would alias in the table and the winners would be recomputed from truncated indices
        raise NotImplementedError("reduce_contacts=True supports triangle meshes with fewer than 2^22 vertices")
    if capacity is None:
        capacity = int(shape_vertex_range[:, 1].sum().item()) * max(P, 1)
    out = dict(count=torch.zeros(1, dtype=i32, device=dev), blk=torch.zeros((max(P, 1), 2), dtype=i32, device=dev),
               pair=torch.full((max(capacity, 1),), -1, dtype=i32, device=dev), vertex=torch.full((max(capacity, 1),), -1, dtype=i32, device=dev),
               data=torch.zeros((max(capacity, 1), 9), dtype=f32, device=dev))
    a = _lib.nt_mesh_plane_args()
    a.pairs, a.pair_count = pairs.data_ptr(), P
    a.shape_type, a.shape_transform, a.shape_data, a.shape_gap = (shape_type.data_ptr(), shape_transform.data_ptr(),
                                                                   shape_data.data_ptr(), shape_gap.data_ptr())
    a.shape_vertex_range, a.vertices = shape_vertex_range.data_ptr(), vertices.data_ptr()
    a.shape_aabb_lower, a.shape_aabb_upper, a.shape_voxel_res = (shape_collision_aabb_lower.data_ptr(), shape_collision_aabb_upper.data_ptr(),
                                                                 shape_voxel_resolution.data_ptr())
    a.reduce = int(bool(reduce_contacts))
    a.out_count, a.out_pair, a.out_key, a.out_data, a.capacity, a.out_blk = (out["count"].data_ptr(), out["pair"].data_ptr(),
                                                                            out["vertex"].data_ptr(), out["data"].data_ptr(),
                                                                            int(capacity), out["blk"].data_ptr())
    _lib.check(lib.nt_mesh_plane_pairs(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "nt_mesh_plane_pairs")
    return out
