"""newton.geometry broad phases on arbitrary AABB arrays (newton/_src/geometry/broad_phase_nxn.py:221-535,
broad_phase_sap.py:395-848, broad_phase_common.py:271-388): ``BroadPhaseAllPairs``, ``BroadPhaseSAP``, ``BroadPhaseExplicit``.

Same constructor / ``launch`` surface as the reference classes, with torch CUDA tensors where the reference takes
``wp.array``: lower / upper ``[n, 3]`` float32, gaps ``[n]`` float32, groups / worlds ``[n]`` int32, ``candidate_pair``
``[cap, 2]`` int32, ``candidate_pair_count`` ``[1]`` int32.  The kernels live in ``csrc/nt_broadphase.hip`` behind
``nt_broadphase_nxn`` / ``nt_broadphase_sap`` / ``nt_broadphase_explicit``; there is no CPU fallback.  Pairs are appended
in unspecified order (the reference appends atomically as well); ``sort_candidate_pairs`` gives a canonical order.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .enums import ShapeFlags


def precompute_world_map(shape_world, shape_flags=None):
    """(index_map, slice_ends): the colliding shapes of every world (ascending world id, ascending shape index) each
    followed by the shared world -1 shapes, then one trailing segment with only the shared shapes
    (broad_phase_common.py:271-388)."""
    world = np.asarray(shape_world, dtype=np.int64)
    if shape_flags is not None:
        flags = np.asarray(shape_flags)
        if flags.shape[0] != world.shape[0]:
            raise ValueError("shape_flags and shape_world must have the same length")
        keep = (flags & int(ShapeFlags.COLLIDE_SHAPES)) != 0
    else:
        keep = np.ones(world.shape[0], dtype=bool)
    if np.any(world < -1):
        bad = np.unique(world[world < -1]).tolist()
        raise ValueError(f"Invalid world IDs detected: {bad}. Only world ID -1 (global/shared) and non-negative IDs "
                         "(0, 1, 2, ...) are supported.")
    idx = np.flatnonzero(keep)
    shared = idx[world[idx] == -1]
    local = idx[world[idx] >= 0]
    local = local[np.argsort(world[local], kind="stable")]
    ids, starts = np.unique(world[local], return_index=True)
    bounds = list(starts) + [local.shape[0]]
    chunks, ends, pos = [], [], 0
    for k in range(len(ids)):
        seg = local[bounds[k]:bounds[k + 1]]
        chunks += [seg, shared]
        pos += seg.shape[0] + shared.shape[0]
        ends.append(pos)
    chunks.append(shared)
    ends.append(pos + shared.shape[0])
    index_map = np.concatenate(chunks).astype(np.int32) if chunks else np.zeros(0, dtype=np.int32)
    return index_map, np.asarray(ends, dtype=np.int32)


def sort_candidate_pairs(candidate_pair, count):
    """The first ``count`` pairs in lexicographic order (a deterministic view of the atomically appended list)."""
    import torch  # noqa: PLC0415

    n = min(int(count), candidate_pair.shape[0])
    p = candidate_pair[:n].to(torch.int64)
    return candidate_pair[:n][torch.argsort(p[:, 0] * (1 << 32) + p[:, 1])]


def _torch():
    import torch  # noqa: PLC0415

    return torch


def _dev_i32(x, device):
    torch = _torch()
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.int32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.int32), device=device)


def _check(t, dtype, name):
    torch = _torch()
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise TypeError(f"{name} must be a contiguous CUDA tensor of dtype {dtype}")
    return t


class _BroadPhaseBase:
    def __init__(self, device=None):
        torch = _torch()
        self._lib = _lib.load()  # raises loudly when the HIP extension is missing
        if not torch.cuda.is_available():
            raise _lib.NewtonHipError("newton_amd.geometry broad phases run only on an MI355X (no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda:0")

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _view(self, lower, upper, gap, group, world, filter_pairs, num_filter_pairs, shape_body, body_flags,
              include_static_kinematic_pairs, shape_displacement):
        torch = _torch()
        v = _lib.nt_broadphase_in()
        keep = [_check(lower, torch.float32, "shape_lower"), _check(upper, torch.float32, "shape_upper")]
        if shape_displacement is not None:
            # broad_phase_nxn.py:389-393,511-515 / broad_phase_sap.py:722-726: one displacement per shape bound
            if shape_displacement.shape[0] != lower.shape[0]:
                raise ValueError("shape_displacement length must match the shape bounds "
                                 f"({lower.shape[0]}), got {shape_displacement.shape[0]}")
            keep.append(_check(shape_displacement, torch.float32, "shape_displacement"))
        v.lower, v.upper = lower.data_ptr(), upper.data_ptr()
        if gap is not None and gap.numel() > 0:
            keep.append(_check(gap, torch.float32, "shape_gap"))
            v.gap = gap.data_ptr()
        if group is not None:
            keep.append(_check(group, torch.int32, "shape_collision_group"))
            v.group = group.data_ptr()
        if world is not None:
            keep.append(_check(world, torch.int32, "shape_world"))
            v.world = world.data_ptr()
        nf = 0
        if filter_pairs is not None:
            nf = int(filter_pairs.shape[0] if num_filter_pairs is None else num_filter_pairs)
            if nf > 0:
                keep.append(_check(filter_pairs, torch.int32, "filter_pairs"))
                v.filter_pairs = filter_pairs.data_ptr()
        v.num_filter_pairs = nf
        v.include_static_kinematic_pairs = int(bool(include_static_kinematic_pairs))
        if shape_body is not None and shape_body.numel() > 0:
            keep.append(_check(shape_body, torch.int32, "shape_body"))
            v.shape_body = shape_body.data_ptr()
            if body_flags is not None and body_flags.numel() > 0:
                keep.append(_check(body_flags, torch.int32, "body_flags"))
                v.body_flags = body_flags.data_ptr()
        return v, keep

    @staticmethod
    def _motion(shape_displacement, sort_axis_displacement_limit=None):
        """-> nt_broadphase_motion for the *_swept entry points (include/newton_hip_broadphase.h), or None for the static test
        (no array, or no shapes: check_aabb_overlap_moving falls back to check_aabb_overlap on an empty array,
        broad_phase_common.py:51-54)."""
        if sort_axis_displacement_limit is None:
            limit = -1.0  # broad_phase_sap.py:727-728: uncapped
        else:
            if not np.isfinite(sort_axis_displacement_limit) or sort_axis_displacement_limit < 0.0:
                raise ValueError("sort_axis_displacement_limit must be a non-negative finite number, "
                                 f"got {sort_axis_displacement_limit!r}")
            limit = float(sort_axis_displacement_limit)
        if shape_displacement is None or shape_displacement.numel() == 0:
            return None
        m = _lib.nt_broadphase_motion()
        m.displacement = shape_displacement.data_ptr()
        m.sort_axis_displacement_limit = limit
        return m

    @staticmethod
    def _out(candidate_pair, candidate_pair_count, skip_count_zero):
        torch = _torch()
        _check(candidate_pair, torch.int32, "candidate_pair")
        _check(candidate_pair_count, torch.int32, "candidate_pair_count")
        if candidate_pair.dim() != 2 or candidate_pair.shape[1] != 2:
            raise ValueError("candidate_pair must have shape [max_candidate_pair, 2]")
        if not skip_count_zero:
            candidate_pair_count.zero_()
        return int(candidate_pair.shape[0])


class BroadPhaseAllPairs(_BroadPhaseBase):
    """All pairs inside every world segment (broad_phase_nxn.py:221-535)."""

    def __init__(self, shape_world, shape_flags=None, device=None):
        torch = _torch()
        if isinstance(shape_world, torch.Tensor) and device is None and shape_world.is_cuda:
            device = shape_world.device
        super().__init__(device)
        w = shape_world.cpu().numpy() if isinstance(shape_world, torch.Tensor) else shape_world
        f = shape_flags.cpu().numpy() if isinstance(shape_flags, torch.Tensor) else shape_flags
        index_map, slice_ends = precompute_world_map(w, f)
        self.num_regular_worlds = max(0, len(slice_ends) - 1)
        self.world_index_map = _dev_i32(index_map, self.device)
        self.world_slice_ends = _dev_i32(slice_ends, self.device)
        n = np.diff(np.concatenate([[0], slice_ends])).astype(np.int64)
        self.num_kernel_threads = int(np.sum(n * (n - 1) // 2))  # pair tests per launch, like the reference attribute

    def _map_for_launch(self, view_keep, lower, gap):
        return self.world_index_map

    _entry = "nt_broadphase_nxn"

    def launch(self, shape_lower, shape_upper, shape_gap, shape_collision_group, shape_world, shape_count, candidate_pair,
               candidate_pair_count, device=None, filter_pairs=None, num_filter_pairs=None, skip_count_zero=False, *,
               shape_body=None, body_flags=None, include_static_kinematic_pairs=True, shape_displacement=None):
        cap = self._out(candidate_pair, candidate_pair_count, skip_count_zero)
        v, keep = self._view(shape_lower, shape_upper, shape_gap, shape_collision_group, shape_world, filter_pairs,
                             num_filter_pairs, shape_body, body_flags, include_static_kinematic_pairs, shape_displacement)
        m = self._map_for_launch(keep, shape_lower, shape_gap)
        motion = self._motion(shape_displacement)
        entry = self._entry if motion is None else self._entry + "_swept"
        head = (C.byref(v),) if motion is None else (C.byref(v), C.byref(motion))
        _lib.check(getattr(self._lib, entry)(*head, m.data_ptr(), self.world_slice_ends.data_ptr(),
                                             int(self.world_slice_ends.shape[0]), int(self.num_regular_worlds), int(m.shape[0]),
                                             candidate_pair.data_ptr(), candidate_pair_count.data_ptr(), cap, self._stream()), entry)


class BroadPhaseSAP(BroadPhaseAllPairs):
    """Sort and sweep inside every world segment (broad_phase_sap.py:395-848), entirely on the device: the gap-widened AABBs
    are projected on the reference's fixed axis, every world segment is sorted in LDS by one workgroup (bitonic network,
    padding keys 1e30 like the reference's tile sort) and each shape sweeps forward until the projected intervals stop
    overlapping (nt_broadphase_sap_device).  Emits the same pair set as ``BroadPhaseAllPairs``; worlds with more than 4096
    colliding shapes are outside the LDS tile and raise."""

    def __init__(self, shape_world, shape_flags=None, sweep_thread_count_multiplier: int = 5, sort_type="segmented",
                 tile_block_dim=None, device=None):
        if sort_type not in ("segmented", "tile"):
            raise ValueError(f"sort_type must be 'segmented' or 'tile', got {sort_type!r}")
        super().__init__(shape_world, shape_flags, device)
        self.sort_type = sort_type  # both names run the same per-segment LDS sort here
        torch = _torch()
        ends = self.world_slice_ends.cpu().numpy().astype(np.int64)
        self._max_segment = int(np.max(np.diff(np.concatenate([[0], ends])))) if len(ends) else 0
        n = int(self.world_index_map.shape[0])
        self._sorted_map = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        self._proj = torch.zeros((2, max(n, 1)), dtype=torch.float32, device=self.device)

    def launch(self, shape_lower, shape_upper, shape_gap, shape_collision_group, shape_world, shape_count, candidate_pair,
               candidate_pair_count, device=None, filter_pairs=None, num_filter_pairs=None, skip_count_zero=False, *,
               shape_body=None, body_flags=None, include_static_kinematic_pairs=True, shape_displacement=None,
               sort_axis_displacement_limit=None):
        cap = self._out(candidate_pair, candidate_pair_count, skip_count_zero)
        v, keep = self._view(shape_lower, shape_upper, shape_gap, shape_collision_group, shape_world, filter_pairs,
                             num_filter_pairs, shape_body, body_flags, include_static_kinematic_pairs, shape_displacement)
        motion = self._motion(shape_displacement, sort_axis_displacement_limit)
        entry = "nt_broadphase_sap_device" if motion is None else "nt_broadphase_sap_device_swept"
        head = (C.byref(v),) if motion is None else (C.byref(v), C.byref(motion))
        _lib.check(getattr(self._lib, entry)(
            *head, self.world_index_map.data_ptr(), self.world_slice_ends.data_ptr(), int(self.world_slice_ends.shape[0]),
            int(self.num_regular_worlds), int(self.world_index_map.shape[0]), self._max_segment, self._sorted_map.data_ptr(),
            self._proj.data_ptr(), candidate_pair.data_ptr(), candidate_pair_count.data_ptr(), cap, self._stream()), entry)


class BroadPhaseExplicit(_BroadPhaseBase):
    """AABB test over a precomputed pair list (broad_phase_nxn.py:29-69, BroadPhaseExplicit)."""

    def launch(self, shape_lower, shape_upper, shape_gap, shape_pairs, shape_pair_count, candidate_pair, candidate_pair_count,
               device=None, skip_count_zero=False, *, shape_body=None, body_flags=None, include_static_kinematic_pairs=True,
               shape_displacement=None):
        torch = _torch()
        cap = self._out(candidate_pair, candidate_pair_count, skip_count_zero)
        v, keep = self._view(shape_lower, shape_upper, shape_gap, None, None, None, 0, shape_body, body_flags,
                             include_static_kinematic_pairs, shape_displacement)
        _check(shape_pairs, torch.int32, "shape_pairs")
        motion = self._motion(shape_displacement)
        entry = "nt_broadphase_explicit" if motion is None else "nt_broadphase_explicit_swept"
        head = (C.byref(v),) if motion is None else (C.byref(v), C.byref(motion))
        _lib.check(getattr(self._lib, entry)(*head, shape_pairs.data_ptr(), int(shape_pair_count), candidate_pair.data_ptr(),
                                             candidate_pair_count.data_ptr(), cap, self._stream()), entry)


class HydroelasticSDF:
    """newton.geometry.HydroelasticSDF (sdf_hydroelastic.py:365-560): the configuration object ``CollisionPipeline(
    sdf_hydroelastic_config=HydroelasticSDF.Config(...))`` takes.  The stages themselves are device kernels of the collide pipeline
    (csrc/nt_sdf.hip: nt_hydro_pairs); the buffer-sizing knobs of the reference (buffer_fraction, buffer_mult_*, grid_size) have
    no counterpart -- the octree of a pair lives in the LDS of the pair's workgroup -- and are accepted and ignored."""

    from dataclasses import dataclass as _dataclass

    @_dataclass
    class Config:
        reduce_contacts: bool = True
        pre_prune_contacts: bool = True
        buffer_fraction: float = 1.0
        buffer_mult_broad: int = 1
        buffer_mult_iso: int = 1
        buffer_mult_contact: int = 1
        contact_buffer_fraction: float = 0.5
        contact_reduction_hashtable_size_factor: float = 0.25
        grid_size: int = 256 * 8 * 128
        output_contact_surface: bool = False
        normal_matching: bool = True
        anchor_contact: bool = False
        moment_matching: bool = False
        margin_contact_area: float = 1.0e-2
        pressure_func: object = None
        pressure_data: object = None
        mc_edge_clamp_min: float = 0.02

        def __post_init__(self):
            if not (0.0 <= self.mc_edge_clamp_min <= 0.5):
                raise ValueError(f"mc_edge_clamp_min must be in [0.0, 0.5], got {self.mc_edge_clamp_min}")
            if not (0.0 < self.buffer_fraction <= 1.0):
                raise ValueError(f"buffer_fraction must be in (0, 1], got {self.buffer_fraction}")
            if self.moment_matching:
                self.anchor_contact = True
