"""World-range views of a finalized Model: ``slice_worlds`` (the per-rank shard of a global model)
and ``tile_worlds`` (a k-fold replication of an already finalized model, used by the env-count sweep of bench.py where the
Python builder would take minutes for 10^6 worlds).

The reference lays every per-world entity out world-major (model.py:881-900: ``*_world_start`` offsets), with the shared
world -1 shapes in front of and / or behind the local block; both functions keep that order, so a shard is itself a valid
Model (same EnvTemplate, smaller ``world_count``) and concatenating the shards' states in rank order reproduces the
unsharded state arrays.
"""
from __future__ import annotations

import copy

import numpy as np

_BODY = ("body_q", "body_qd", "body_com", "body_mass", "body_inertia", "body_inv_mass", "body_inv_inertia", "body_flags")
_JOINT = ("joint_type", "joint_enabled", "joint_X_p", "joint_X_c", "joint_dof_dim")
_DOF = ("joint_qd", "joint_f", "joint_target_qd", "joint_axis", "joint_limit_lower", "joint_limit_upper", "joint_limit_ke",
        "joint_limit_kd", "joint_target_ke", "joint_target_kd", "joint_damping", "joint_armature", "joint_effort_limit",
        "joint_velocity_limit", "joint_friction")
_SHAPE = ("shape_transform", "shape_type", "shape_scale", "shape_flags", "shape_collision_group", "shape_mesh_start",
          "shape_mesh_count", "shape_collision_aabb_lower", "shape_collision_aabb_upper", "shape_margin", "shape_gap",
          "shape_collision_radius", "shape_material_ke", "shape_material_kd", "shape_material_kf", "shape_material_ka",
          "shape_material_mu", "shape_material_restitution", "shape_material_mu_torsional", "shape_material_mu_rolling",
          "shape_material_kh", "_shape_sdf_index", "shape_edge_range", "_shape_voxel_resolution", "mesh_vertex_range", "mesh_triangle_range", "shape_heightfield_index")
# (the SDF table Model._texture_sdf_data, the edge tables mesh_edge_centers / _halves and the triangle meshes' vertex table
# mesh_vertices / index table mesh_indices are shared assets: copied by reference)


def _span(world_arr, b, e):
    """[first, last) index range of the entries whose world lies in [b, e) (entries are world-major and contiguous)."""
    w = np.asarray(world_arr)
    idx = np.flatnonzero((w >= b) & (w < e))
    if len(idx) == 0:
        return 0, 0
    if not np.array_equal(idx, np.arange(idx[0], idx[0] + len(idx))):
        raise NotImplementedError("worlds must be laid out contiguously (world-major)")
    return int(idx[0]), int(idx[0] + len(idx))


def slice_worlds(model, begin: int, end: int, device=None):
    """The sub-model holding worlds [begin, end) of `model` plus every global (world -1) shape, re-indexed from zero."""
    W = model.world_count
    if not (0 <= begin < end <= W):
        raise ValueError(f"bad world range [{begin}, {end}) for a model with {W} worlds")
    m = copy.copy(model)
    m._dev = None
    m.device = device if device is not None else model.device
    m.world_count = end - begin
    b0, b1 = _span(model.body_world, begin, end)
    j0, j1 = _span(model.joint_world, begin, end)
    a0, a1 = _span(model.articulation_world, begin, end) if model.articulation_count else (0, 0)
    for k in _BODY:
        setattr(m, k, np.array(getattr(model, k)[b0:b1]))
    m.body_world = np.asarray(model.body_world[b0:b1]) - begin
    m.body_label = list(model.body_label[b0:b1])
    m.body_count = b1 - b0
    m.gravity = np.concatenate([model.gravity[begin:end], model.gravity[W:W + 1]]).astype(np.float32)

    m.joint_count = j1 - j0
    for k in _JOINT:
        setattr(m, k, np.array(getattr(model, k)[j0:j1]))

    def shift(a, off):
        a = np.asarray(a)
        return np.where(a >= 0, a - off, a).astype(a.dtype)

    m.joint_parent = shift(model.joint_parent[j0:j1], b0)
    m.joint_child = shift(model.joint_child[j0:j1], b0)
    m.joint_ancestor = shift(model.joint_ancestor[j0:j1], j0)
    m.joint_articulation = shift(model.joint_articulation[j0:j1], a0)
    m.joint_world = np.asarray(model.joint_world[j0:j1]) - begin
    m.joint_label = list(model.joint_label[j0:j1])

    def edge(starts, total):  # [first, last) of a per-joint start table
        s = np.asarray(starts)
        first = int(s[j0]) if j1 > j0 else 0
        last = int(s[j1]) if j1 < len(s) else int(total)
        return first, last

    q0, q1 = edge(model.joint_q_start, model.joint_coord_count)
    d0, d1 = edge(model.joint_qd_start, model.joint_dof_count)
    t0, t1 = edge(model.joint_target_q_start, len(model.joint_target_q))
    m.joint_q_start = np.asarray(model.joint_q_start[j0:j1]) - q0
    m.joint_qd_start = np.asarray(model.joint_qd_start[j0:j1]) - d0
    m.joint_target_q_start = np.asarray(model.joint_target_q_start[j0:j1]) - t0
    m.joint_q = np.array(model.joint_q[q0:q1])
    m.joint_target_q = np.array(model.joint_target_q[t0:t1])
    for k in _DOF:
        setattr(m, k, np.array(getattr(model, k)[d0:d1]))
    m.joint_coord_count, m.joint_dof_count = q1 - q0, d1 - d0
    m.articulation_count = a1 - a0
    m.articulation_start = np.asarray(model.articulation_start[a0:a1]) - j0
    m.articulation_end = np.asarray(model.articulation_end[a0:a1]) - j0
    m.articulation_world = np.asarray(model.articulation_world[a0:a1]) - begin
    m.articulation_label = list(model.articulation_label[a0:a1])

    sw = np.asarray(model.shape_world)
    keep = np.flatnonzero((sw < 0) | ((sw >= begin) & (sw < end)))
    new_id = -np.ones(model.shape_count + 1, dtype=np.int64)
    new_id[keep] = np.arange(len(keep))
    for k in _SHAPE:
        if getattr(model, k, None) is not None:
            setattr(m, k, np.array(np.asarray(getattr(model, k))[keep]))
    m.shape_body = shift(np.asarray(model.shape_body)[keep], b0)
    m.shape_world = np.where(sw[keep] >= 0, sw[keep] - begin, -1).astype(np.int32)
    m.shape_label = [model.shape_label[i] for i in keep]
    m.shape_source = [model.shape_source[i] for i in keep]
    m.shape_count = len(keep)
    m._global_shape_ids = keep.astype(np.int64)  # shape id of the source model for every shape of the slice (hetero.py)
    m._world_groups = None
    m.shape_collision_filter_pairs = {(int(new_id[a]), int(new_id[b])) for a, b in model.shape_collision_filter_pairs
                                      if new_id[a] >= 0 and new_id[b] >= 0}
    pairs = np.asarray(model.shape_contact_pairs, dtype=np.int64).reshape(-1, 2)
    if len(pairs):
        pa, pb = new_id[pairs[:, 0]], new_id[pairs[:, 1]]
        ok = (pa >= 0) & (pb >= 0)
        pairs = np.stack([pa[ok], pb[ok]], axis=1)
    m.shape_contact_pairs = pairs.astype(np.int32).reshape(-1, 2)
    m.shape_contact_pair_count = len(m.shape_contact_pairs)
    m._build_env_template()
    return m


def tile_worlds(model, reps: int, device=None, filter_pairs: bool = True):
    """`reps` copies of `model`'s worlds back to back (world w of copy r becomes world r * W + w); global shapes stay
    single.  Equivalent to replicating the source builder reps * W times when the worlds only differ in per-world values
    (the arrays are copied verbatim, per-world jitter included).  ``filter_pairs=False`` leaves
    ``shape_collision_filter_pairs`` empty (a Python set of 78 tuples per quadruped is the slow part at 10^5+ worlds; the
    candidate pairs in ``shape_contact_pairs`` are already filtered, only the standalone broad phases read the set)."""
    if reps < 1:
        raise ValueError("reps must be >= 1")
    W = model.world_count
    if reps == 1:
        return slice_worlds(model, 0, W, device=device)
    t = model.env
    m = copy.copy(model)
    m._dev = None
    m.device = device if device is not None else model.device
    m.world_count = W * reps
    B, J, D, Q, A = model.body_count, model.joint_count, model.joint_dof_count, model.joint_coord_count, model.articulation_count
    TQ = len(model.joint_target_q)

    def rep(a):
        a = np.asarray(a)
        return np.tile(a, (reps,) + (1,) * (a.ndim - 1))

    def rep_off(a, step):  # tiled with a per-copy offset on the non-negative entries
        a = np.asarray(a)
        out = rep(a).astype(np.int64)
        off = np.repeat(np.arange(reps) * step, len(a))
        return np.where(out >= 0, out + off, out).astype(a.dtype)

    for k in _BODY:
        setattr(m, k, rep(getattr(model, k)))
    m.body_world = rep_off(model.body_world, W)
    m.body_label = list(model.body_label) * reps
    m.body_count = B * reps
    m.gravity = np.concatenate([rep(model.gravity[:W]), model.gravity[W:W + 1]]).astype(np.float32)
    for k in _JOINT:
        setattr(m, k, rep(getattr(model, k)))
    m.joint_parent = rep_off(model.joint_parent, B)
    m.joint_child = rep_off(model.joint_child, B)
    m.joint_ancestor = rep_off(model.joint_ancestor, J)
    m.joint_articulation = rep_off(model.joint_articulation, A)
    m.joint_world = rep_off(model.joint_world, W)
    m.joint_label = list(model.joint_label) * reps
    m.joint_q_start = rep_off(model.joint_q_start, Q)
    m.joint_qd_start = rep_off(model.joint_qd_start, D)
    m.joint_target_q_start = rep_off(model.joint_target_q_start, TQ)
    m.joint_q, m.joint_target_q = rep(model.joint_q), rep(model.joint_target_q)
    for k in _DOF:
        setattr(m, k, rep(getattr(model, k)))
    m.joint_count, m.joint_dof_count, m.joint_coord_count = J * reps, D * reps, Q * reps
    m.articulation_count = A * reps
    m.articulation_start = rep_off(model.articulation_start, J)
    m.articulation_end = rep_off(model.articulation_end, J)
    m.articulation_world = rep_off(model.articulation_world, W)
    m.articulation_label = list(model.articulation_label) * reps

    # shapes: [globals in front] [local block x reps] [globals behind]
    sw = np.asarray(model.shape_world)
    L0, nloc = t.shape_local0, t.ns * W
    front, local, back = np.arange(0, L0), np.arange(L0, L0 + nloc), np.arange(L0 + nloc, model.shape_count)
    order = np.concatenate([front, np.tile(local, reps), back])
    copy_of = np.concatenate([np.zeros(len(front), dtype=np.int64), np.repeat(np.arange(reps), nloc),
                              np.zeros(len(back), dtype=np.int64)])
    for k in _SHAPE:
        if getattr(model, k, None) is not None:
            setattr(m, k, np.array(np.asarray(getattr(model, k))[order]))
    sb = np.asarray(model.shape_body)[order].astype(np.int64)
    m.shape_body = np.where(sb >= 0, sb + copy_of * B, sb).astype(np.int32)
    m.shape_world = np.where(sw[order] >= 0, sw[order] + copy_of * W, -1).astype(np.int32)
    m.shape_label = [model.shape_label[i] for i in order]
    m.shape_source = [model.shape_source[i] for i in order]
    m.shape_count = len(order)

    def new_ids(col, r):  # old shape id -> id in the tiled model for copy r
        col = np.asarray(col, dtype=np.int64)
        return np.where(col < L0, col, np.where(col < L0 + nloc, col + r * nloc, col + (reps - 1) * nloc))

    filt = np.asarray(sorted(model.shape_collision_filter_pairs), dtype=np.int64).reshape(-1, 2)
    out = set()
    for r in range(reps if (len(filt) and filter_pairs) else 0):
        a, b = new_ids(filt[:, 0], r), new_ids(filt[:, 1], r)
        out.update(zip(a.tolist(), b.tolist()))
    m.shape_collision_filter_pairs = out
    pairs = np.asarray(model.shape_contact_pairs, dtype=np.int64).reshape(-1, 2)
    is_loc = ((pairs >= L0) & (pairs < L0 + nloc)).any(axis=1) if len(pairs) else np.zeros(0, dtype=bool)
    gg, lp = pairs[~is_loc], pairs[is_loc]  # global-global pairs come first in the reference's order
    tiled = [gg] + [np.stack([new_ids(lp[:, 0], r), new_ids(lp[:, 1], r)], axis=1) for r in range(reps)]
    m.shape_contact_pairs = np.concatenate(tiled).astype(np.int32).reshape(-1, 2)
    m.shape_contact_pair_count = len(m.shape_contact_pairs)
    m._build_env_template()
    return m
