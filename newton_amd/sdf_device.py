"""Device side of the texture SDFs: upload a newton_amd.sdf.TextureSDF, sample it on the MI355X (nt_sdf_sample) and run the
mesh-vs-SDF narrow phase (nt_mesh_sdf_collide, newton/_src/geometry/sdf_contact.py:1098-1515 with reduce_contacts=False).

The inputs are Newton's flat arrays (world shape transforms, shape_data = scale + margin, shape_gap, shape_sdf_index,
shape_edge_range / mesh_edge_centers / mesh_edge_halves); the outputs are ContactData rows appended through an atomic counter
and ordered afterwards by the reference's contact sort key (pair, edge, mode) -- contact_data.py:60-90."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .sdf import QuantizationMode, TextureSDF


def _torch():
    import torch  # noqa: PLC0415

    return torch


class DeviceSDF:
    """A TextureSDF resident in HBM + its nt_sdf descriptor."""

    def __init__(self, sdf: TextureSDF, device="cuda:0"):
        torch = _torch()
        self.host = sdf
        self.device = torch.device(device)
        self.coarse = torch.from_numpy(np.ascontiguousarray(sdf.coarse)).to(self.device)
        sub = np.ascontiguousarray(sdf.subgrid)
        if sub.dtype == np.uint16:  # torch has no uint16 arithmetic, but the bytes are all the kernel needs
            self.subgrid = torch.from_numpy(sub.view(np.int16)).to(self.device)
        else:
            self.subgrid = torch.from_numpy(sub).to(self.device)
        self.slots = torch.from_numpy(np.ascontiguousarray(sdf.slots).view(np.int32)).to(self.device)
        d = _lib.nt_sdf()
        d.coarse, d.subgrid, d.slots = self.coarse.data_ptr(), self.subgrid.data_ptr(), self.slots.data_ptr()
        d.cx, d.cy, d.cz = (int(x) for x in sdf.slots.shape)
        d.tex_size, d.subgrid_size = int(sdf.subgrid.shape[0]), int(sdf.subgrid_size)
        d.quantization, d.scale_baked = int(sdf.quantization_mode), int(bool(sdf.scale_baked))
        for k in range(3):
            d.box_lower[k], d.box_upper[k] = float(sdf.box_lower[k]), float(sdf.box_upper[k])
            d.inv_dx[k], d.voxel_size[k] = float(sdf.inv_dx[k]), float(sdf.voxel_size[k])
        d.voxel_radius, d.min_value, d.value_range = float(sdf.voxel_radius), float(sdf.min_value), float(sdf.value_range)
        self.desc = d

    def sample(self, points, grad: bool = False):
        """texture_sample_sdf (and the narrow phase's centred-difference gradient) at local points [N,3] (tensor or array)."""
        torch = _torch()
        lib = _lib.load()
        p = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32) if not isinstance(points, torch.Tensor) else points,
                            dtype=torch.float32, device=self.device).contiguous()
        n = p.shape[0]
        dist = torch.empty(n, dtype=torch.float32, device=self.device)
        g = torch.empty((n, 3), dtype=torch.float32, device=self.device) if grad else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(lib.nt_sdf_sample(C.byref(self.desc), p.data_ptr(), n, dist.data_ptr(), g.data_ptr() if grad else None, stream),
                   "nt_sdf_sample")
        return (dist, g) if grad else dist


def mesh_sdf_collide(pairs, shape_transform, shape_data, shape_gap, shape_sdf_index, sdfs, shape_edge_range, edge_centers,
                     edge_halves, capacity: int | None = None, device="cuda:0", reduce=None):
    """Run nt_mesh_sdf_collide; returns a dict of numpy arrays sorted by (pair, key): pair, key, center [n,3], normal [n,3],
    distance, margin0, margin1, plus `count` (the atomic counter, which keeps counting past the capacity).
    `sdfs`: list of DeviceSDF (or None) indexed by shape_sdf_index.
    `reduce`: (shape_collision_aabb_lower, shape_collision_aabb_upper, shape_voxel_resolution) -- see
    newton_amd.sdf.mesh_reduction_tables -- runs nt_mesh_sdf_collide_reduced instead: the reference's global contact reduction
    (GlobalContactReducer, deterministic packing) fused into the pair's workgroup."""
    torch = _torch()
    lib = _lib.load()
    dev = torch.device(device)

    def up(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        return torch.from_numpy(a if a.size else np.zeros(1, dtype=dtype)).to(dev)

    pairs_np = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    t_pairs, t_X = up(pairs_np, np.int32), up(shape_transform, np.float32)
    t_data, t_gap, t_idx = up(shape_data, np.float32), up(shape_gap, np.float32), up(shape_sdf_index, np.int32)
    t_er, t_ec, t_eh = up(shape_edge_range, np.int32), up(edge_centers, np.float32), up(edge_halves, np.float32)
    table = (_lib.nt_sdf * max(len(sdfs), 1))()
    for k, s in enumerate(sdfs):
        if s is not None:
            table[k] = s.desc
    t_table = torch.from_numpy(np.frombuffer(bytes(table), dtype=np.uint8).copy()).to(dev)
    if capacity is None:
        er = np.asarray(shape_edge_range, dtype=np.int64).reshape(-1, 2)
        capacity = int(sum(er[a, 1] + er[b, 1] for a, b in pairs_np)) + 1  # unreduced: at most one contact per edge and mode
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    o_pair = torch.full((capacity,), -1, dtype=torch.int32, device=dev)
    o_key = torch.zeros(capacity, dtype=torch.int32, device=dev)
    o_data = torch.zeros((capacity, 9), dtype=torch.float32, device=dev)
    a = _lib.nt_mesh_sdf_args()
    a.pairs, a.pair_count = t_pairs.data_ptr(), len(pairs_np)
    a.shape_transform, a.shape_data, a.shape_gap = t_X.data_ptr(), t_data.data_ptr(), t_gap.data_ptr()
    a.shape_sdf_index, a.sdf_table, a.sdf_count = t_idx.data_ptr(), t_table.data_ptr(), len(sdfs)
    a.shape_edge_range, a.edge_centers, a.edge_halves = t_er.data_ptr(), t_ec.data_ptr(), t_eh.data_ptr()
    a.out_count, a.out_pair, a.out_key, a.out_data, a.capacity = (count.data_ptr(), o_pair.data_ptr(), o_key.data_ptr(),
                                                                   o_data.data_ptr(), capacity)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if reduce is not None:
        r = _lib.nt_contact_reduce_shapes()
        t_lo, t_hi, t_res = up(reduce[0], np.float32), up(reduce[1], np.float32), up(reduce[2], np.int32)
        r.shape_aabb_lower, r.shape_aabb_upper, r.shape_voxel_res = t_lo.data_ptr(), t_hi.data_ptr(), t_res.data_ptr()
        _lib.check(lib.nt_mesh_sdf_collide_reduced(C.byref(a), C.byref(r), stream), "nt_mesh_sdf_collide_reduced")
    else:
        _lib.check(lib.nt_mesh_sdf_collide(C.byref(a), stream), "nt_mesh_sdf_collide")
    torch.cuda.current_stream(dev).synchronize()
    n_total = int(count.item())
    n = min(n_total, capacity)
    pair, key, data = o_pair[:n].cpu().numpy(), o_key[:n].cpu().numpy(), o_data[:n].cpu().numpy()
    order = np.lexsort((key, pair))
    pair, key, data = pair[order], key[order], data[order]
    return {"count": n_total, "pair": pair, "key": key, "center": data[:, 0:3], "normal": data[:, 3:6], "distance": data[:, 6],
            "margin0": data[:, 7], "margin1": data[:, 8]}


def hydro_collide(pairs, shape_transform, shape_data, shape_gap, shape_kh, shape_sdf_index, sdfs, capacity: int = 1 << 18,
                  margin_contact_area: float = 1.0e-2, edge_clamp_min: float = 0.02, device="cuda:0"):
    """Run nt_hydro_collide (HydroelasticSDF.launch, unreduced): returns numpy arrays sorted by (pair, fingerprint): pair, key,
    shape_a, shape_b, center [n,3], normal [n,3] (a -> b), distance (margin-relative separation), stiffness
    (Contacts.rigid_contact_stiffness), area, pressure, and `count`."""
    from .mc_tables import tables  # noqa: PLC0415

    torch = _torch()
    lib = _lib.load()
    dev = torch.device(device)

    def up(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        return torch.from_numpy(a if a.size else np.zeros(1, dtype=dtype)).to(dev)

    pairs_np = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    tri_range, flat = tables()
    t_pairs, t_X, t_data = up(pairs_np, np.int32), up(shape_transform, np.float32), up(shape_data, np.float32)
    t_gap, t_kh, t_idx = up(shape_gap, np.float32), up(shape_kh, np.float32), up(shape_sdf_index, np.int32)
    t_tr, t_fe = up(tri_range, np.int32), up(flat, np.uint8)
    table = (_lib.nt_sdf * max(len(sdfs), 1))()
    for k, s in enumerate(sdfs):
        if s is not None:
            table[k] = s.desc
    t_table = torch.from_numpy(np.frombuffer(bytes(table), dtype=np.uint8).copy()).to(dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    o_pair = torch.full((capacity,), -1, dtype=torch.int32, device=dev)
    o_key = torch.zeros(capacity, dtype=torch.int32, device=dev)
    o_shapes = torch.zeros((capacity, 2), dtype=torch.int32, device=dev)
    o_data = torch.zeros((capacity, 10), dtype=torch.float32, device=dev)
    a = _lib.nt_hydro_args()
    a.pairs, a.pair_count = t_pairs.data_ptr(), len(pairs_np)
    a.shape_transform, a.shape_data, a.shape_gap, a.shape_kh = t_X.data_ptr(), t_data.data_ptr(), t_gap.data_ptr(), t_kh.data_ptr()
    a.shape_sdf_index, a.sdf_table, a.sdf_count = t_idx.data_ptr(), t_table.data_ptr(), len(sdfs)
    a.tri_range, a.flat_edge_verts = t_tr.data_ptr(), t_fe.data_ptr()
    a.margin_contact_area, a.edge_clamp_min = float(margin_contact_area), float(edge_clamp_min)
    a.out_count, a.out_pair, a.out_key, a.out_shapes, a.out_data, a.capacity = (count.data_ptr(), o_pair.data_ptr(), o_key.data_ptr(),
                                                                                 o_shapes.data_ptr(), o_data.data_ptr(), capacity)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.nt_hydro_collide(C.byref(a), stream), "nt_hydro_collide")
    torch.cuda.current_stream(dev).synchronize()
    n_total = int(count.item())
    n = min(n_total, capacity)
    pair, key = o_pair[:n].cpu().numpy(), o_key[:n].cpu().numpy()
    shapes, data = o_shapes[:n].cpu().numpy(), o_data[:n].cpu().numpy()
    order = np.lexsort((key, pair))
    pair, key, shapes, data = pair[order], key[order], shapes[order], data[order]
    return {"count": n_total, "pair": pair, "key": key, "shape_a": shapes[:, 0], "shape_b": shapes[:, 1], "center": data[:, 0:3],
            "normal": data[:, 3:6], "distance": data[:, 6], "stiffness": data[:, 7], "area": data[:, 8], "pressure": data[:, 9]}


class MeshSdfNarrowPhase:
    """Device-resident mesh-vs-SDF narrow phase with the global contact reduction: the tables a Model carries for it (per-shape
    scale / margin, gap, SDF index, edge range, edge tables, local AABB + voxel resolution) are uploaded once; `launch` takes
    the broad phase's candidate pairs and counter and the shapes' world transforms as device tensors and appends reduced
    ContactData rows -- no host round trip, everything on the caller's stream.  Mirrors the mesh-mesh leg of
    NarrowPhase.launch (narrow_phase.py:2588-2760) with reduce_contacts=True, deterministic packing."""

    def __init__(self, shape_data, shape_gap, shape_sdf_index, sdfs, shape_edge_range, edge_centers, edge_halves, reduce_tables,
                 device="cuda:0"):
        torch = _torch()
        self._lib = _lib.load()
        self.device = torch.device(device)

        def up(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            return torch.from_numpy(a if a.size else np.zeros(1, dtype=dtype)).to(self.device)

        self._sdfs = list(sdfs)
        table = (_lib.nt_sdf * max(len(sdfs), 1))()
        for k, s in enumerate(sdfs):
            if s is not None:
                table[k] = s.desc
        self._t = dict(data=up(shape_data, np.float32), gap=up(shape_gap, np.float32), idx=up(shape_sdf_index, np.int32),
                       er=up(shape_edge_range, np.int32), ec=up(edge_centers, np.float32), eh=up(edge_halves, np.float32),
                       table=torch.from_numpy(np.frombuffer(bytes(table), dtype=np.uint8).copy()).to(self.device),
                       lo=up(reduce_tables[0], np.float32), hi=up(reduce_tables[1], np.float32), res=up(reduce_tables[2], np.int32))
        self.shape_count = int(np.asarray(shape_gap).shape[0])

    def launch(self, shape_transform, pairs, pair_count, out_count, out_pair, out_key, out_data, reduce: bool = True,
               threads: int = 0):
        """shape_transform [S,7] float32, pairs [P,2] int32, pair_count [1] int32 (device), outputs as nt_mesh_sdf_args; the
        caller zeroes out_count."""
        torch = _torch()
        t = self._t
        a = _lib.nt_mesh_sdf_args()
        a.pairs, a.pair_count, a.pair_count_device = pairs.data_ptr(), int(pairs.shape[0]), pair_count.data_ptr()
        a.shape_transform, a.shape_data, a.shape_gap = shape_transform.data_ptr(), t["data"].data_ptr(), t["gap"].data_ptr()
        a.shape_sdf_index, a.sdf_table, a.sdf_count = t["idx"].data_ptr(), t["table"].data_ptr(), len(self._sdfs)
        a.shape_edge_range, a.edge_centers, a.edge_halves = t["er"].data_ptr(), t["ec"].data_ptr(), t["eh"].data_ptr()
        a.out_count, a.out_pair, a.out_key, a.out_data, a.capacity = (out_count.data_ptr(), out_pair.data_ptr(),
                                                                       out_key.data_ptr(), out_data.data_ptr(),
                                                                       int(out_pair.shape[0]))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if reduce:
            r = _lib.nt_contact_reduce_shapes()
            r.shape_aabb_lower, r.shape_aabb_upper, r.shape_voxel_res = t["lo"].data_ptr(), t["hi"].data_ptr(), t["res"].data_ptr()
            r.threads = int(threads)
            _lib.check(self._lib.nt_mesh_sdf_collide_reduced(C.byref(a), C.byref(r), stream), "nt_mesh_sdf_collide_reduced")
        else:
            _lib.check(self._lib.nt_mesh_sdf_collide(C.byref(a), stream), "nt_mesh_sdf_collide")
