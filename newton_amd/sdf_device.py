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
                     edge_halves, capacity: int | None = None, device="cuda:0", reduce=None, staged: bool = False):
    """Run nt_mesh_sdf_collide; returns a dict of numpy arrays sorted by (pair, key): pair, key, center [n,3], normal [n,3],
    distance, margin0, margin1, plus `count` (the atomic counter, which keeps counting past the capacity).
    `sdfs`: list of DeviceSDF (or None) indexed by shape_sdf_index.
    `reduce`: (shape_collision_aabb_lower, shape_collision_aabb_upper, shape_voxel_resolution) -- see
    newton_amd.sdf.mesh_reduction_tables -- runs nt_mesh_sdf_collide_reduced instead: the reference's global contact reduction
    (GlobalContactReducer, deterministic packing) fused into the pair's workgroup; `staged` selects the three dense launches
    (cull over all pairs -> one lane per survivor -> reduction per pair) that the collide pipeline uses: same rows."""
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
        if staged:
            er = np.asarray(shape_edge_range, dtype=np.int64).reshape(-1, 2)
            stripes = 4
            # at most every edge of both modes survives; four times that, so that no stripe of an uneven deal runs full
            hcap = (int(sum(er[p, 1] + er[q, 1] for p, q in pairs_np)) + stripes) * 4 // stripes * stripes
            npairs = max(len(pairs_np), 1)
            h_count = torch.zeros(4, dtype=torch.int32, device=dev)
            h_stripes = torch.zeros(stripes * 16, dtype=torch.int32, device=dev)
            u_ctx = torch.zeros((npairs, 2, 24), dtype=torch.float32, device=dev)
            h_pair, h_fp = torch.zeros(hcap, dtype=torch.int32, device=dev), torch.zeros(hcap, dtype=torch.int32, device=dev)
            h_rec = torch.zeros((hcap, 8), dtype=torch.float32, device=dev)
            h_blk = torch.zeros((npairs, 2, 2), dtype=torch.int32, device=dev)
            a.hit_count, a.hit_stripes, a.hit_stripe_count, a.hit_capacity = h_count.data_ptr(), h_stripes.data_ptr(), stripes, hcap
            a.hit_pair, a.hit_fp, a.hit_rec, a.hit_blk, a.unit_ctx = (h_pair.data_ptr(), h_fp.data_ptr(), h_rec.data_ptr(),
                                                                      h_blk.data_ptr(), u_ctx.data_ptr())
        _lib.check(lib.nt_mesh_sdf_collide_reduced(C.byref(a), C.byref(r), stream), "nt_mesh_sdf_collide_reduced")
    else:
        _lib.check(lib.nt_mesh_sdf_collide(C.byref(a), stream), "nt_mesh_sdf_collide")
    torch.cuda.current_stream(dev).synchronize()
    if reduce is not None and staged and int(h_count[0].item()) != 0:
        raise RuntimeError("staged mesh-SDF narrow phase dropped survivors (a stripe of the list ran full)")
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


class MeshSdfContactStage:
    """Mesh-SDF contacts between the mesh-like shapes of a Model, outside the fixed-slot environment tiles: the mesh-mesh leg of
    CollisionPipeline.collide (newton/_src/sim/collide.py:1925-2050 -> NarrowPhase.launch, narrow_phase.py:2588-2760) with
    broad_phase="sap", reduce_contacts=True, followed by the contact writer (collide.py:203-254) into Newton's flat Contacts
    arrays, and the penalty-force consumer SolverSemiImplicit / SolverFeatherstone share (eval_body_contact,
    semi_implicit/kernels_contact.py:381-556) accumulating into State.body_f.

        stage = MeshSdfContactStage(model)            # shapes: every CONVEX_MESH / MESH shape on a body, SDFs built once
        state.clear_forces(); stage.collide(state); stage.apply_forces(state); pipe.collide(state, contacts); solver.step(...)

    Every launch is on the current stream; nothing returns to the host (the broad phase's candidate counter and the narrow
    phase's row counter are read by the next kernel on the device).  The shape pairs handled here must be filtered out of
    Model.shape_contact_pairs (the tiles would run them through MPR / GJK as well)."""

    def __init__(self, model, shape_ids=None, sdf_resolution: int = 24, sdf_margin: float = 0.02, narrow_band_range=(-0.1, 0.1),
                 threads: int = 64, pairs_per_shape: int = 12, contacts_per_shape: int = 40, friction_smoothing: float = 1.0):
        from . import geometry
        from . import sdf as S
        from .enums import GeoType
        from .mesh import mesh_edge_tables

        torch = _torch()
        self.model = model
        dm = model.device_model()
        self.device = dev = dm.device
        self._lib = _lib.load()
        stype, sbody = np.asarray(model.shape_type), np.asarray(model.shape_body)
        if shape_ids is None:
            shape_ids = [s for s in range(model.shape_count)
                         if int(stype[s]) in (int(GeoType.CONVEX_MESH), int(GeoType.MESH)) and sbody[s] >= 0]
        self.shape_ids = ids = np.asarray(shape_ids, dtype=np.int64)
        n = self.n = len(ids)
        if n == 0:
            raise ValueError("MeshSdfContactStage: the model has no mesh shapes on bodies")
        self.threads, self.friction_smoothing = int(threads), float(friction_smoothing)
        # unique mesh assets -> edge tables, SDFs, local AABBs / voxel grids (every environment shares them)
        assets, asset_of = {}, np.zeros(n, dtype=np.int32)
        scale = np.asarray(model.shape_scale, dtype=np.float32)[ids]
        ecs, ehs, ranges, sdfs, verts = [], [], [], [], []
        for k, s in enumerate(ids):
            src = model.shape_source[int(s)]
            key = (id(src), tuple(scale[k].tolist()))
            if key not in assets:
                assets[key] = len(sdfs)
                tri = np.asarray(src.indices).reshape(-1, 3)
                ec, eh = mesh_edge_tables(src.vertices, tri, scale=scale[k])
                ranges.append((sum(len(e) for e in ecs), len(ec)))
                ecs.append(ec)
                ehs.append(eh)
                v = np.asarray(src.vertices, dtype=np.float64) * scale[k]
                verts.append(v)
                sdfs.append(S.create_texture_sdf_from_mesh(v, tri, margin=sdf_margin, narrow_band_range=narrow_band_range,
                                                           max_resolution=sdf_resolution, scale_baked=True,
                                                           quantization_mode=S.QuantizationMode.UINT16))
            asset_of[k] = assets[key]
        self.sdfs = sdfs
        lo, hi, res = S.mesh_reduction_tables(verts, [(1.0, 1.0, 1.0)] * len(verts))
        margin = np.asarray(model.shape_margin, dtype=np.float32)[ids]
        self.shape_gap = np.asarray(model.shape_gap, dtype=np.float32)[ids]
        data = np.concatenate([np.ones((n, 3), np.float32), margin[:, None]], axis=1)  # the scale is baked into edges and SDFs
        self.narrow = MeshSdfNarrowPhase(data, self.shape_gap, asset_of, [DeviceSDF(t, device=dev) for t in sdfs],
                                         np.asarray(ranges, np.int32)[asset_of], np.concatenate(ecs), np.concatenate(ehs),
                                         (lo[asset_of], hi[asset_of], res[asset_of]), device=dev)
        t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, dtype=d)).to(dev)  # noqa: E731
        corners = np.array([[(lo[a][0], hi[a][0])[i], (lo[a][1], hi[a][1])[j], (lo[a][2], hi[a][2])[l]]
                            for a in asset_of for i in (0, 1) for j in (0, 1) for l in (0, 1)], np.float32).reshape(n, 8, 3)
        self._corners = t(corners, np.float32)
        self._body_of = t(sbody[ids], np.int64)
        self._shape_body = t(sbody[ids], np.int32)
        self._local = t(np.asarray(model.shape_transform, np.float32)[ids], np.float32)
        self._local_identity = bool(np.all(np.asarray(model.shape_transform, np.float32)[ids] == np.array([0, 0, 0, 0, 0, 0, 1], np.float32)))
        world = np.asarray(model.shape_world, dtype=np.int32)[ids]
        self._world = t(world, np.int32)
        self._group = t(np.asarray(model.shape_collision_group, np.int32)[ids], np.int32)
        self._gap = t(self.shape_gap, np.float32)
        self._mat = {k: t(np.asarray(getattr(model, "shape_material_" + k), np.float32)[ids], np.float32)
                     for k in ("ke", "kd", "kf", "ka", "mu")}
        self._body_com = t(np.asarray(model.body_com, np.float32), np.float32)
        self._global_id = t(ids, np.int32)
        self.broad = geometry.BroadPhaseSAP(world, None, device=dev)
        self.pair_max, self.contact_max = n * int(pairs_per_shape), n * int(contacts_per_shape)
        i32, f32 = torch.int32, torch.float32
        self.pairs = torch.zeros((self.pair_max, 2), dtype=i32, device=dev)
        self.pair_count = torch.zeros(1, dtype=i32, device=dev)
        self.row_count = torch.zeros(1, dtype=i32, device=dev)
        self._row_pair, self._row_key = torch.zeros(self.contact_max, dtype=i32, device=dev), torch.zeros(self.contact_max, dtype=i32, device=dev)
        self._row_data = torch.zeros((self.contact_max, 9), dtype=f32, device=dev)
        c = self.contact_max
        self.shape0, self.shape1 = torch.full((c,), -1, dtype=i32, device=dev), torch.full((c,), -1, dtype=i32, device=dev)
        self.point0, self.point1, self.offset0, self.offset1, self.normal = (torch.zeros((c, 3), dtype=f32, device=dev) for _ in range(5))
        self.margin0, self.margin1 = torch.zeros(c, dtype=f32, device=dev), torch.zeros(c, dtype=f32, device=dev)
        self._body_q = None

    # -- Newton-shaped views of the stage's rows (stage-local shape ids -> Model shape ids); rows with shape0 == -1 are inert
    @property
    def rigid_contact_count(self):
        return self.row_count

    def rigid_contact_shapes(self):
        n = min(int(self.row_count.item()), self.contact_max)
        g = self._global_id
        s0, s1 = self.shape0[:n].long(), self.shape1[:n].long()
        live = s0 >= 0
        return _torch().where(live, g[s0.clamp(min=0)], -1), _torch().where(live, g[s1.clamp(min=0)], -1)

    def _shape_world_transforms(self, body_q):
        torch = _torch()
        bq = body_q[self._body_of]
        if self._local_identity:
            return bq.contiguous()
        p, q, lp, lq = bq[:, :3], bq[:, 3:], self._local[:, :3], self._local[:, 3:]
        qv, w = q[:, :3], q[:, 3:]
        rp = lp * (2.0 * w * w - 1.0) + torch.cross(qv, lp, dim=-1) * w * 2.0 + qv * (qv * lp).sum(-1, keepdim=True) * 2.0
        ax, ay, az, aw = q.unbind(-1)
        bx, by, bz, bw = lq.unbind(-1)
        qq = torch.stack([aw * bx + bw * ax + ay * bz - by * az, aw * by + bw * ay + az * bx - bz * ax,
                          aw * bz + bw * az + ax * by - bx * ay, aw * bw - ax * bx - ay * by - az * bz], dim=-1)
        return torch.cat([p + rp, qq], dim=-1).contiguous()

    def collide(self, state):
        """AABBs -> per-world sort-and-sweep -> edge-vs-SDF contacts with the global reduction -> flat contact rows."""
        torch = _torch()
        self._body_q = body_q = state.body_q  # [B, 7] AoS copy of the env-major state
        X = self._shape_world_transforms(body_q)
        p, q = X[:, None, :3], X[:, None, 3:]
        qv, w = q[..., :3], q[..., 3:]
        qv, c = torch.broadcast_tensors(qv, self._corners)
        world = c * (2.0 * w * w - 1.0) + torch.cross(qv, c, dim=-1) * w * 2.0 + qv * (qv * c).sum(-1, keepdim=True) * 2.0 + p
        lower, upper = world.amin(dim=1).contiguous(), world.amax(dim=1).contiguous()
        self.broad.launch(lower, upper, self._gap, self._group, self._world, self.n, self.pairs, self.pair_count)
        self.row_count.zero_()
        self.narrow.launch(X, self.pairs, self.pair_count, self.row_count, self._row_pair, self._row_key, self._row_data,
                           reduce=True, threads=self.threads)
        a = _lib.nt_contact_rows()
        a.row_count, a.row_count_device = self.contact_max, self.row_count.data_ptr()
        a.row_pair, a.pairs, a.row_data = self._row_pair.data_ptr(), self.pairs.data_ptr(), self._row_data.data_ptr()
        a.body_q, a.shape_body, a.shape_gap = body_q.data_ptr(), self._shape_body.data_ptr(), self._gap.data_ptr()
        for k in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
            setattr(a, "out_" + k, getattr(self, k).data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib.nt_contact_rows_write(C.byref(a), stream), "nt_contact_rows_write")

    def apply_forces(self, state):
        """eval_body_contact over the stage's rows, added to State.body_f (call between clear_forces and the solver step)."""
        torch = _torch()
        if self._body_q is None:
            raise RuntimeError("MeshSdfContactStage.apply_forces before collide")
        body_qd = state.body_qd
        body_f = torch.zeros((body_qd.shape[0], 6), dtype=torch.float32, device=self.device)
        f = _lib.nt_flat_contact_forces()
        f.body_q, f.body_qd, f.body_com = self._body_q.data_ptr(), body_qd.data_ptr(), self._body_com.data_ptr()
        m = self._mat
        f.shape_ke, f.shape_kd, f.shape_kf, f.shape_ka, f.shape_mu = (m[k].data_ptr() for k in ("ke", "kd", "kf", "ka", "mu"))
        f.shape_body, f.contact_count, f.contact_max = self._shape_body.data_ptr(), self.row_count.data_ptr(), self.contact_max
        for k in ("point0", "point1", "normal", "shape0", "shape1", "margin0", "margin1"):
            setattr(f, k, getattr(self, k).data_ptr())
        f.friction_smoothing, f.body_f = self.friction_smoothing, body_f.data_ptr()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib.nt_eval_body_contact_flat(C.byref(f), stream), "nt_eval_body_contact_flat")
        state.body_f = state.body_f + body_f
        return body_f
