"""The mesh-SDF leg of CollisionPipeline.collide (newton/_src/sim/collide.py:1999 -> geometry/narrow_phase.py:2838-3167 ->
sdf_contact.py:1534-1990): shape pairs whose two shapes carry a texture SDF and collision edges go through the edge-vs-SDF narrow
phase with the global contact reduction, and their contacts are appended after the primitive / GJK-MPR ones.

Device stages (csrc/nt_sdf_pipeline.hip, csrc/nt_sdf.hip), all on the caller's stream, nothing returns to the host:

    nt_collide               exports world transforms + gap-widened AABBs of every shape (the tile kernel's compute_shape_aabbs)
    nt_sdf_candidate_pairs   per world: ordered compaction of the world's SDF pair list against those AABBs + scan over worlds
    nt_mesh_sdf_collide_reduced   edges vs SDF both ways + the 245-slot reduction table per pair, in four dense stages: a lane per
                             pair (edge-independent setup), a wave per runnable pair (culling; survivors into a striped list), a
                             lane per survivor (Brent search), a workgroup per pair with survivors (reduction in LDS)
    nt_hydro_pairs           pairs of two hydroelastic shapes (CollisionPipeline(sdf_hydroelastic_config=...)): SAT, octree, marching
                             cubes; every face as a row, or the reference's hydroelastic contact reduction (Config defaults)
    nt_sdf_rows_finalize     final row ranges (world-major, pairs ascending, fingerprint / rank order), write_contact, per-body row blocks

The rows live in ``FlatRows`` (owned by the Contacts object): Newton's flat contact arrays, deterministic -- two runs give
bit-identical rows -- and consumable by SolverXPBD (inside the step kernel) and SolverSemiImplicit / SolverFeatherstone
(nt_flat_rows_forces: ordered per-body sums, no float atomics)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .enums import GeoType


def _torch():
    import torch  # noqa: PLC0415

    return torch


def model_has_sdf_pairs(model) -> bool:
    return len(getattr(model.env, "sdf_pair", ())) > 0


class FlatRows:
    """Contact rows outside the fixed slots (nt_flat_rows): Newton's AoS arrays + per-world ranges + per-body row blocks."""

    def __init__(self, model, capacity: int, pairs_per_world: int, per_contact_shape_properties: bool = False):
        torch = _torch()
        t = model.env
        dev = model.device_model().device
        i32, f32 = torch.int32, torch.float32
        self.capacity = c = int(capacity)
        self.row_start = torch.zeros(t.env_count + 1, dtype=i32, device=dev)
        self.shape0 = torch.full((c,), -1, dtype=i32, device=dev)
        self.shape1 = torch.full((c,), -1, dtype=i32, device=dev)
        self.point0, self.point1, self.offset0, self.offset1, self.normal = (torch.zeros((c, 3), dtype=f32, device=dev) for _ in range(5))
        self.margin0, self.margin1 = torch.zeros(c, dtype=f32, device=dev), torch.zeros(c, dtype=f32, device=dev)
        self.key = torch.zeros(c, dtype=i32, device=dev)
        self.stiffness = self.damping = self.friction_scale = None
        if per_contact_shape_properties:
            self.stiffness, self.damping, self.friction_scale = (torch.zeros(c, dtype=f32, device=dev) for _ in range(3))
        self.body_blk_start = torch.zeros(t.env_count * (t.nb + 1), dtype=i32, device=dev)
        self.body_blk_list = torch.zeros((t.env_count * 2 * pairs_per_world, 2), dtype=i32, device=dev)
        self.cw = torch.zeros((c, 10), dtype=f32, device=dev)
        # Contacts.force requested (contacts.py:170-226): SolverXPBD accumulates the rows' weighted impulses here
        self.impulse = (torch.zeros((c, 6), dtype=f32, device=dev)
                        if "force" in model.get_requested_contact_attributes() else None)
        self.restitution = None  # [cap][14] scratch of SolverXPBD(enable_restitution=True), allocated on first use

    def restitution_scratch(self):
        if self.restitution is None:
            self.restitution = _torch().zeros((self.capacity, 14), dtype=_torch().float32, device=self.cw.device)
        return self.restitution

    def desc(self) -> _lib.nt_flat_rows:
        d = _lib.nt_flat_rows()
        for k in ("row_start", "shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1",
                  "body_blk_start", "body_blk_list", "cw"):
            setattr(d, k, getattr(self, k).data_ptr())
        if self.impulse is not None:
            d.impulse = self.impulse.data_ptr()
        if self.restitution is not None:
            d.restitution = self.restitution.data_ptr()
        if self.stiffness is not None:
            d.stiffness, d.damping, d.friction_scale = (self.stiffness.data_ptr(), self.damping.data_ptr(),
                                                        self.friction_scale.data_ptr())
        return d


class SdfLeg:
    """Model-level tables + per-pipeline work buffers of the mesh-SDF leg."""

    def __init__(self, model, pairs_per_shape: int = 12, contacts_per_shape: int = 40, threads: int = 64, hydro_config=None,
                 staged: bool = True, survivors_per_row: int = 2, hydro_faces_per_shape: int = 400, hydro_staged: bool = True,
                 hydro_blocks_per_pair: int = 4, triangle_rows_per_pair: int = 245):
        from .sdf_device import DeviceSDF  # noqa: PLC0415

        torch = _torch()
        self.model = model
        dm = self.dm = model.device_model()
        self.lib = dm.lib
        dev = self.device = dm.device
        t = self.t = model.env
        S = model.shape_count
        self.threads = int(threads)

        def up(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            return torch.from_numpy(a if a.size else np.zeros(1, dtype=dtype)).to(dev)

        # shapes of one world that take part (for the capacities)
        used = np.unique(t.sdf_pair.reshape(-1))
        n_shapes = max(int((used < t.ns).sum()), 1)
        self.pairs_per_world = int(min(len(t.sdf_pair), max(16, n_shapes * int(pairs_per_shape))))
        self.rows_per_world = int(max(64, n_shapes * int(contacts_per_shape)))
        E, PPW = t.env_count, self.pairs_per_world
        self.row_capacity = E * self.rows_per_world
        # Newton-id tables
        scale = np.asarray(model.shape_scale, np.float32)
        self._shape_data = up(np.concatenate([scale, np.asarray(model.shape_margin, np.float32)[:, None]], axis=1), np.float32)
        self._shape_gap = up(model.shape_gap, np.float32)
        self._sdf_index = up(model._shape_sdf_index, np.int32)
        self._edge_range = up(model.shape_edge_range, np.int32)
        self._edge_centers, self._edge_halves = up(model.mesh_edge_centers, np.float32), up(model.mesh_edge_halves, np.float32)
        self._sdfs = [DeviceSDF(s, device=dev) for s in model._texture_sdf_data]
        table = (_lib.nt_sdf * max(len(self._sdfs), 1))()
        for k, s in enumerate(self._sdfs):
            table[k] = s.desc
        self._sdf_table = torch.from_numpy(np.frombuffer(bytes(table), dtype=np.uint8).copy()).to(dev)
        self._red_lo, self._red_hi = up(model.shape_collision_aabb_lower, np.float32), up(model.shape_collision_aabb_upper, np.float32)
        self._red_res = up(model._shape_voxel_resolution, np.int32)
        er, ecr = np.asarray(model.shape_edge_range, np.int64).reshape(-1, 2), np.asarray(model.mesh_edge_centers, np.float32).reshape(-1, 4)
        self._edge_rmax = up([float(ecr[a:a + n, 3].max()) if n > 0 else 0.0 for a, n in er], np.float32)
        mat = np.stack([np.asarray(getattr(model, "shape_material_" + k), np.float32) for k in ("ke", "kd", "kf", "ka", "mu")], axis=1)
        self._material = up(mat, np.float32)
        # scene (template) tables
        self._template_pair = up(t.sdf_pair, np.int32)
        self._gshape_id = up(t.gshape_id, np.int32)
        self._shape_body = up(t.shape_body[: t.ns], np.int32)
        sc = _lib.nt_sdf_scene()
        sc.env_count, sc.env_stride, sc.nb, sc.ns, sc.shape_local0 = E, t.env_stride, t.nb, t.ns, t.shape_local0
        sc.template_pairs, sc.template_pair = len(t.sdf_pair), self._template_pair.data_ptr()
        sc.gshape_id, sc.shape_body, sc.shape_gap = self._gshape_id.data_ptr(), self._shape_body.data_ptr(), self._shape_gap.data_ptr()
        sc.pairs_per_world = PPW
        # pair kinds: with a hydroelastic configuration the pairs of two HYDROELASTIC shapes take the SDF-SDF leg (kind 1)
        self.hydro = hydro_config
        kind = (t.sdf_pair_hydro if hydro_config is not None else np.zeros(len(t.sdf_pair), bool)).astype(np.uint8)
        # (triangle mesh, infinite plane): the vertex leg (kind 2, NT_PAIR_KIND_MESH_PLANE; include/newton_hip_mesh.h)
        mesh_plane = np.asarray(getattr(t, "sdf_pair_mesh_plane", np.zeros(len(t.sdf_pair), bool)), bool)
        kind[mesh_plane] = 2
        # (triangle mesh, convex primitive): the triangle leg (kind 3, NT_PAIR_KIND_MESH_TRIANGLE; include/newton_hip_mesh.h)
        mesh_tri = np.asarray(getattr(t, "sdf_pair_mesh_tri", np.zeros(len(t.sdf_pair), bool)), bool)
        kind[mesh_tri] = 3
        if np.any((kind == 0) & ~t.sdf_pair_has_edges):
            raise NotImplementedError("pairs of hydroelastic shapes without collision edges need CollisionPipeline(sdf_hydroelastic_config=...)")
        self.has_hydro_pairs = bool((kind == 1).any())
        self.has_mesh_plane_pairs = bool((kind == 2).any())
        self.has_mesh_tri_pairs = bool((kind == 3).any())
        self.has_edge_pairs = bool((kind == 0).any())
        self.mesh_plane_reduce = True  # CollisionPipeline(reduce_contacts=...): set by the pipeline
        self.edge_reduce = True  # ... and the edge leg: False = every contact the edge search admits (keep_all), staged variant
        self.hydro_reduce, self.face_capacity, self.hydro_staged = 0, 0, False
        self._template_kind = up(kind, np.uint8)
        self.world_pair_kind = torch.zeros(E * PPW, dtype=torch.uint8, device=dev)
        if self.has_mesh_plane_pairs:
            sc.template_kind, sc.world_pair_kind = self._template_kind.data_ptr(), self.world_pair_kind.data_ptr()
            self._shape_type = up(model.shape_type, np.int32)
            self._vertex_range = up(model.mesh_vertex_range, np.int32)
            self._vertices = up(model.mesh_vertices, np.float32)
            if int(np.asarray(model.mesh_vertex_range)[:, 1].max()) >= (1 << 22):
                # the reduction's packed values carry the vertex index in 22 bits (contact_reduction_global.py: FINGERPRINT bits)
                raise NotImplementedError("triangle meshes with 2^22 or more vertices are not supported by the vertex leg")
            # rows a world can get from its vertex pairs: every vertex when the reduction is off, else the reduction's table
            mp_rows = sum(int(model.mesh_vertex_range[t.shape_local0 + l if l < t.ns else int(t.gshape_id[l - t.ns]), 1])
                          for pr in t.sdf_pair[mesh_plane] for l in pr)
            # the vertex rows come ON TOP of the other legs' rows: a world with both kinds of pairs shares one row budget
            other_rows = self.rows_per_world if (self.has_edge_pairs or self.has_hydro_pairs) else 0
            self.rows_per_world = max(self.rows_per_world, other_rows + mp_rows + 64)
            self.row_capacity = E * self.rows_per_world
        if self.has_mesh_tri_pairs:
            sc.template_kind, sc.world_pair_kind = self._template_kind.data_ptr(), self.world_pair_kind.data_ptr()
            self._shape_type = up(model.shape_type, np.int32)
            self._vertex_range = up(model.mesh_vertex_range, np.int32)
            self._vertices = up(model.mesh_vertices, np.float32)
            self._triangle_range = up(model.mesh_triangle_range, np.int32)
            self._indices = up(model.mesh_indices, np.int32)
            # bounds of every 64 consecutive triangles, one table per distinct mesh (shapes sharing a mesh share its blocks)
            from .mesh import triangle_block_bounds  # noqa: PLC0415

            vr, tr = np.asarray(model.mesh_vertex_range).reshape(-1, 2), np.asarray(model.mesh_triangle_range).reshape(-1, 2)
            blk_start, blk_of, tables, n_blk = np.zeros(S, np.int32), {}, [], 0
            for i in range(S):
                if tr[i, 1] <= 0:
                    continue
                key = (int(vr[i, 0]), int(tr[i, 0]), int(tr[i, 1]))
                if key not in blk_of:
                    blk_of[key] = n_blk
                    tables.append(triangle_block_bounds(np.asarray(model.mesh_vertices)[vr[i, 0]:vr[i, 0] + vr[i, 1]],
                                                        np.asarray(model.mesh_indices)[tr[i, 0]:tr[i, 0] + tr[i, 1]]))
                    n_blk += len(tables[-1])
                blk_start[i] = blk_of[key]
            self._block_bounds = up(np.concatenate(tables) if tables else np.zeros((1, 6), np.float32), np.float32)
            self._block_start = up(blk_start, np.int32)
            # CONVEX_MESH partners: the hulls' vertex tables (Model.mesh_points: every distinct hull once, deduplicated vertices)
            self._hull_range = up(np.stack([np.asarray(model.shape_mesh_start, np.int32), np.asarray(model.shape_mesh_count, np.int32)], axis=1), np.int32)
            self._hull_points = up(model.mesh_points, np.float32)
            ntri_max = int(np.asarray(model.mesh_triangle_range)[:, 1].max())
            if ntri_max >= (1 << 18):
                # the reduction's packed values carry the fingerprint (triangle << 4 | 8 | manifold index) in 22 bits
                raise NotImplementedError("triangle meshes with 2^18 or more triangles are not supported by the triangle leg")

            # heightfields as the mesh-like shape of a pair: HeightfieldData records + elevation grids (include/newton_hip_mesh.h)
            self._hf_index = self._hf_data = self._hf_elev = None
            hf_tris = np.zeros(S, np.int64)
            if int(getattr(model, "heightfield_count", 0)) > 0:
                hf = (_lib.nt_heightfield * model.heightfield_count)()
                for k, (off, nrow, ncol, hx, hy, zlo, zhi) in enumerate(model.heightfield_data):
                    hf[k] = _lib.nt_heightfield(int(off), int(nrow), int(ncol), float(hx), float(hy), float(zlo), float(zhi))
                self._hf_data = torch.from_numpy(np.frombuffer(bytes(hf), dtype=np.uint8).copy()).to(dev)
                self._hf_index = up(model.shape_heightfield_index, np.int32)
                self._hf_elev = up(model.heightfield_elevations, np.float32)
                for i, h in enumerate(np.asarray(model.shape_heightfield_index)):
                    if h >= 0:
                        hf_tris[i] = 2 * (model.heightfield_data[h][1] - 1) * (model.heightfield_data[h][2] - 1)

            def tri_count(l):
                i = t.shape_local0 + l if l < t.ns else int(t.gshape_id[l - t.ns])
                return int(model.mesh_triangle_range[i, 1]) + int(hf_tris[i])

            # rows a world can get from its triangle pairs: the reduction's table per pair (245 slots), bounded by 5 contacts per
            # triangle; CollisionPipeline(reduce_contacts=False) passes triangle_rows_per_pair for the unreduced budget
            per_pair = [min(5 * max(tri_count(int(a)), tri_count(int(b))), int(triangle_rows_per_pair)) for a, b in t.sdf_pair[mesh_tri]]
            other_rows = self.rows_per_world if (self.has_edge_pairs or self.has_hydro_pairs or self.has_mesh_plane_pairs) else 0
            self.rows_per_world = max(self.rows_per_world, other_rows + sum(per_pair) + 64)
            self.row_capacity = E * self.rows_per_world
        if self.has_hydro_pairs:
            from .mc_tables import tables  # noqa: PLC0415

            self.hydro_reduce = (1 | (2 if hydro_config.pre_prune_contacts else 0) | (4 if hydro_config.normal_matching else 0) |
                                 (8 if hydro_config.anchor_contact or hydro_config.moment_matching else 0) |
                                 (16 if hydro_config.moment_matching else 0) if hydro_config.reduce_contacts else 0)
            self.face_capacity = E * n_shapes * int(hydro_faces_per_shape) if self.hydro_reduce else 0
            if hydro_config.pressure_func is not None:
                raise NotImplementedError("custom pressure_func callbacks are not supported (linear pressure -kh * depth only)")
            sc.template_kind, sc.world_pair_kind = self._template_kind.data_ptr(), self.world_pair_kind.data_ptr()
            tri_range, flat = tables()
            self._mc_tri, self._mc_edges = up(tri_range, np.int32), up(flat, np.uint8)
            self._shape_kh = up(model.shape_material_kh, np.float32)
            nblocks = max(int(np.prod(s.slots.shape)) for s in model._texture_sdf_data)
            if nblocks > 4096:
                raise NotImplementedError(f"hydroelastic SDFs with more than 4096 subgrid blocks ({nblocks}) are not supported")
        self.scene = sc
        # work buffers
        i32, f32 = torch.int32, torch.float32
        self.world_xform = torch.zeros((S, 7), dtype=f32, device=dev)
        self.aabb_lower, self.aabb_upper = torch.zeros((S, 3), dtype=f32, device=dev), torch.zeros((S, 3), dtype=f32, device=dev)
        self.world_pairs = torch.zeros((E * PPW, 2), dtype=i32, device=dev)
        self.pair_count = torch.zeros(E, dtype=i32, device=dev)
        self.pair_prefix = torch.zeros(E + 1, dtype=i32, device=dev)
        self.blk = torch.zeros((E * PPW, 2), dtype=i32, device=dev)
        self.pair_row = torch.zeros(E * PPW, dtype=i32, device=dev)
        self.world_rows = torch.zeros(E, dtype=i32, device=dev)
        self.raw_count = torch.zeros(1, dtype=i32, device=dev)
        # staged narrow phase (units -> cull -> resolve -> reduce): the list of culling survivors over all pairs, in stripes; the
        # reduced rows of a pair are written from the start of its survivor block, so the raw row arrays span the list (the
        # hydroelastic leg appends its rows behind it through the counter)
        self.staged = bool(staged) and os.environ.get("NT_SDF_STAGED", "1") != "0"
        self.hit_stripe_count = 1024
        self.hit_capacity = 0
        if self.staged:
            self.hit_capacity = -(-self.row_capacity * int(survivors_per_row) // self.hit_stripe_count) * self.hit_stripe_count
            self.hit_count = torch.zeros(4, dtype=i32, device=dev)
            self.hit_stripes = torch.zeros(self.hit_stripe_count * 16, dtype=i32, device=dev)
            self.unit_ctx = torch.zeros((E * PPW, 2, 24), dtype=f32, device=dev)
            self.hit_pair = torch.zeros(self.hit_capacity, dtype=i32, device=dev)
            self.hit_fp = torch.zeros(self.hit_capacity, dtype=i32, device=dev)
            self.hit_rec = torch.zeros((self.hit_capacity, 8), dtype=f32, device=dev)
            self.hit_blk = torch.zeros((E * PPW, 2, 2), dtype=i32, device=dev)
        self.raw_capacity = self.hit_capacity + self.row_capacity
        self.raw_pair = torch.zeros(self.raw_capacity, dtype=i32, device=dev)
        self.raw_key = torch.zeros(self.raw_capacity, dtype=i32, device=dev)
        self.raw_data = torch.zeros((self.raw_capacity, 9), dtype=f32, device=dev)
        self.raw_radius = torch.zeros((self.raw_capacity, 2), dtype=f32, device=dev) if self.has_mesh_tri_pairs else None
        self.raw_rank = torch.zeros(self.raw_capacity if self.has_hydro_pairs else 1, dtype=i32, device=dev)
        self.raw_stiffness = torch.zeros(self.raw_capacity if self.has_hydro_pairs else 1, dtype=f32, device=dev)
        if self.has_hydro_pairs and self.hydro_reduce:  # the reduction's face buffer (scratch of a call) + per-row friction scale
            self.face_count = torch.zeros(2, dtype=i32, device=dev)
            self.face_rec = torch.zeros((self.face_capacity, 12), dtype=f32, device=dev)
            self.raw_friction = torch.zeros(self.raw_capacity, dtype=f32, device=dev)
            # dense stages of the reduced pipeline (nt_hydro_args.stage_*): (pair, block) queue, its per-item / per-64-voxel records
            self.hydro_staged = bool(hydro_staged)
            if self.hydro_staged:
                self.stage_queue_capacity = max(E * PPW * int(hydro_blocks_per_pair), 1024)
                self.stage_chunk_capacity = 2 * self.stage_queue_capacity
                self.stage_count = torch.zeros(8, dtype=i32, device=dev)
                self.stage_active = torch.zeros(E * PPW, dtype=i32, device=dev)
                self.stage_queue = torch.zeros((self.stage_queue_capacity, 2), dtype=i32, device=dev)
                self.stage_item = torch.zeros((self.stage_queue_capacity, 2), dtype=i32, device=dev)
                self.stage_pair = torch.zeros((E * PPW, 2), dtype=i32, device=dev)
                self.stage_chunk = torch.zeros((self.stage_chunk_capacity, 4), dtype=i32, device=dev)

    def new_rows(self, per_contact_shape_properties: bool = False) -> FlatRows:
        # hydroelastic rows carry Contacts.rigid_contact_stiffness: allocated whenever the leg can produce them
        return FlatRows(self.model, self.row_capacity, self.pairs_per_world, per_contact_shape_properties or self.has_hydro_pairs)

    def export_pointers(self, d: _lib.nt_contacts) -> None:
        """Ask nt_collide for the shapes' world transforms and AABBs."""
        d.world_xform, d.world_aabb_lower, d.world_aabb_upper = (self.world_xform.data_ptr(), self.aabb_lower.data_ptr(),
                                                                 self.aabb_upper.data_ptr())

    def collide(self, state, rows: FlatRows, stream) -> None:
        """Candidate pairs -> narrow phase + reduction -> final rows.  nt_collide has run on `state` (world_xform / AABBs)."""
        lib, sc = self.lib, self.scene
        _lib.check(lib.nt_sdf_candidate_pairs(C.byref(sc), self.aabb_lower.data_ptr(), self.aabb_upper.data_ptr(),
                                              self.world_pairs.data_ptr(), self.pair_count.data_ptr(), self.pair_prefix.data_ptr(),
                                              stream), "nt_sdf_candidate_pairs")
        self.raw_count.fill_(self.hit_capacity)  # rows appended through the counter (single kernel, hydroelastic leg) start here
        a = _lib.nt_mesh_sdf_args()
        a.pairs, a.pair_count = self.world_pairs.data_ptr(), int(self.world_pairs.shape[0])
        a.shape_transform, a.shape_data, a.shape_gap = self.world_xform.data_ptr(), self._shape_data.data_ptr(), self._shape_gap.data_ptr()
        a.shape_sdf_index, a.sdf_table, a.sdf_count = self._sdf_index.data_ptr(), self._sdf_table.data_ptr(), len(self._sdfs)
        a.shape_edge_range, a.edge_centers, a.edge_halves = (self._edge_range.data_ptr(), self._edge_centers.data_ptr(),
                                                             self._edge_halves.data_ptr())
        a.out_count, a.out_pair, a.out_key, a.out_data = (self.raw_count.data_ptr(), self.raw_pair.data_ptr(),
                                                          self.raw_key.data_ptr(), self.raw_data.data_ptr())
        a.capacity = self.raw_capacity
        a.pair_world_prefix, a.worlds, a.pairs_per_world, a.out_blk = (self.pair_prefix.data_ptr(), sc.env_count,
                                                                       sc.pairs_per_world, self.blk.data_ptr())
        r = _lib.nt_contact_reduce_shapes()
        r.shape_aabb_lower, r.shape_aabb_upper, r.shape_voxel_res = self._red_lo.data_ptr(), self._red_hi.data_ptr(), self._red_res.data_ptr()
        r.threads, r.shape_edge_radius_max = self.threads, self._edge_rmax.data_ptr()
        if not self.edge_reduce:
            if not self.staged:
                raise NotImplementedError("reduce_contacts=False with mesh-SDF pairs runs the staged narrow phase (NT_SDF_STAGED=0 is set)")
            r.keep_all = 1
        if self.has_hydro_pairs or self.has_mesh_plane_pairs or self.has_mesh_tri_pairs:
            a.pair_kind = self.world_pair_kind.data_ptr()
        if self.staged:
            a.hit_count, a.hit_stripes, a.hit_stripe_count, a.hit_capacity = (
                self.hit_count.data_ptr(), self.hit_stripes.data_ptr(), self.hit_stripe_count, self.hit_capacity)
            a.hit_pair, a.hit_fp, a.hit_rec, a.hit_blk, a.unit_ctx = (
                self.hit_pair.data_ptr(), self.hit_fp.data_ptr(), self.hit_rec.data_ptr(), self.hit_blk.data_ptr(),
                self.unit_ctx.data_ptr())
        if self.has_edge_pairs:
            _lib.check(lib.nt_mesh_sdf_collide_reduced(C.byref(a), C.byref(r), stream), "nt_mesh_sdf_collide_reduced")
        if self.has_mesh_plane_pairs:  # (triangle mesh, infinite plane): rows appended through the counter, one block per pair
            mp = _lib.nt_mesh_plane_args()
            mp.pairs, mp.pair_world_prefix, mp.worlds, mp.pairs_per_world = (self.world_pairs.data_ptr(), self.pair_prefix.data_ptr(),
                                                                           sc.env_count, sc.pairs_per_world)
            mp.pair_kind, mp.shape_type = self.world_pair_kind.data_ptr(), self._shape_type.data_ptr()
            mp.shape_transform, mp.shape_data, mp.shape_gap = (self.world_xform.data_ptr(), self._shape_data.data_ptr(),
                                                               self._shape_gap.data_ptr())
            mp.shape_vertex_range, mp.vertices = self._vertex_range.data_ptr(), self._vertices.data_ptr()
            mp.shape_aabb_lower, mp.shape_aabb_upper, mp.shape_voxel_res = (self._red_lo.data_ptr(), self._red_hi.data_ptr(),
                                                                            self._red_res.data_ptr())
            mp.reduce = int(bool(self.mesh_plane_reduce))
            mp.out_count, mp.out_pair, mp.out_key, mp.out_data, mp.capacity = (self.raw_count.data_ptr(), self.raw_pair.data_ptr(),
                                                                              self.raw_key.data_ptr(), self.raw_data.data_ptr(),
                                                                              self.raw_capacity)
            mp.out_blk = self.blk.data_ptr()
            _lib.check(lib.nt_mesh_plane_pairs(C.byref(mp), stream), "nt_mesh_plane_pairs")
        if self.has_mesh_tri_pairs:  # (triangle mesh, convex primitive): rows appended through the counter, one block per pair
            mt = _lib.nt_mesh_triangle_args()
            mt.pairs, mt.pair_world_prefix, mt.worlds, mt.pairs_per_world = (self.world_pairs.data_ptr(), self.pair_prefix.data_ptr(),
                                                                           sc.env_count, sc.pairs_per_world)
            mt.pair_kind, mt.shape_type = self.world_pair_kind.data_ptr(), self._shape_type.data_ptr()
            mt.shape_transform, mt.shape_data, mt.shape_gap = (self.world_xform.data_ptr(), self._shape_data.data_ptr(),
                                                               self._shape_gap.data_ptr())
            mt.shape_vertex_range, mt.vertices = self._vertex_range.data_ptr(), self._vertices.data_ptr()
            mt.shape_triangle_range, mt.indices = self._triangle_range.data_ptr(), self._indices.data_ptr()
            mt.shape_aabb_lower, mt.shape_aabb_upper, mt.shape_voxel_res = (self._red_lo.data_ptr(), self._red_hi.data_ptr(),
                                                                            self._red_res.data_ptr())
            mt.reduce = int(bool(self.mesh_plane_reduce))
            mt.out_count, mt.out_pair, mt.out_key, mt.out_data, mt.capacity = (self.raw_count.data_ptr(), self.raw_pair.data_ptr(),
                                                                              self.raw_key.data_ptr(), self.raw_data.data_ptr(),
                                                                              self.raw_capacity)
            mt.out_radius, mt.out_blk = self.raw_radius.data_ptr(), self.blk.data_ptr()
            mt.hull_points, mt.shape_hull_range = self._hull_points.data_ptr(), self._hull_range.data_ptr()
            if self._hf_index is not None:
                mt.shape_heightfield_index, mt.heightfields, mt.elevations = (self._hf_index.data_ptr(), self._hf_data.data_ptr(),
                                                                              self._hf_elev.data_ptr())
            if os.environ.get("NT_TRIANGLE_BLOCKS", "1") != "0":  # (0: the plain scan over every triangle -- measurements)
                mt.block_bounds, mt.shape_block_start = self._block_bounds.data_ptr(), self._block_start.data_ptr()
            _lib.check(lib.nt_mesh_triangle_pairs(C.byref(mt), stream), "nt_mesh_triangle_pairs")
        if self.has_hydro_pairs:
            h = _lib.nt_hydro_args()
            h.pairs, h.pair_count = self.world_pairs.data_ptr(), int(self.world_pairs.shape[0])
            h.shape_transform, h.shape_data, h.shape_gap, h.shape_kh = (self.world_xform.data_ptr(), self._shape_data.data_ptr(),
                                                                        self._shape_gap.data_ptr(), self._shape_kh.data_ptr())
            h.shape_sdf_index, h.sdf_table, h.sdf_count = self._sdf_index.data_ptr(), self._sdf_table.data_ptr(), len(self._sdfs)
            h.tri_range, h.flat_edge_verts = self._mc_tri.data_ptr(), self._mc_edges.data_ptr()
            h.margin_contact_area, h.edge_clamp_min = float(self.hydro.margin_contact_area), float(self.hydro.mc_edge_clamp_min)
            h.out_count, h.out_pair, h.out_key, h.out_data, h.capacity = (self.raw_count.data_ptr(), self.raw_pair.data_ptr(),
                                                                          self.raw_key.data_ptr(), self.raw_data.data_ptr(),
                                                                          self.raw_capacity)
            h.pair_world_prefix, h.worlds, h.pairs_per_world = self.pair_prefix.data_ptr(), sc.env_count, sc.pairs_per_world
            h.pair_kind, h.out_pairs_normalized, h.out_blk = (self.world_pair_kind.data_ptr(), self.world_pairs.data_ptr(),
                                                              self.blk.data_ptr())
            h.out_rank, h.out_stiffness = self.raw_rank.data_ptr(), self.raw_stiffness.data_ptr()
            if self.hydro_reduce:
                self.face_count.zero_()
                h.reduce = self.hydro_reduce
                h.shape_aabb_lower, h.shape_aabb_upper, h.shape_voxel_res = (self._red_lo.data_ptr(), self._red_hi.data_ptr(),
                                                                              self._red_res.data_ptr())
                h.face_count, h.face_rec, h.face_capacity = self.face_count.data_ptr(), self.face_rec.data_ptr(), self.face_capacity
                h.out_friction = self.raw_friction.data_ptr()
                if self.hydro_staged:
                    h.stage_count, h.stage_queue, h.stage_queue_capacity = (self.stage_count.data_ptr(), self.stage_queue.data_ptr(),
                                                                            self.stage_queue_capacity)
                    h.stage_pair, h.stage_item = self.stage_pair.data_ptr(), self.stage_item.data_ptr()
                    h.stage_active = self.stage_active.data_ptr()
                    h.stage_chunk, h.stage_chunk_capacity = self.stage_chunk.data_ptr(), self.stage_chunk_capacity
            _lib.check(lib.nt_hydro_pairs(C.byref(h), stream), "nt_hydro_pairs")
        io = _lib.nt_sdf_rows_io()
        io.pair_count, io.world_pairs, io.blk, io.pair_row = (self.pair_count.data_ptr(), self.world_pairs.data_ptr(),
                                                              self.blk.data_ptr(), self.pair_row.data_ptr())
        io.row_start, io.raw_count, io.raw_pair, io.raw_key, io.raw_data = (rows.row_start.data_ptr(), self.raw_count.data_ptr(),
                                                                            self.raw_pair.data_ptr(), self.raw_key.data_ptr(),
                                                                            self.raw_data.data_ptr())
        io.raw_capacity, io.row_capacity = self.raw_capacity, rows.capacity
        io.raw_base = self.hit_capacity
        for k in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1", "key"):
            setattr(io, k, getattr(rows, k).data_ptr())
        if self.raw_radius is not None:
            io.raw_radius = self.raw_radius.data_ptr()
        if self.has_hydro_pairs or self.has_mesh_plane_pairs or self.has_mesh_tri_pairs:  # (pair kinds in play: the writer asks for the rank array)
            io.raw_rank, io.raw_stiffness = self.raw_rank.data_ptr(), self.raw_stiffness.data_ptr()
            if self.hydro_reduce:
                io.raw_friction = self.raw_friction.data_ptr()
        if rows.stiffness is not None:
            io.stiffness, io.damping, io.friction_scale = rows.stiffness.data_ptr(), rows.damping.data_ptr(), rows.friction_scale.data_ptr()
        _lib.check(lib.nt_sdf_rows_finalize(C.byref(sc), C.byref(io), state._soa["body_q"].data_ptr(), self.world_rows.data_ptr(),
                                            rows.body_blk_start.data_ptr(), rows.body_blk_list.data_ptr(), stream),
                   "nt_sdf_rows_finalize")
        self._io, self._io_rows = io, rows  # the pair tables + row arrays of this frame, for FlatRowMatcher

    def overflow(self, rows: FlatRows) -> dict:
        """Host check (tests / benches, synchronises): did any world exceed its candidate capacity, or the rows their buffer?"""
        pc = int(self.pair_count.max().item()) if self.pair_count.numel() else 0
        appended = int(self.raw_count.item()) - self.hit_capacity  # rows that came through the counter
        total = int(self.world_rows.sum().item())  # unclamped: row_start itself never leaves the row arrays
        info = {"pairs_per_world_max": pc, "pairs_per_world_capacity": self.pairs_per_world, "appended_rows": appended,
                "rows": total, "row_capacity": rows.capacity, "candidate_pairs": int(self.pair_prefix[-1].item()),
                "pairs_with_rows": int((self.blk[:, 1] > 0).sum().item())}
        over = pc > self.pairs_per_world or appended > self.row_capacity or total > rows.capacity
        if self.has_hydro_pairs and self.hydro_reduce:
            info["hydro_faces"], info["hydro_face_capacity"] = int(self.face_count[0].item()), self.face_capacity
            info["hydro_pairs_truncated"] = int(self.face_count[1].item())
            over = over or info["hydro_faces"] > self.face_capacity or info["hydro_pairs_truncated"] > 0
            if self.hydro_staged:
                sc = self.stage_count.cpu().numpy()
                info["hydro_blocks"], info["hydro_block_capacity"] = int(sc[0]), self.stage_queue_capacity
                info["hydro_chunks"], info["hydro_chunk_capacity"] = int(sc[1]), self.stage_chunk_capacity
                info["hydro_units_dropped"] = int(sc[2])
                over = over or int(sc[2]) > 0
        if self.staged:
            fill = self.hit_stripes[::16]
            info["cull_survivors"], info["survivor_capacity"] = int(fill.sum().item()), self.hit_capacity
            info["stripe_fill_max"] = int(fill.max().item())
            info["stripe_capacity"] = self.hit_capacity // max(int(self.hit_count[2].item()), 1)
            info["dropped_survivors"] = int(self.hit_count[0].item())
            over = over or info["dropped_survivors"] > 0
        info["overflow"] = bool(over)
        return info

    def add_forces(self, state, rows: FlatRows, body_f, friction_smoothing: float, stream) -> None:
        """eval_body_contact over the rows, ordered per-body sums ADDED to `body_f` (env-major [6][nb][ES])."""
        p = _lib.nt_flat_force_params()
        p.body_q, p.body_qd = state._soa["body_q"].data_ptr(), state._soa["body_qd"].data_ptr()
        p.body_com = self.dm.desc.body_param  # (the descriptor's own table: rows BP_COM.. of body_param)
        p.shape_material, p.friction_smoothing, p.body_f = self._material.data_ptr(), float(friction_smoothing), body_f.data_ptr()
        d = rows.desc()
        _lib.check(self.lib.nt_flat_rows_forces(C.byref(self.scene), C.byref(d), C.byref(p), stream), "nt_flat_rows_forces")


class FlatRowMatcher:
    """Frame-to-frame matching of the SDF leg's rows (newton/_src/geometry/contact_match.py:602-1055 on the part of the contact
    list the mesh-SDF kernels produce): nt_flat_rows_match / _replay_matched / _save_history on the (world, pair) row blocks.
    Results are ROW indices of the previous frame; collide.ContactMatcher maps them into the exported contact order."""

    def __init__(self, leg: "SdfLeg", sticky: bool = False):
        torch = _torch()
        self.leg = leg
        dev, E, PPW, cap = leg.device, leg.t.env_count, leg.pairs_per_world, leg.row_capacity
        i32, f32 = torch.int32, torch.float32
        self.prev_row_start = torch.zeros(E + 1, dtype=i32, device=dev)
        self.prev_pair_count = torch.zeros(E, dtype=i32, device=dev)
        self.prev_world_pairs = torch.zeros((E * PPW, 2), dtype=i32, device=dev)
        self.prev_pair_row = torch.zeros(E * PPW, dtype=i32, device=dev)
        self.prev_pair_rows = torch.zeros(E * PPW, dtype=i32, device=dev)
        self.prev_live = torch.zeros(cap, dtype=torch.uint8, device=dev)
        self.prev_pos_world = torch.zeros((cap, 3), dtype=f32, device=dev)
        self.prev_normal = torch.zeros((cap, 3), dtype=f32, device=dev)
        self.prev_claim = torch.full((cap,), -1, dtype=torch.int64, device=dev)
        self.prev_body_frame = torch.zeros((cap, 12), dtype=f32, device=dev) if sticky else None
        self.match_index = torch.full((cap,), -1, dtype=i32, device=dev)
        self.sticky = bool(sticky)
        h = _lib.nt_flat_history()
        for k in ("prev_row_start", "prev_pair_count", "prev_world_pairs", "prev_pair_row", "prev_pair_rows", "prev_live",
                  "prev_pos_world", "prev_normal", "prev_claim"):
            setattr(h, k, getattr(self, k).data_ptr())
        if sticky:
            h.prev_body_frame = self.prev_body_frame.data_ptr()
        self._h = h

    def reset(self, world_mask=None) -> None:
        """Forget the history of the selected worlds (all when None): a world without previous pairs matches nothing."""
        if world_mask is None:
            self.prev_pair_count.zero_()
        else:
            torch = _torch()
            m = torch.as_tensor(world_mask, device=self.leg.device).bool()[: self.leg.t.env_count]
            self.prev_pair_count[m] = 0

    def _args(self, state, rows):
        leg = self.leg
        if getattr(leg, "_io_rows", None) is not rows:
            raise ValueError("FlatRowMatcher: collide() has not run on these Contacts")
        return C.byref(leg.scene), C.byref(leg._io), state._soa["body_q"].data_ptr(), C.byref(self._h)

    def match(self, state, rows, pos_threshold: float, normal_dot_threshold: float):
        """-> int32 [row_capacity]: previous ROW index, -1 (not found / inert row) or -2 (broken) for every row of this frame."""
        sc, io, q, h = self._args(state, rows)
        _lib.check(self.leg.lib.nt_flat_rows_match(sc, io, q, h, float(pos_threshold), float(normal_dot_threshold),
                                                   self.match_index.data_ptr(), self.leg.dm.stream()), "nt_flat_rows_match")
        return self.match_index

    def replay_matched(self, state, rows) -> None:
        sc, io, q, h = self._args(state, rows)
        _lib.check(self.leg.lib.nt_flat_rows_replay_matched(sc, io, q, h, self.match_index.data_ptr(), self.leg.dm.stream()),
                   "nt_flat_rows_replay_matched")

    def save_history(self, state, rows) -> None:
        sc, io, q, h = self._args(state, rows)
        _lib.check(self.leg.lib.nt_flat_rows_save_history(sc, io, q, h, self.leg.dm.stream()), "nt_flat_rows_save_history")

    def previous_rows_alive(self):
        """bool [row_capacity]: rows of the previous frame that were contacts and whose world still has its history."""
        torch = _torch()
        cap = self.prev_live.numel()
        world = torch.bucketize(torch.arange(cap, device=self.prev_live.device), self.prev_row_start[1:].to(torch.int64), right=True)
        E = self.leg.t.env_count
        has = torch.zeros(E + 1, dtype=torch.bool, device=self.prev_live.device)
        has[:E] = self.prev_pair_count > 0
        return (self.prev_live != 0) & has[world.clamp(max=E)]


def sdf_pair_shape_types_ok(model) -> None:
    """The SDF leg handles MESH / CONVEX_MESH / BOX shapes (texture SDF + collision edges); heightfields are out of scope."""
    t = model.env
    mesh_plane = getattr(t, "sdf_pair_mesh_plane", np.zeros(len(t.sdf_pair), bool))
    mesh_tri = getattr(t, "sdf_pair_mesh_tri", np.zeros(len(t.sdf_pair), bool))
    for (a, b), hydro, mp in zip(t.sdf_pair, t.sdf_pair_hydro, mesh_plane | mesh_tri):
        if mp:
            continue
        for s in (a, b):
            if not hydro and int(t.shape_type[s]) not in (int(GeoType.MESH), int(GeoType.CONVEX_MESH), int(GeoType.BOX)):
                raise NotImplementedError(f"SDF contact pairs with shape type {GeoType(int(t.shape_type[s])).name} are not supported")
