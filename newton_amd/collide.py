"""Contacts + CollisionPipeline (newton/_src/sim/contacts.py:118-420, newton/_src/sim/collide.py:1065-2207).

``CollisionPipeline.collide(state, contacts)`` runs the gfx950 collide kernel (AABBs -> broad phase ->
narrow phase -> contact writer) through the C ABI ``nt_collide``.  Contacts live in fixed per-env slots
(slot = pair * cpp + k, env-major SoA); the Newton-shaped flat arrays (``rigid_contact_shape0`` ...,
in the reference's append order) are produced on read by ``nt_contacts_export``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .enums import GeoType, ShapeFlags
from .sdf_pipeline import FlatRowMatcher, SdfLeg, model_has_sdf_pairs, sdf_pair_shape_types_ok

_RIGID_CONTACT_MIN_CAPACITY = 1000  # collide.py: _estimate_rigid_contact_max floor


def _torch():
    import torch  # noqa: PLC0415

    return torch


class Contacts:
    """Rigid contact buffers (contacts.py:227-277).  ``rigid_contact_max`` = env_count * pairs_per_env * cpp."""

    def __init__(self, model, rigid_contact_max: int | None = None, sort_by_key: bool = False,
                 per_contact_shape_properties: bool = False, sdf_leg=None):
        torch = _torch()
        self.model = model
        # rows of the pipeline's mesh-SDF leg (sdf_pipeline.FlatRows): appended after the slot contacts in the flat views
        if sdf_leg is None and model_has_sdf_pairs(model):
            # the SDF pairs are not in the tile pair list (model.py EnvTemplate.sdf_pair): a Contacts built without the
            # pipeline's leg would silently carry no contact for them
            raise ValueError("this model routes shape pairs through the SDF legs: create its Contacts with "
                             "CollisionPipeline(model).contacts(), not Contacts(model)")
        self._flat = sdf_leg.new_rows(per_contact_shape_properties) if sdf_leg is not None else None
        self._sdf_leg = sdf_leg
        # CollisionPipeline(deterministic=True): the flat arrays come out in the reference's sorted order, ascending
        # make_contact_sort_key = (shape0, shape1, sub-contact index) (contact_data.py:60-90, contact_sort.py)
        self.sort_by_key = bool(sort_by_key)
        dm = model.device_model()
        t = model.env
        self._slots = t.np * t.cpp
        self.rigid_contact_max = t.env_count * self._slots if rigid_contact_max is None else int(rigid_contact_max)
        self._slot_contact_max = self.rigid_contact_max
        if self._flat is not None:
            self.rigid_contact_max += self._flat.capacity
        self.soft_contact_max = 0
        ns = max(self._slots, 1)
        dev = dm.device
        self._shape0 = torch.full((ns, t.env_stride), -1, dtype=torch.int32, device=dev)
        self._shape1 = torch.full((ns, t.env_stride), -1, dtype=torch.int32, device=dev)
        self._data = torch.zeros((_lib.NT_CONTACT_FLOATS, ns, t.env_stride), dtype=torch.float32, device=dev)
        self._env_count = torch.zeros(t.env_stride, dtype=torch.int32, device=dev)
        self._pair_hit = torch.zeros((max(t.np, 1), t.env_stride), dtype=torch.uint8, device=dev)
        self._scan = torch.zeros(4 * (t.env_stride + 1), dtype=torch.int32, device=dev)
        # pair-heavy scenes keep the solvers' per-contact records here instead of LDS (nt_model.contact_scratch_in_hbm)
        self._cw = (torch.zeros((15, ns, t.env_stride), dtype=torch.float32, device=dev)
                    if dm.desc.contact_scratch_in_hbm else None)
        # ... and the fused XPBD rollout of that tile its contact records, one 128-byte line per (environment, slot) (nt_contacts.cr: 3 GB
        # for config C5's geometry at 2 048 worlds).  Only that rollout reads them, so they are allocated by prepare_rollout() -- which
        # SolverXPBD.rollout calls on first use and newton_amd.graph.capture before it records -- not for collide-only / per-step /
        # restitution / SDF-leg users (the kernels fall back to the Contacts buffers while nt_contacts.cr is NULL)
        self._cr = None
        self._cr_shape = (t.env_count, ns, 32) if dm.desc.contact_scratch_in_hbm else None
        # optional per-contact stiffness / damping / friction scale (contacts.py:227-277: rigid_contact_stiffness, _damping,
        # _friction; allocated with per_contact_shape_properties, e.g. for hydroelastic faces), slot layout [3][slots][ES];
        # a positive entry overrides the shape materials in eval_body_contact (SemiImplicit / Featherstone)
        self._prop = (torch.zeros((3, ns, t.env_stride), dtype=torch.float32, device=dev)
                      if per_contact_shape_properties else None)
        self._export = None
        self._generation = 0
        self._export_generation = -1
        # extended attribute (contacts.py:170-226): [rigid_contact_max, 6] force / torque on shape0's body, filled by
        # solver.update_contacts(); _impulse holds the solver's per-slot accumulated impulses of the last step
        self.force = None
        self._impulse = None
        if "force" in model.get_requested_contact_attributes():
            self.force = torch.zeros((max(self.rigid_contact_max, 1), 6), dtype=torch.float32, device=dev)
            self._impulse = torch.zeros((6, ns, t.env_stride), dtype=torch.float32, device=dev)

    def prepare_rollout(self) -> None:
        """Allocate what only the fused XPBD rollout of a pair-heavy scene reads (nt_contacts.cr).  Idempotent; call it before a hipGraph
        capture that records a first rollout (a capture must not allocate; newton_amd.graph.capture does it for its `contacts`).  Inside a
        capture that nobody prepared the rollout runs without the buffer (same results, more HBM traffic)."""
        if self._cr is None and self._cr_shape is not None:
            torch = _torch()
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return
            self._cr = torch.zeros(self._cr_shape, dtype=torch.float32, device=self._data.device)

    def _desc(self) -> _lib.nt_contacts:
        d = _lib.nt_contacts()
        d.shape0 = self._shape0.data_ptr()
        d.shape1 = self._shape1.data_ptr()
        d.data = self._data.data_ptr()
        d.env_count = self._env_count.data_ptr()
        d.pair_hit = self._pair_hit.data_ptr()
        if self._cw is not None:
            d.cw = self._cw.data_ptr()
        if self._cr is not None:
            d.cr = self._cr.data_ptr()
        if self._prop is not None:
            d.prop = self._prop.data_ptr()
        if self._flat is not None:
            d.flat = self._flat.desc()
        return d

    def set_slot_properties(self, stiffness=None, damping=None, friction_scale=None):
        """Fill the per-contact overrides for every contact slot ([pairs_per_env * cpp, env_count] arrays or scalars; slot =
        pair * cpp + sub-contact in the device pair order).  Requires per_contact_shape_properties=True."""
        if self._prop is None:
            raise ValueError("Contacts were created without per_contact_shape_properties")
        torch = _torch()
        t = self.model.env
        for k, v in enumerate((stiffness, damping, friction_scale)):
            if v is None:
                continue
            v = torch.as_tensor(v, dtype=torch.float32, device=self._prop.device)
            self._prop[k, : self._slots, : t.env_count] = v if v.ndim == 0 else v.reshape(self._slots, t.env_count)

    # -- Newton-shaped flat views (append order = env, pair, sub-contact; collide.py:166-254) ---------------
    def _exported(self):
        if self._export is not None and self._export_generation == self._generation:
            return self._export
        torch = _torch()
        dm = self.model.device_model()
        cap = max(self.rigid_contact_max, 1)
        dev = dm.device
        e = {
            "count": torch.zeros(1, dtype=torch.int32, device=dev),
            "shape0": torch.full((cap,), -1, dtype=torch.int32, device=dev),
            "shape1": torch.full((cap,), -1, dtype=torch.int32, device=dev),
            "point0": torch.zeros((cap, 3), dtype=torch.float32, device=dev),
            "point1": torch.zeros((cap, 3), dtype=torch.float32, device=dev),
            "offset0": torch.zeros((cap, 3), dtype=torch.float32, device=dev),
            "offset1": torch.zeros((cap, 3), dtype=torch.float32, device=dev),
            "normal": torch.zeros((cap, 3), dtype=torch.float32, device=dev),
            "margin0": torch.zeros((cap,), dtype=torch.float32, device=dev),
            "margin1": torch.zeros((cap,), dtype=torch.float32, device=dev),
        }
        d = self._desc()
        _lib.check(dm.lib.nt_contacts_export(
            C.byref(dm.desc), C.byref(d), self._slot_contact_max, e["count"].data_ptr(), e["shape0"].data_ptr(),
            e["shape1"].data_ptr(), e["point0"].data_ptr(), e["point1"].data_ptr(), e["offset0"].data_ptr(),
            e["offset1"].data_ptr(), e["normal"].data_ptr(), e["margin0"].data_ptr(), e["margin1"].data_ptr(),
            self._scan.data_ptr(), dm.stream()), "nt_contacts_export")
        has_prop = self._prop is not None or (self._flat is not None and self._flat.stiffness is not None)
        if has_prop:  # Contacts.rigid_contact_stiffness / _damping / _friction (contacts.py:227-277), zero = shape materials
            for name in ("stiffness", "damping", "friction"):
                e[name] = torch.zeros((cap,), dtype=torch.float32, device=dev)
            if self._prop is not None:  # slot overrides, in the export's (env, slot) order of the live slots
                t = self.model.env
                live_slots = (self._shape0[: self._slots, : t.env_count] != self._shape1[: self._slots, : t.env_count]).T
                es = torch.nonzero(live_slots)
                ns_ = min(int(es.shape[0]), self._slot_contact_max)
                for k, name in enumerate(("stiffness", "damping", "friction")):
                    e[name][:ns_] = self._prop[k][es[:ns_, 1], es[:ns_, 0]]
        if self._flat is not None:  # the SDF leg's rows follow (collide.py:1999: its launches come last); inert rows are skipped
            f = self._flat
            n0 = min(int(e["count"].item()), self._slot_contact_max)
            nf = min(int(f.row_start[-1].item()), f.capacity)
            live = torch.nonzero(f.shape0[:nf] != f.shape1[:nf]).flatten()
            k = int(live.numel())
            for name in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
                e[name][n0:n0 + k] = getattr(f, name)[live]
            if f.stiffness is not None:
                e["stiffness"][n0:n0 + k] = f.stiffness[live]
                e["damping"][n0:n0 + k] = f.damping[live]
                e["friction"][n0:n0 + k] = f.friction_scale[live]
            e["count"][0] = n0 + k
            self._flat_live, self._flat_n0 = live, n0
        self._sort_order = None
        if self.sort_by_key:
            n = min(int(e["count"].item()), cap)
            if n > 1:
                # a pair's contacts are exported consecutively in sub-contact order, so a stable sort on (shape0, shape1)
                # is the sort on the full key; 32 bits per id (the reference packs 20: contact_data.py:60-90, which
                # aliases beyond 2^20 shapes)
                key = e["shape0"][:n].to(torch.int64) * (1 << 32) + e["shape1"][:n].to(torch.int64)
                order = torch.sort(key, stable=True).indices
                for k, v in e.items():
                    if k != "count":
                        v[:n] = v[:n][order]
                self._sort_order = order  # SolverXPBD.update_contacts applies the same permutation to Contacts.force
        self._export, self._export_generation = e, self._generation
        return e

    def export_order(self):
        """Permutation from the raw append order to the order of the rigid_contact_* arrays (None = identity): row i of the
        flat arrays is raw row export_order()[i].  Non-trivial only with CollisionPipeline(deterministic=True)."""
        self._exported()
        return self._sort_order

    rigid_contact_count = property(lambda self: self._exported()["count"])
    rigid_contact_shape0 = property(lambda self: self._exported()["shape0"])
    rigid_contact_shape1 = property(lambda self: self._exported()["shape1"])
    rigid_contact_point0 = property(lambda self: self._exported()["point0"])
    rigid_contact_point1 = property(lambda self: self._exported()["point1"])
    rigid_contact_offset0 = property(lambda self: self._exported()["offset0"])
    rigid_contact_offset1 = property(lambda self: self._exported()["offset1"])
    rigid_contact_normal = property(lambda self: self._exported()["normal"])
    rigid_contact_margin0 = property(lambda self: self._exported()["margin0"])
    rigid_contact_margin1 = property(lambda self: self._exported()["margin1"])
    # per-contact overrides (None unless the contacts carry them: per_contact_shape_properties or hydroelastic rows)
    rigid_contact_stiffness = property(lambda self: self._exported().get("stiffness"))
    rigid_contact_damping = property(lambda self: self._exported().get("damping"))
    rigid_contact_friction = property(lambda self: self._exported().get("friction"))

    @property
    def rigid_contact_count_per_env(self):
        """[E] int32: contacts emitted per environment (bit-exact parity target)."""
        return self._env_count[: self.model.env.env_count]

    @property
    def candidate_pair_mask(self):
        """[E, pairs_per_env] bool: broad-phase candidate pair set per environment, in Model.shape_contact_pairs order."""
        t = self.model.env
        torch = _torch()
        hit = self._pair_hit[: t.np, : t.env_count].T.bool()
        out = torch.zeros_like(hit)
        out[:, torch.as_tensor(t.pair_order, device=hit.device)] = hit
        return out

    def invalidate_views(self):
        """The device buffers were rewritten behind this object's back (a hipGraph replay of recorded collide launches): drop the
        cached flat views."""
        self._generation += 1

    def clear(self):
        """No contacts (contacts.py:227-277 Contacts.clear zeroes the counts): the slots, and the rows of the SDF legs with the
        per-world ranges and per-body block lists the solvers walk."""
        self._shape0.fill_(-1)
        self._shape1.fill_(-1)
        self._env_count.zero_()
        if getattr(self, "_flat", None) is not None:
            self._flat.row_start.zero_()
            self._flat.body_blk_start.zero_()
            self._flat.shape0.fill_(-1)
            self._flat.shape1.fill_(-1)
        self._generation += 1


class ContactMatcher:
    """Frame-to-frame contact matching (newton/_src/geometry/contact_match.py:602-1055, non-sticky subset): for every contact of
    the current frame the index of the same physical contact in the previous frame's (deterministically ordered) flat arrays,
    MATCH_NOT_FOUND (-1) when its shape pair had no contact, MATCH_BROKEN (-2) when nothing lies within the position / normal
    thresholds or a closer new contact claimed the candidate.  The device kernels work on the fixed-slot layout
    (nt_contacts_match / nt_contacts_save_history: a pair's slots are its key range, no sort, no atomics)."""

    MATCH_NOT_FOUND, MATCH_BROKEN = -1, -2

    def __init__(self, model, pos_threshold: float = 0.0005, normal_dot_threshold: float = 0.995, sticky: bool = False,
                 sdf_leg=None):
        torch = _torch()
        self.model = model
        # the rows of the mesh-SDF leg are matched on their own (world, pair) blocks (sdf_pipeline.FlatRowMatcher)
        self._rows = FlatRowMatcher(sdf_leg, sticky=sticky) if sdf_leg is not None else None
        self._prev_flat_rows = None  # export-order index of every previous row
        self.dm = model.device_model()
        t = model.env
        ns, dev = max(t.np * t.cpp, 1), self.dm.device
        self.pos_threshold, self.normal_dot_threshold = float(pos_threshold), float(normal_dot_threshold)
        self._pos = torch.zeros((3, ns, t.env_stride), dtype=torch.float32, device=dev)
        self._normal = torch.zeros((3, ns, t.env_stride), dtype=torch.float32, device=dev)
        self._live = torch.zeros((ns, t.env_stride), dtype=torch.uint8, device=dev)
        self._match = torch.full((ns, t.env_stride), -1, dtype=torch.int32, device=dev)
        self._prev_flat = None  # flat (export-order) index of every previous slot
        self._reset_mask = None
        h = _lib.nt_contact_history()
        h.prev_pos_world, h.prev_normal, h.prev_live = self._pos.data_ptr(), self._normal.data_ptr(), self._live.data_ptr()
        self.sticky = bool(sticky)
        if self.sticky:  # body-frame points / offsets of the record used last frame (contact_match.py:690-702)
            self._body_frame = torch.zeros((12, ns, t.env_stride), dtype=torch.float32, device=dev)
            h.prev_body_frame = self._body_frame.data_ptr()
        self._h = h

    def reset(self, world_mask=None):
        """Forget the history of the selected worlds (all when None): their next contacts report MATCH_NOT_FOUND."""
        torch = _torch()
        t = self.model.env
        if self._rows is not None:
            self._rows.reset(world_mask)
        if world_mask is None:
            self._live.zero_()
            self._reset_mask = None
        else:
            m = torch.as_tensor(world_mask, device=self.dm.device).to(torch.uint8)[: t.env_count].contiguous()
            self._reset_mask = m
            self._live[:, : t.env_count] *= (1 - m)[None, :]

    def previous_rows_alive(self, n: int):
        """bool [n] over the previous frame's flat rows: True while the row's history has not been reset."""
        torch = _torch()
        t = self.model.env
        alive = torch.zeros(max(n, 1), dtype=torch.bool, device=self.dm.device)
        if self._prev_flat is not None and n > 0:
            keep = (self._prev_flat >= 0) & (self._live[: t.np * t.cpp, : t.env_count] != 0)
            alive[self._prev_flat[keep]] = True
        if self._rows is not None and self._prev_flat_rows is not None and n > 0:
            keep = self._rows.previous_rows_alive() & (self._prev_flat_rows >= 0)
            alive[self._prev_flat_rows[keep]] = True
        return alive[:n]

    def _flat_index(self, live):
        """Export-order index of every live slot: all envs' analytic contacts in (env, slot) order, then the convex ones."""
        torch = _torch()
        t = self.model.env
        E, nas = t.env_count, t.np_analytic * t.cpp
        live = live[:, :E].bool()
        out = torch.full(live.shape, -1, dtype=torch.int64, device=live.device)
        base = 0
        for lo, hi in ((0, nas), (nas, t.np * t.cpp)):
            part = live[lo:hi].T.contiguous()  # [E, slots]
            idx = torch.cumsum(part.reshape(-1).to(torch.int64), 0) - 1 + base
            out[lo:hi] = torch.where(part, idx.reshape(part.shape), torch.full_like(idx.reshape(part.shape), -1)).T
            base += int(part.sum().item())
        return out

    def match(self, state, contacts):
        """-> int32 tensor over the CURRENT frame's flat contacts (export order): previous flat index, -1 or -2.  With
        CollisionPipeline(deterministic=True) both orders are the reference's key-sorted order."""
        torch = _torch()
        dm, t = self.dm, self.model.env
        d_s, d_c = state._desc(), contacts._desc()
        mask_ptr = self._reset_mask.data_ptr() if self._reset_mask is not None else None
        if t.np * t.cpp > 0:  # (a model whose pairs all take the SDF / vertex legs has no slot contacts to match)
            _lib.check(dm.lib.nt_contacts_match(C.byref(dm.desc), C.byref(d_s), C.byref(d_c), C.byref(self._h), self.pos_threshold,
                                                self.normal_dot_threshold, mask_ptr, self._match.data_ptr(), dm.stream()),
                       "nt_contacts_match")
        self._reset_mask = None
        E = t.env_count
        live_now = (contacts._shape0[: t.np * t.cpp] >= 0) & (contacts._shape0[: t.np * t.cpp] != contacts._shape1[: t.np * t.cpp])
        flat_now = self._flat_index(live_now.to(torch.uint8))
        m = self._match[: t.np * t.cpp, :E].to(torch.int64)
        if self._prev_flat is not None:
            slot = m.clamp(min=0)
            env = torch.arange(E, device=m.device)[None, :].expand_as(slot)
            m = torch.where(m >= 0, self._prev_flat[slot, env], m)
        n = int(live_now[:, :E].sum().item())
        order = contacts.export_order()  # deterministic mode re-orders the flat rows: follow it
        k, mr = 0, None
        if self._rows is not None:  # the SDF leg's live rows follow the slot contacts in the raw export order
            live_rows = contacts._flat_live
            k = int(live_rows.numel())
            mr = self._rows.match(state, contacts._flat, self.pos_threshold, self.normal_dot_threshold)[live_rows].to(torch.int64)
            if self._prev_flat_rows is not None:
                mr = torch.where(mr >= 0, self._prev_flat_rows[mr.clamp(min=0)], mr)
        out = torch.full((max(n + k, 1),), -1, dtype=torch.int32, device=m.device)
        sel = flat_now >= 0
        out[flat_now[sel]] = m[sel].to(torch.int32)
        if k:
            out[n:n + k] = mr.to(torch.int32)
        if order is not None:
            out[: order.numel()] = out[: order.numel()][order]
        return out[:n + k]

    def replay_matched(self, state, contacts):
        """Sticky mode (contact_match.py:933-996): matched contacts that still touch keep last frame's body-frame points,
        offsets and normal.  Call after match() -- it uses that call's slot-space result -- and before save_sorted_state()."""
        if not self.sticky:
            raise ValueError("replay_matched requires ContactMatcher(sticky=True)")
        dm = self.dm
        d_s, d_c = state._desc(), contacts._desc()
        if self.model.env.np * self.model.env.cpp > 0:
            _lib.check(dm.lib.nt_contacts_replay_matched(C.byref(dm.desc), C.byref(d_s), C.byref(d_c), C.byref(self._h),
                                                         self._match.data_ptr(), dm.stream()), "nt_contacts_replay_matched")
        if self._rows is not None:
            self._rows.replay_matched(state, contacts._flat)
        contacts._generation += 1

    def save_sorted_state(self, state, contacts):
        """Persist this frame's contacts as the next frame's history (call after match, with the state they were made on)."""
        dm, t = self.dm, self.model.env
        d_s, d_c = state._desc(), contacts._desc()
        if t.np * t.cpp > 0:
            _lib.check(dm.lib.nt_contacts_save_history(C.byref(dm.desc), C.byref(d_s), C.byref(d_c), C.byref(self._h), dm.stream()),
                       "nt_contacts_save_history")
        self._prev_flat = self._flat_index(self._live[: t.np * t.cpp])
        order = contacts.export_order()
        torch = _torch()
        if self._rows is not None:
            self._rows.save_history(state, contacts._flat)
            live_rows, n0 = contacts._flat_live, contacts._flat_n0
            self._prev_flat_rows = torch.full((contacts._flat.capacity,), -1, dtype=torch.int64, device=live_rows.device)
            self._prev_flat_rows[live_rows] = n0 + torch.arange(live_rows.numel(), device=live_rows.device)
        if order is not None:  # flat rows were permuted by the key sort: rank of every raw row in the sorted order
            rank = torch.empty_like(order)
            rank[order] = torch.arange(order.numel(), device=order.device)
            ok = self._prev_flat >= 0
            self._prev_flat[ok] = rank[self._prev_flat[ok]]
            if self._prev_flat_rows is not None:
                ok = self._prev_flat_rows >= 0
                self._prev_flat_rows[ok] = rank[self._prev_flat_rows[ok]]


def estimate_rigid_contact_max(model) -> int:
    """Capacity heuristic of the reference (collide.py:553-652) for models with precomputed pairs:
    max(1000, min(neighbor-budget heuristic, pairs * contacts_per_pair))."""
    types = np.asarray(model.shape_type)
    colliding = (np.asarray(model.shape_flags) & int(ShapeFlags.COLLIDE_SHAPES)) != 0
    plane = colliding & (types == int(GeoType.PLANE))
    non_plane = colliding & ~plane
    n_non_planes = int(non_plane.sum())
    cpp_prim, neighbors = 5, 20
    non_plane_contacts = (n_non_planes * neighbors * cpp_prim) // 2
    world = np.asarray(model.shape_world)
    n_worlds = max(model.world_count, 1)
    glob = world == -1
    per_world_planes = np.bincount(world[~glob & plane], minlength=n_worlds)[:n_worlds] + int((glob & plane).sum())
    per_world_non = np.bincount(world[~glob & non_plane], minlength=n_worlds)[:n_worlds] + int((glob & non_plane).sum())
    plane_pairs = int(np.sum(per_world_planes * per_world_non))
    if n_worlds > 1:
        plane_pairs -= (n_worlds - 1) * int((glob & plane).sum()) * int((glob & non_plane).sum())
    total = non_plane_contacts + plane_pairs * (cpp_prim if n_non_planes else 0)
    if model.shape_contact_pair_count > 0:
        total = min(total, int(model.shape_contact_pair_count) * cpp_prim)
    return max(_RIGID_CONTACT_MIN_CAPACITY, total)


class CollisionPipeline:
    """newton.CollisionPipeline(model, *, broad_phase=..., rigid_contact_max=None, ...)  (collide.py:1104-1133).

    All three broad-phase modes emit the same per-env candidate set (the reference's tests assert exactly that,
    newton/tests/test_broad_phase.py:91-145); with one workgroup per group of environments the per-env pair list
    is small, so every mode evaluates the precomputed per-env pair template against the fresh AABBs.
    """

    _BROAD_PHASES = {"explicit": 0, "nxn": 1, "sap": 2, None: 0}

    def __new__(cls, model=None, **kwargs):
        if model is not None and getattr(model, "is_heterogeneous", False):  # one pipeline per world group behind the same surface (hetero.py)
            from .hetero import GroupedCollisionPipeline  # noqa: PLC0415

            return GroupedCollisionPipeline(cls, model, **kwargs)
        return super().__new__(cls)

    def __init__(self, model, *, broad_phase=None, rigid_contact_max=None, reduce_contacts=True, deterministic=False,
                 sdf_hydroelastic_config=None, envs_per_block: int = 0, contact_matching: str = "disabled",
                 contact_matching_pos_threshold: float = 0.0005, contact_matching_normal_dot_threshold: float = 0.995,
                 contact_report: bool = False, sdf_pairs_per_shape: int = 12, sdf_contacts_per_shape: int = 40,
                 sdf_hydro_faces_per_shape: int = 400, sdf_hydro_staged: bool = True, speculative_config=None, **unsupported):
        if unsupported:
            raise NotImplementedError(f"CollisionPipeline options not supported: {sorted(unsupported)}")
        # collide.py:1087-1132,1239-1246,1823-1836: None disables speculative contacts, and `collide(dt=...)` is then ignored, exactly
        # what this pipeline does; a config asks for swept AABBs + write_contact_speculative (collide.py:258-280), which are not built
        if speculative_config is not None:
            raise NotImplementedError("speculative contacts (CollisionPipeline(speculative_config=...)) are not supported; "
                                      "pass None (the reference's default: collide(dt=...) is then ignored)")
        # frame-to-frame matching (collide.py:1126-1129,1253-1268): "latest" fills contacts.rigid_contact_match_index every
        # collide(); "sticky" additionally replays last frame's contact geometry on matched rows that still touch
        if contact_matching not in ("disabled", "latest", "sticky"):
            raise ValueError(f"contact_matching must be one of 'disabled', 'latest', 'sticky', got {contact_matching!r}")
        if contact_matching_pos_threshold < 0.0:
            raise ValueError(f"contact_matching_pos_threshold must be non-negative, got {contact_matching_pos_threshold}")
        if not -1.0 <= contact_matching_normal_dot_threshold <= 1.0:
            raise ValueError(f"contact_matching_normal_dot_threshold must be in [-1, 1], got {contact_matching_normal_dot_threshold}")
        if contact_report and contact_matching == "disabled":
            raise ValueError('contact_report=True requires contact_matching != "disabled"')
        if broad_phase not in self._BROAD_PHASES:
            raise ValueError(f"broad_phase must be one of 'nxn', 'sap', 'explicit', got {broad_phase!r}")
        self.model = model
        self.dm = model.device_model()  # raises loudly without GPU / extension
        if model.env.np > 0:
            self.dm.require_fit("CollisionPipeline")
        self.broad_phase = broad_phase or "explicit"
        self.params = _lib.nt_collide_params(self._BROAD_PHASES[broad_phase], int(envs_per_block))
        t = model.env
        self._rigid_contact_max = t.env_count * t.np * t.cpp
        # shape pairs with texture SDFs + collision edges on both sides: the mesh-SDF leg (narrow_phase.py:620-640, 2838-3167)
        self._sdf_leg = None
        if model_has_sdf_pairs(model):
            sdf_pair_shape_types_ok(model)
            self._sdf_leg = SdfLeg(model, pairs_per_shape=sdf_pairs_per_shape, contacts_per_shape=sdf_contacts_per_shape,
                                   hydro_config=sdf_hydroelastic_config, hydro_faces_per_shape=sdf_hydro_faces_per_shape,
                                   hydro_staged=sdf_hydro_staged,
                                   # (triangle mesh, convex primitive) pairs: 245 reduction slots per pair, every generated contact when off
                                   triangle_rows_per_pair=245 if reduce_contacts else (1 << 30))
            self._sdf_leg.mesh_plane_reduce = bool(reduce_contacts)  # (triangle mesh, plane / primitive) pairs: every contact when off
            # edge pairs: every contact the edge search admits when off (narrow_phase.py:3044,3097-3130 launches mesh_sdf_collision_kernel
            # instead of the global-reduce kernel); size sdf_contacts_per_shape for it, an overflow is reported by the leg
            self._sdf_leg.edge_reduce = bool(reduce_contacts)
            if contact_matching != "disabled" and self._sdf_leg.has_hydro_pairs:
                # the reference matcher does not cover hydroelastic contacts either (contact_match.py:497-501)
                raise NotImplementedError("contact_matching is not supported for hydroelastic contact pairs")
            self._rigid_contact_max += self._sdf_leg.row_capacity
        model.rigid_contact_max = self._rigid_contact_max
        # fixed slots + ordered reductions: results are reproducible either way; deterministic=True additionally orders the
        # flat contact arrays by the reference's contact sort key (collide.py deterministic mode, contact_sort.py)
        # matching indexes the key-sorted rows of the previous frame, so it implies the deterministic order (the reference
        # builds its ContactSorter for both, collide.py:1653-1667)
        self.contact_matching, self.contact_report = contact_matching, bool(contact_report)
        self.deterministic = bool(deterministic) or contact_matching != "disabled"
        self._matcher = (ContactMatcher(model, contact_matching_pos_threshold, contact_matching_normal_dot_threshold,
                                        sticky=contact_matching == "sticky", sdf_leg=self._sdf_leg)
                         if contact_matching != "disabled" else None)
        self._prev_count = 0

    @property
    def rigid_contact_max(self):
        return self._rigid_contact_max

    def contacts(self, per_contact_shape_properties: bool = False) -> Contacts:
        c = Contacts(self.model, sort_by_key=self.deterministic, per_contact_shape_properties=per_contact_shape_properties,
                     sdf_leg=self._sdf_leg)
        c._contact_matching_mode = self.contact_matching
        if self._matcher is not None:
            torch = _torch()
            dev, n = self.dm.device, max(self._rigid_contact_max, 1)
            c.rigid_contact_match_index = torch.full((n,), -1, dtype=torch.int32, device=dev)
            if self.contact_report:
                c.rigid_contact_new_indices = torch.zeros(n, dtype=torch.int32, device=dev)
                c.rigid_contact_new_count = torch.zeros(1, dtype=torch.int32, device=dev)
                c.rigid_contact_broken_indices = torch.zeros(n, dtype=torch.int32, device=dev)
                c.rigid_contact_broken_count = torch.zeros(1, dtype=torch.int32, device=dev)
        return c

    def reset_contact_matching(self, world_mask=None) -> None:
        """Forget the matching history of the selected worlds (all when None), collide.py:1732-1760."""
        if self._matcher is None:
            raise ValueError('reset_contact_matching requires contact_matching != "disabled"')
        self._matcher.reset(world_mask)
        if world_mask is None:
            self._prev_count = 0

    def _match(self, state, contacts):
        """contacts.rigid_contact_match_index (+ the new / broken report) for this frame, then save the frame as history."""
        torch = _torch()
        if getattr(contacts, "rigid_contact_match_index", None) is None:
            raise ValueError("CollisionPipeline has contact_matching enabled but the Contacts buffer was created without it. "
                             "Use pipeline.contacts() to create a compatible buffer.")
        m = self._matcher.match(state, contacts)
        n = int(m.numel()) if int(contacts.rigid_contact_count[0].item()) > 0 else 0
        contacts.rigid_contact_match_index.fill_(-1)
        contacts.rigid_contact_match_index[:n] = m[:n]
        if self.contact_report:
            new = torch.nonzero(m[:n] < 0).flatten().to(torch.int32)
            contacts.rigid_contact_new_indices[: new.numel()] = new
            contacts.rigid_contact_new_count[0] = new.numel()
            # broken: rows of the previous frame that no contact of this frame matched (worlds reset since are skipped by
            # the matcher: their history is gone, so they report neither matches nor broken rows)
            hit = torch.zeros(max(self._prev_count, 1), dtype=torch.bool, device=m.device)
            ok = m[:n][m[:n] >= 0].to(torch.int64)
            hit[ok] = True
            live_prev = self._matcher.previous_rows_alive(self._prev_count)
            broken = torch.nonzero(~hit[: self._prev_count] & live_prev).flatten().to(torch.int32)
            contacts.rigid_contact_broken_indices[: broken.numel()] = broken
            contacts.rigid_contact_broken_count[0] = broken.numel()
        if self.contact_matching == "sticky":
            self._matcher.replay_matched(state, contacts)
        self._matcher.save_sorted_state(state, contacts)
        self._prev_count = n

    def collide(self, state, contacts: Contacts, *, soft_contact_margin=None, dt=None):
        if contacts.model is not self.model:
            raise ValueError("contacts were created for a different model")
        d_state = state._desc()
        d_ct = contacts._desc()
        if self._sdf_leg is not None:
            if contacts._flat is None:
                raise ValueError("the model has SDF contact pairs: create the Contacts with pipeline.contacts()")
            self._sdf_leg.export_pointers(d_ct)  # world transforms + AABBs of every shape, for the stages below
        _lib.check(self.dm.lib.nt_collide(C.byref(self.dm.desc), C.byref(d_state), C.byref(d_ct), C.byref(self.params),
                                          self.dm.stream()), "nt_collide")
        if self._sdf_leg is not None:
            self._sdf_leg.collide(state, contacts._flat, self.dm.stream())
        contacts._generation += 1
        if self._matcher is not None:
            self._match(state, contacts)
