"""Heterogeneous worlds (newton/_src/sim/model.py:881-900: every per-world entity carries its world index and worlds may differ
in topology; newton/_src/sim/builder.py add_world / add_builder with different sub-builders).

The fused kernels keep one topology per launch (one workgroup = several environments of the SAME template, tables in LDS,
DESIGN.md section 1).  A model whose worlds differ is therefore cut into *world groups*: maximal contiguous runs of worlds with
one topology (``world_runs``), each a valid homogeneous sub-model (``worlds.slice_worlds``) with its own device tables, and
the Newton-shaped objects a caller holds (State / Control / Contacts / CollisionPipeline / the solver classes) become thin
composites over the groups' objects:

* the reference lays every flat array out world-major, so concatenating the groups' arrays in group order IS the global array
  (the same property the per-rank shards use) - getters concatenate, setters split;
* ``collide`` / ``step`` / ``rollout`` issue one launch per group, each group on its own HIP stream forked from and joined
  back into the caller's stream (the groups are independent, and a small group does not fill 256 CUs on its own);
* contact shape ids are translated back to the global model's shape ids, rows are the groups' rows in group order, which is
  the reference's append order over the world-major candidate pairs.

Nothing here touches the arithmetic: a group of k identical worlds steps exactly like a k-world replicated model.
"""
from __future__ import annotations

import functools

import numpy as np

from .worlds import _BODY, _DOF, _JOINT, _SHAPE, slice_worlds


def _torch():
    import torch  # noqa: PLC0415

    return torch


def _world_signature(model, w, spans) -> bytes:
    """Everything the EnvTemplate must find identical between two worlds, relative to the world's own first body / joint /
    coordinate / shape (model.py:881-900 start offsets removed)."""
    (b0, b1), (j0, j1), sh = spans
    jt = np.asarray(model.joint_type[j0:j1], dtype=np.int64)

    def rel(a, off):
        a = np.asarray(a, dtype=np.int64)
        return np.where(a >= 0, a - off, a)

    parts = [np.asarray([b1 - b0, j1 - j0, len(sh)], dtype=np.int64), jt, rel(model.joint_parent[j0:j1], b0),
             rel(model.joint_child[j0:j1], b0), np.asarray(model.joint_dof_dim[j0:j1], dtype=np.int64).ravel(),
             np.asarray(model.joint_q_start[j0:j1], dtype=np.int64) - (int(model.joint_q_start[j0]) if j1 > j0 else 0),
             np.asarray(model.joint_qd_start[j0:j1], dtype=np.int64) - (int(model.joint_qd_start[j0]) if j1 > j0 else 0),
             np.asarray(model.shape_type, dtype=np.int64)[sh], rel(np.asarray(model.shape_body)[sh], b0),
             np.asarray(model.shape_flags, dtype=np.int64)[sh], np.asarray(model.shape_collision_group, dtype=np.int64)[sh],
             np.asarray(model.shape_mesh_count, dtype=np.int64)[sh]]
    return b"|".join(np.ascontiguousarray(p).tobytes() for p in parts)


def world_runs(model) -> list[tuple[int, int]]:
    """Maximal contiguous runs [begin, end) of worlds that share one topology signature."""
    W = model.world_count
    if W <= 1:
        return [(0, max(W, 1))]

    def spans_of(world_arr):
        w = np.asarray(world_arr)
        loc = w[w >= 0]
        first = int(np.flatnonzero(w >= 0)[0]) if len(loc) else 0
        if len(loc) and np.any(np.diff(loc) < 0):
            raise NotImplementedError("worlds must be laid out contiguously (world-major)")
        bounds = first + np.searchsorted(loc, np.arange(W + 1))
        return [(int(bounds[i]), int(bounds[i + 1])) for i in range(W)]

    bs, js = spans_of(model.body_world), spans_of(model.joint_world)
    sw = np.asarray(model.shape_world)
    pairs = np.asarray(model.shape_contact_pairs, dtype=np.int64).reshape(-1, 2)
    n_glob_front = int(np.flatnonzero(sw >= 0)[0]) if np.any(sw >= 0) else 0
    sigs = []
    for w in range(W):
        sh = np.flatnonzero(sw == w)
        sig = _world_signature(model, w, (bs[w], js[w], sh))
        if len(pairs) and len(sh):  # candidate pairs of this world, local ids relative to the world's first shape
            s0 = int(sh[0])
            mine = pairs[(sw[pairs[:, 0]] == w) | (sw[pairs[:, 1]] == w)]
            loc = np.where(sw[mine] >= 0, mine - s0, np.where(mine < n_glob_front, -1 - mine, -1_000_000 - (mine - len(sw))))
            sig += b"|" + np.ascontiguousarray(loc).tobytes()
        sigs.append(sig)
    runs, b = [], 0
    for w in range(1, W + 1):
        if w == W or sigs[w] != sigs[b]:
            runs.append((b, w))
            b = w
    return runs


class WorldGroups:
    """The homogeneous sub-models of a heterogeneous Model, in world order."""

    def __init__(self, model):
        self.model = model
        self.ranges: list[tuple[int, int]] = []
        self.parts = []
        for b, e in world_runs(model):
            part = slice_worlds(model, b, e)
            if part.env is None:  # the signature missed a difference the EnvTemplate insists on: one world per group
                for w in range(b, e):
                    self.ranges.append((w, w + 1))
                    self.parts.append(slice_worlds(model, w, w + 1))
            else:
                self.ranges.append((b, e))
                self.parts.append(part)
        for p in self.parts:
            p._requested_state_attributes = model._requested_state_attributes  # shared sets: later requests reach the groups
            p._requested_contact_attributes = model._requested_contact_attributes
        self._streams = None
        self.concurrent = True  # False: the groups' launches stay on the caller's stream, one after the other

    def __len__(self):
        return len(self.parts)

    _HOST_KEYS = _BODY + _JOINT + _DOF + _SHAPE + ("gravity", "joint_q", "joint_target_q")

    def sync_host(self):
        """The groups hold copies of the global model's host arrays taken when they were cut.  Callers edit the global arrays
        between finalize() and the first State / solver (initial poses, masses, gains): re-slice every group that has not been
        uploaded yet.  Called whenever a composite object is created."""
        for (b, e), part in zip(self.ranges, self.parts):
            if part._dev is not None:
                continue
            fresh = slice_worlds(self.model, b, e)
            if fresh.env is None:
                raise NotImplementedError("heterogeneous worlds: the topology of a world group changed after it was cut")
            for k in self._HOST_KEYS:
                setattr(part, k, getattr(fresh, k))

    def refresh_parameters(self):
        """Host arrays of the global model were edited (masses, gains, materials, gravity ...): copy them into the groups and
        re-upload (solver.py:394-440 notify_model_changed)."""
        for (b, e), part in zip(self.ranges, self.parts):
            fresh = slice_worlds(self.model, b, e)
            if fresh.env is None:
                raise NotImplementedError("heterogeneous worlds: the topology of a world group changed after a runtime edit")
            for k in self._HOST_KEYS:
                setattr(part, k, getattr(fresh, k))
            part.notify_model_changed()

    # -- one launch per group, groups on sibling streams ---------------------------------------------------------------
    def run(self, fn):
        """fn(i, part) for every group.  With more than one group on a real device each group runs on its own stream between
        a fork from and a join into the caller's current stream."""
        torch = _torch()
        dev = self.parts[0].device
        if len(self.parts) == 1 or not self.concurrent or not _real_device(torch):
            for i, p in enumerate(self.parts):
                fn(i, p)
            return
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=dev) for _ in self.parts]
        cur = torch.cuda.current_stream(dev)
        fork = torch.cuda.Event()
        fork.record(cur)
        for i, (p, s) in enumerate(zip(self.parts, self._streams)):
            s.wait_event(fork)
            with torch.cuda.stream(s):
                fn(i, p)
            done = torch.cuda.Event()
            done.record(s)
            cur.wait_event(done)


def _real_device(torch) -> bool:
    try:
        return bool(torch.cuda.is_available()) and torch.version.hip is not None and torch.cuda.device_count() > 0 and \
            not getattr(torch.cuda, "_newton_emulated", False)
    except Exception:  # noqa: BLE001
        return False


def _cat(values):
    torch = _torch()
    if isinstance(values[0], torch.Tensor):
        return torch.cat(values)
    return np.concatenate(values)


def _split(value, sizes):
    torch = _torch()
    if isinstance(value, torch.Tensor):
        flat = value.reshape(sum(sizes), -1) if value.ndim > 1 else value
        return list(torch.split(flat, sizes))
    a = np.asarray(value)
    a = a.reshape(sum(sizes), -1) if a.ndim > 1 else a
    return np.split(a, np.cumsum(sizes)[:-1])


class _Composite:
    """Concatenating getter / splitting setter over the groups' containers."""

    _PART_CLASS = None

    def __init__(self, model):
        self.model = model
        self.groups: WorldGroups = model.world_groups
        self.groups.sync_host()
        self.parts = [self._PART_CLASS(p) for p in self.groups.parts]
        self.requires_grad = False
        for name, (_ncomp, _slots, default) in self._PART_CLASS._FIELDS.items():
            if default is not None:  # initial values come from the GLOBAL model's arrays as they are now (model.py:1779-1863)
                self._set(name, getattr(model, default))

    def _rows(self, name):
        out = []
        for c in self.parts:
            ncomp, slots, _ = c._FIELDS[name]
            out.append(c.model.env.env_count * getattr(c.model.env, slots))
        return out

    def _get(self, name):
        return _cat([getattr(c, name) for c in self.parts])

    def _set(self, name, value):
        rows = self._rows(name)
        torch = _torch()
        n = value.numel() if isinstance(value, torch.Tensor) else np.asarray(value).size
        ncomp = self.parts[0]._FIELDS[name][0]
        if n != sum(rows) * ncomp:
            raise ValueError(f"{name}: expected {sum(rows) * ncomp} values, got {n}")
        if ncomp > 1 and not isinstance(value, torch.Tensor):
            value = np.asarray(value).reshape(-1, ncomp)
        elif ncomp > 1:
            value = value.reshape(-1, ncomp)
        for c, v in zip(self.parts, _split(value, rows)):
            setattr(c, name, v)


def _split_world_mask(world_mask, groups):
    """Per-group pieces of a world mask of length world_count or the reference's world_count + 1 (trailing global-entity
    slot, core/reset.py:13-60); None stays None."""
    if world_mask is None:
        return None
    sizes = [e - b for b, e in groups.ranges]
    n = int(world_mask.shape[0]) if hasattr(world_mask, "shape") else len(world_mask)
    if n not in (sum(sizes), sum(sizes) + 1):
        raise ValueError(f"'world_mask' length {n} must equal model.world_count + 1 ({sum(sizes) + 1})")
    return _split(world_mask[: sum(sizes)], sizes)


def _composite_property(name):
    return property(lambda self: self._get(name), lambda self, v: self._set(name, v))


@functools.lru_cache(maxsize=1)
def _make_state_classes():
    from .state import Control, State  # noqa: PLC0415

    class GroupedState(_Composite):
        """State of a heterogeneous model (state.py:113-171): same attributes, world-major concatenation of the groups."""

        _PART_CLASS = State
        body_q = _composite_property("body_q")
        body_qd = _composite_property("body_qd")
        body_f = _composite_property("body_f")
        joint_q = _composite_property("joint_q")
        joint_qd = _composite_property("joint_qd")

        def __init__(self, model):
            super().__init__(model)
            self.particle_count = 0

        @property
        def body_parent_f(self):
            vals = [c.body_parent_f for c in self.parts]
            return None if any(v is None for v in vals) else _cat(vals)

        @property
        def body_count(self):
            return self.model.body_count

        @property
        def joint_coord_count(self):
            return self.model.joint_coord_count

        @property
        def joint_dof_count(self):
            return self.model.joint_dof_count

        def clear_forces(self):
            for c in self.parts:
                c.clear_forces()

        def assign(self, other):
            for c, o in zip(self.parts, other.parts):
                c.assign(o)

        def reset(self, source, world_mask=None):
            """State.reset over the groups (solver.py:344-375): the mask is cut at the group boundaries."""
            masks = _split_world_mask(world_mask, self.groups)
            for i, (c, o) in enumerate(zip(self.parts, source.parts)):
                c.reset(o, None if masks is None else masks[i])

    class GroupedControl(_Composite):
        """Control of a heterogeneous model (control.py:31-68)."""

        _PART_CLASS = Control
        joint_f = _composite_property("joint_f")
        joint_target_q = _composite_property("joint_target_q")
        joint_target_qd = _composite_property("joint_target_qd")

        def clear(self, model=None):
            for c in self.parts:
                c.clear(c.model if model is not None else None)

    return GroupedState, GroupedControl


class GroupedContacts:
    """Contacts of a heterogeneous model (contacts.py:227-277), shape ids of the global model.  Row order = the reference's
    append order: its primitive narrow-phase launch writes first and its MPR / GJK launch second (narrow_phase.py:458-1014,
    1221-1452), each over the world-major candidate pairs - so the analytic rows of every group in group order, then the
    convex rows of every group (a group's own export already has this two-segment shape, nt_contacts_export)."""

    def __init__(self, model, parts):
        self.model = model
        self.parts = parts
        self.rigid_contact_max = sum(c.rigid_contact_max for c in parts)
        self.soft_contact_max = 0
        from .model import pair_types_analytic  # noqa: PLC0415

        torch = _torch()
        nt_ = int(np.max(model.shape_type)) + 1 if model.shape_count else 1
        tab = np.array([[pair_types_analytic(a, b) for b in range(nt_)] for a in range(nt_)], dtype=bool)
        dev = parts[0]._shape0.device
        self._analytic = torch.from_numpy(tab).to(dev)
        self._types = [torch.as_tensor(np.asarray(c.model.shape_type), dtype=torch.int64, device=dev) for c in parts]
        self._gids = [torch.as_tensor(np.asarray(c.model._global_shape_ids), dtype=torch.int64, device=dev) for c in parts]
        self._seg_cache = (None, None)  # (generations, segments): one host sync per group and per collide, not per field read
        self._row_cache = {}

    def _segments(self):
        """Per group (row count, analytic row count) of its current export."""
        torch = _torch()
        gen = tuple(c._generation for c in self.parts)
        if self._seg_cache[0] == gen:
            return self._seg_cache[1]
        seg = []
        for c, ty in zip(self.parts, self._types):
            n = min(int(c.rigid_contact_count.cpu().numpy()[0]), c.rigid_contact_max)
            a, b = c.rigid_contact_shape0[:n].to(torch.int64), c.rigid_contact_shape1[:n].to(torch.int64)
            na = int(self._analytic[ty[a], ty[b]].sum().item()) if n and not c.sort_by_key else n
            seg.append((n, na))
        self._seg_cache = (gen, seg)
        self._row_cache = {}
        return seg

    @property
    def rigid_contact_count(self):
        torch = _torch()
        return torch.stack([c.rigid_contact_count.reshape(-1)[0] for c in self.parts]).sum().reshape(1).to(torch.int32)

    def _rows(self, get, translate=False, key=None):
        torch = _torch()
        seg = self._segments()
        if key is not None and key in self._row_cache:
            return self._row_cache[key]
        first, second = [], []
        for c, ids, (n, na) in zip(self.parts, self._gids, seg):
            v = get(c)[:n]
            if translate:
                v = ids[v.to(torch.int64)].to(torch.int32)
            first.append(v[:na])
            second.append(v[na:])
        out = torch.cat(first + second)
        if key is not None:
            self._row_cache[key] = out
        return out

    @staticmethod
    def _field(name, translate=False):
        return property(lambda self: self._rows(lambda c: getattr(c, name), translate, key=name))

    @property
    def rigid_contact_count_per_env(self):
        return _cat([c.rigid_contact_count_per_env for c in self.parts])

    @property
    def force(self):
        if any(c.force is None for c in self.parts):
            return None
        return self._rows(lambda c: c.force)

    def clear(self):
        for c in self.parts:
            c.clear()

    def invalidate_views(self):
        for c in self.parts:
            c.invalidate_views()


GroupedContacts.rigid_contact_shape0 = GroupedContacts._field("rigid_contact_shape0", True)
GroupedContacts.rigid_contact_shape1 = GroupedContacts._field("rigid_contact_shape1", True)
for _n in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
    setattr(GroupedContacts, "rigid_contact_" + _n, GroupedContacts._field("rigid_contact_" + _n))


class GroupedCollisionPipeline:
    """CollisionPipeline over the world groups (collide.py:1765-2207: one collide() per frame on the whole model)."""

    def __init__(self, cls, model, **kwargs):
        self.model = model
        self.groups: WorldGroups = model.world_groups
        if kwargs.get("rigid_contact_max") is not None:
            raise NotImplementedError("heterogeneous worlds: rigid_contact_max is derived per world group")
        # a group's sorted export is group-local: with a world -1 shape (ground plane) in the pairs its rows would sit inside
        # every group instead of in one global (shape0, shape1) block, and match indices would be group-local row numbers
        for opt, off in (("deterministic", False), ("contact_matching", "disabled"), ("contact_report", False)):
            if kwargs.get(opt, off) != off:
                raise NotImplementedError(f"heterogeneous worlds: CollisionPipeline({opt}=...) is not supported; build the worlds "
                                          "with one topology (ModelBuilder.replicate) to use it")
        self.groups.sync_host()
        self.parts = [cls(p, **kwargs) for p in self.groups.parts]
        self.deterministic = self.parts[0].deterministic

    @property
    def rigid_contact_max(self):
        return sum(p.rigid_contact_max for p in self.parts)

    def contacts(self, **kwargs):
        return GroupedContacts(self.model, [p.contacts(**kwargs) for p in self.parts])

    def reset_contact_matching(self, world_mask=None):
        masks = _split_world_mask(world_mask, self.groups)
        for i, p in enumerate(self.parts):
            p.reset_contact_matching(None if masks is None else masks[i])

    def collide(self, state, contacts, **kwargs):
        self.groups.run(lambda i, _p: self.parts[i].collide(state.parts[i], contacts.parts[i], **kwargs))


class GroupedSolver:
    """One solver instance per world group behind the reference's solver surface (solvers/solver.py:190-450)."""

    def __init__(self, cls, model, *args, **kwargs):
        self.model = model
        self.groups: WorldGroups = model.world_groups
        self.groups.sync_host()
        self.parts = [cls(p, *args, **kwargs) for p in self.groups.parts]

    @property
    def device(self):
        return self.model.device

    @staticmethod
    def _part(obj, i):
        return None if obj is None else obj.parts[i]

    def step(self, state_in, state_out, control, contacts, dt):
        self.groups.run(lambda i, _p: self.parts[i].step(state_in.parts[i], state_out.parts[i], self._part(control, i),
                                                         self._part(contacts, i), dt))

    def rollout(self, state_0, state_1, control, contacts, dt, substeps, **kwargs):
        """substeps x {clear_forces, collide, step} per group in one launch each; returns the state holding the result."""
        res = [None] * len(self.parts)

        def go(i, _p):
            res[i] = self.parts[i].rollout(state_0.parts[i], state_1.parts[i], self._part(control, i),
                                           self._part(contacts, i), dt, substeps, **kwargs)

        self.groups.run(go)
        return state_1 if res[0] is state_1.parts[0] else state_0

    def update_contacts(self, contacts, state=None):
        for i, s in enumerate(self.parts):
            s.update_contacts(contacts.parts[i], self._part(state, i))

    def notify_model_changed(self, flags):
        self.groups.refresh_parameters()
        for s in self.parts:
            s.notify_model_changed(flags)

    def reset(self, state, world_mask=None, flags=None):
        masks = _split_world_mask(world_mask, self.groups)
        for i, s in enumerate(self.parts):
            s.reset(state.parts[i], None if masks is None else masks[i], flags)
