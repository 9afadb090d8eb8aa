"""ArticulationView -- the subset of newton.selection.ArticulationView (newton/_src/utils/selection.py:500-1800) that an RL
loop calls between steps: per-world getters / setters for root pose and twist, dof positions / velocities / forces and
link poses, with an optional world mask, plus masked eval_fk.

With the env-major SoA layout these are not gather/scatter kernels: ``State._soa[name]`` has shape
``[comp, slots_per_env, env_stride]``, so "the dofs of every world" is a strided *view* ``soa[0, a:b, :E].T`` (no copy on
read) and a masked set is one ``torch.where`` on that view.  One articulation per world is selected by label pattern
(``fnmatch``); the view covers every world, like the reference with ``exclude_joint_types`` etc. left at their defaults.
"""
from __future__ import annotations

import fnmatch

import numpy as np

from ..articulation import eval_fk
from ..enums import JointType


def _torch():
    import torch  # noqa: PLC0415

    return torch


class ArticulationView:
    def __init__(self, model, pattern: str = "*", verbose: bool = False):
        self.model = model
        t = model.env
        if t.na == 0:
            raise ValueError("the model has no articulations")
        labels = list(getattr(model, "articulation_label", [])) or [f"articulation_{i}" for i in range(model.articulation_count)]
        per_env = labels[: t.na]
        hits = [i for i, lab in enumerate(per_env) if fnmatch.fnmatch(str(lab), pattern)]
        if not hits:
            raise KeyError(f"no articulation matches {pattern!r} (have {per_env})")
        if len(hits) > 1:
            raise NotImplementedError("ArticulationView selects one articulation per world; refine the pattern")
        a = hits[0]
        self.articulation = a
        self.world_count = self.count = t.env_count
        j0, j1 = int(t.art_start[a]), int(t.art_start[a + 1])
        self.joint_range = (j0, j1)
        q_edges = np.concatenate([t.joint_q_start, [t.nc]])
        qd_edges = np.concatenate([t.joint_qd_start, [t.nd]])
        self.coord_range = (int(q_edges[j0]), int(q_edges[j1]))
        self.dof_range = (int(qd_edges[j0]), int(qd_edges[j1]))
        self.link_ids = [int(t.joint_child[j]) for j in range(j0, j1)]
        self.link_count = len(self.link_ids)
        self.joint_count = j1 - j0
        self.joint_dof_count = self.dof_range[1] - self.dof_range[0]
        self.joint_coord_count = self.coord_range[1] - self.coord_range[0]
        self.is_floating_base = int(t.joint_type[j0]) == int(JointType.FREE)
        self.is_fixed_base = not self.is_floating_base
        if verbose:
            print(f"ArticulationView: {self.count} worlds x articulation {a} ({self.joint_count} joints, "
                  f"{self.joint_dof_count} dofs, {self.link_count} links)")

    # ------------------------------------------------------------------ plumbing
    def _rows(self, source, name, lo, hi):
        """[world, hi-lo] view (GPU: zero-copy strided view of the SoA buffer; host: reshaped AoS array)."""
        E = self.world_count
        if getattr(self.model, "is_gpu", False) and hasattr(source, "_soa"):
            return source._soa[name][0, lo:hi, :E].T
        arr = np.asarray(getattr(source, name)).reshape(E, -1)
        return arr[:, lo:hi]

    def _mask(self, mask, like):
        if mask is None:
            return None
        if hasattr(like, "device") and not isinstance(like, np.ndarray):
            torch = _torch()
            m = torch.as_tensor(mask, device=like.device).bool()
        else:
            m = np.asarray(mask, dtype=bool)
        if m.shape[0] not in (self.world_count,):
            raise ValueError(f"mask must have one entry per world ({self.world_count}), got {tuple(m.shape)}")
        return m

    def _assign(self, target, name, lo, hi, values):
        """Write a [world, hi-lo] block back (host models keep AoS numpy arrays behind properties)."""
        E = self.world_count
        if getattr(self.model, "is_gpu", False) and hasattr(target, "_soa"):
            return  # views alias the SoA buffer: nothing to write back
        full = np.asarray(getattr(target, name)).reshape(E, -1).copy()
        full[:, lo:hi] = values
        setattr(target, name, full.reshape(-1))

    def _set(self, target, name, lo, hi, values, mask):
        view = self._rows(target, name, lo, hi)
        is_torch = not isinstance(view, np.ndarray)
        if is_torch:
            torch = _torch()
            vals = torch.as_tensor(values, device=view.device, dtype=view.dtype).reshape(view.shape)
            m = self._mask(mask, view)
            view.copy_(vals if m is None else torch.where(m[:, None], vals, view))
        else:
            vals = np.asarray(values, dtype=view.dtype).reshape(view.shape)
            m = self._mask(mask, view)
            new = vals if m is None else np.where(m[:, None], vals, view)
            self._assign(target, name, lo, hi, new)

    # ------------------------------------------------------------------ root
    def get_root_transforms(self, source):
        if not self.is_floating_base:
            X = np.asarray(self.model.joint_X_p).reshape(self.world_count, -1, 7)[:, self.joint_range[0]]
            return X
        c0 = self.coord_range[0]
        return self._rows(source, "joint_q", c0, c0 + 7)

    def set_root_transforms(self, target, values, mask=None):
        """Call :meth:`eval_fk` afterwards to propagate to the links (like the reference)."""
        if not self.is_floating_base:
            raise NotImplementedError("fixed-base root transforms live in Model.joint_X_p; edit the model and notify the solver")
        c0 = self.coord_range[0]
        self._set(target, "joint_q", c0, c0 + 7, values, mask)

    def get_root_velocities(self, source):
        if not self.is_floating_base:
            return None
        d0 = self.dof_range[0]
        return self._rows(source, "joint_qd", d0, d0 + 6)

    def set_root_velocities(self, target, values, mask=None):
        if not self.is_floating_base:
            return
        d0 = self.dof_range[0]
        self._set(target, "joint_qd", d0, d0 + 6, values, mask)

    # ------------------------------------------------------------------ dofs
    def get_dof_positions(self, source):
        return self._rows(source, "joint_q", *self.coord_range)

    def set_dof_positions(self, target, values, mask=None):
        self._set(target, "joint_q", *self.coord_range, values, mask)

    def get_dof_velocities(self, source):
        return self._rows(source, "joint_qd", *self.dof_range)

    def set_dof_velocities(self, target, values, mask=None):
        self._set(target, "joint_qd", *self.dof_range, values, mask)

    def get_dof_forces(self, control):
        return self._rows(control, "joint_f", *self.dof_range)

    def set_dof_forces(self, control, values, mask=None):
        self._set(control, "joint_f", *self.dof_range, values, mask)

    # ------------------------------------------------------------------ links
    def _links(self, source, name, ncomp):
        E, nb = self.world_count, self.model.env.nb
        if getattr(self.model, "is_gpu", False) and hasattr(source, "_soa"):
            soa = source._soa[name]  # [ncomp, nb, ES]
            return soa[:, self.link_ids, :E].permute(2, 1, 0)
        arr = np.asarray(getattr(source, name)).reshape(E, nb, ncomp)
        return arr[:, self.link_ids]

    def get_link_transforms(self, source):
        """[world, link, 7]"""
        return self._links(source, "body_q", 7)

    def get_link_velocities(self, source):
        """[world, link, 6]"""
        return self._links(source, "body_qd", 6)

    # ------------------------------------------------------------------ kinematics
    def eval_fk(self, state, mask=None):
        """newton.eval_fk restricted to the selected worlds: body_q / body_qd of unselected worlds are left untouched."""
        model = self.model
        if mask is None:
            eval_fk(model, state.joint_q, state.joint_qd, state)
            return
        keep_q, keep_qd = state.body_q, state.body_qd
        keep_q = keep_q.clone() if hasattr(keep_q, "clone") else np.array(keep_q, copy=True)
        keep_qd = keep_qd.clone() if hasattr(keep_qd, "clone") else np.array(keep_qd, copy=True)
        eval_fk(model, state.joint_q, state.joint_qd, state)
        E, nb = self.world_count, model.env.nb
        new_q, new_qd = state.body_q, state.body_qd
        m = self._mask(mask, new_q)
        if isinstance(new_q, np.ndarray):
            sel = np.repeat(m, nb)
            state.body_q = np.where(sel[:, None], new_q, keep_q)
            state.body_qd = np.where(sel[:, None], new_qd, keep_qd)
        else:
            torch = _torch()
            sel = m.repeat_interleave(nb)
            state.body_q = torch.where(sel[:, None], new_q, keep_q)
            state.body_qd = torch.where(sel[:, None], new_qd, keep_qd)
