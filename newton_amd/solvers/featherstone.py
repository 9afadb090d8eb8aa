"""SolverFeatherstone -- drop-in for newton.solvers.SolverFeatherstone
(newton/_src/solvers/featherstone/solver_featherstone.py:136-1066), rigid articulations only.

One ``nt_featherstone_step`` launch replaces the reference's ~20 launches per step: FK, the RNEA passes, penalty contacts
(``eval_body_contact``), the joint-space inertia ``H = J^T M J``, its Cholesky factorisation and solve, generalized
integration and the final FK all run in one gfx950 kernel with every intermediate (S, I_s, H, L ...) resident in LDS.

Like the reference, ``step`` integrates ``joint_q`` / ``joint_qd``, rebuilds ``body_q`` / ``body_qd`` of ``state_out`` and
refreshes ``state_in.body_q`` from ``state_in.joint_q``.  ``state_in is state_out`` is allowed.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from ..enums import BodyFlags, JointType
from .solver import SolverBase


class SolverFeatherstone(SolverBase):
    def __init__(self, model, *, angular_damping: float = 0.05, update_mass_matrix_interval: int = 1,
                 friction_smoothing: float = 1.0, use_tile_gemm: bool = False, fuse_cholesky: bool = True,
                 envs_per_block: int = 0, mass_matrix: str = "tree"):
        super().__init__(model)
        # "tree" (default): composite-inertia H and the dof-tree L^T D L factorisation (nt_featherstone_params.dense_mass_matrix
        # = 0; same solution within the 1e-5 contract); "dense": the reference's operation order (H = J^T M J over the dense
        # lower triangle, dense_cholesky, dense_subs), as close to the reference's bits as the kernels get
        if mass_matrix not in ("tree", "dense"):
            raise ValueError(f"mass_matrix must be 'tree' or 'dense', got {mass_matrix!r}")
        self.mass_matrix = mass_matrix
        t = model.env
        if int(update_mass_matrix_interval) < 1:
            raise ValueError("update_mass_matrix_interval must be >= 1")
        if t.nj == 0 or t.na == 0:
            raise NotImplementedError("SolverFeatherstone needs articulated bodies (every body behind a joint, "
                                      "articulations contiguous and identical in every world)")
        jt = np.asarray(t.joint_type)
        if np.any(jt == int(JointType.ROD)):
            raise NotImplementedError("SolverFeatherstone: ROD joints are not supported")
        if not np.array_equal(np.asarray(t.joint_child), np.arange(t.nj)) or t.nb != t.nj:
            # the reference's eval_rigid_mass indexes body_I_s by joint index (kernels.py:1466-1480)
            raise NotImplementedError("SolverFeatherstone: body j must be the child of joint j")
        kin = (np.asarray(t.body_flags) & int(BodyFlags.KINEMATIC)) != 0
        if np.any(kin & (np.asarray(t.joint_parent) >= 0)):  # child of joint j is body j (checked above)
            raise ValueError("SolverFeatherstone: only root bodies (joint parent = world) can be kinematic")
        self.angular_damping = angular_damping
        # every k-th step rebuilds P / H and refactorises; in between the kernels reuse the factor of the last rebuild, kept in
        # HBM ([nd * max_art_dofs][ES]); solver_featherstone.py:141,767 (_step, _mass_matrix_dirty)
        self.update_mass_matrix_interval = int(update_mass_matrix_interval)
        self._step, self._mass_matrix_dirty, self._factor_cache = 0, False, None
        if self.update_mass_matrix_interval > 1:
            import torch

            self._factor_cache = torch.zeros((max(t.nd * t.max_art_dofs, 1), t.env_stride), dtype=torch.float32,
                                             device=self.dm.device)
        self.friction_smoothing = friction_smoothing
        self.use_tile_gemm = use_tile_gemm      # accepted for signature parity; H never leaves LDS here
        self.fuse_cholesky = fuse_cholesky
        self.envs_per_block = int(envs_per_block)

    def step(self, state_in, state_out, control, contacts, dt: float) -> None:
        dm = self.dm
        if control is None:
            if not hasattr(self, "_control"):
                self._control = self.model.control()
            control = self._control
        p = self._params()
        d_in, d_out, d_c = state_in._desc(), state_out._desc(), control._desc()
        d_in = self._state_desc_with_sdf_forces(state_in, contacts, self.friction_smoothing)
        d_ct = contacts._desc() if contacts is not None else None
        _lib.check(dm.lib.nt_featherstone_step(C.byref(dm.desc), C.byref(p), C.byref(d_in), C.byref(d_out), C.byref(d_c),
                                               C.byref(d_ct) if d_ct is not None else None, float(dt),
                                               self.envs_per_block, dm.stream()), "nt_featherstone_step")
        self._step += 1
        self._mass_matrix_dirty = False

    def _params(self):
        p = _lib.nt_featherstone_params(float(self.angular_damping), float(self.friction_smoothing))
        p.dense_mass_matrix = 1 if self.mass_matrix == "dense" else 0
        if self._factor_cache is not None:
            p.update_mass_matrix_interval, p.step_index = self.update_mass_matrix_interval, self._step
            p.force_update, p.mass_matrix_cache = int(self._mass_matrix_dirty), self._factor_cache.data_ptr()
        return p

    def notify_model_changed(self, flags) -> None:
        """Body / joint-dof property edits invalidate the cached factor (solver_featherstone.py:284-288)."""
        super().notify_model_changed(flags)
        self._mass_matrix_dirty = True

    def rollout(self, state_0, state_1, control, contacts, dt: float, substeps: int):
        """substeps x {clear_forces; collide; step; swap} in ONE launch (``nt_featherstone_rollout``); returns the state
        object holding the result (state_0 for an even number of substeps, state_1 for odd -- the reference loop's swap)."""
        dm = self.dm
        if control is None:
            if not hasattr(self, "_control"):
                self._control = self.model.control()
            control = self._control
        leg = getattr(contacts, "_sdf_leg", None)
        if leg is not None:
            # the SDF legs of collide() are a chain of launches of their own (newton_amd/sdf_pipeline.py): run the reference loop
            # launch by launch, like SolverXPBD.rollout does for such models
            cp = _lib.nt_collide_params(0, self.envs_per_block)
            for _ in range(int(substeps)):
                state_0.clear_forces()
                d_s, d_ct = state_0._desc(), contacts._desc()
                leg.export_pointers(d_ct)
                _lib.check(dm.lib.nt_collide(C.byref(dm.desc), C.byref(d_s), C.byref(d_ct), C.byref(cp), dm.stream()), "nt_collide")
                leg.collide(state_0, contacts._flat, dm.stream())
                contacts._generation += 1
                self.step(state_0, state_1, control, contacts, dt)
                state_0, state_1 = state_1, state_0
            return state_0
        p = self._params()
        cp = _lib.nt_collide_params(0, self.envs_per_block)
        d0, d1, d_c, d_ct = state_0._desc(), state_1._desc(), control._desc(), contacts._desc()
        _lib.check(dm.lib.nt_featherstone_rollout(C.byref(dm.desc), C.byref(p), C.byref(cp), C.byref(d0), C.byref(d1),
                                                  C.byref(d_c), C.byref(d_ct), float(dt), int(substeps), dm.stream()),
                   "nt_featherstone_rollout")
        self._step += int(substeps)
        self._mass_matrix_dirty = False
        contacts._generation += 1
        return state_1 if substeps % 2 else state_0
