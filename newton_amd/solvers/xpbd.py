"""SolverXPBD -- drop-in for newton.solvers.SolverXPBD (newton/_src/solvers/xpbd/solver_xpbd.py:100-862), rigid bodies.

``step`` is one launch of the gfx950 kernel ``xpbd_step_kernel`` through the C ABI ``nt_xpbd_step``;
``rollout`` is the fused replacement for the CUDA-graph-captured substep loop
(newton/examples/basic/example_basic_urdf.py:117-141) through ``nt_xpbd_rollout``.
"""
from __future__ import annotations

import ctypes as C

from .. import _lib
from .solver import SolverBase


class SolverXPBD(SolverBase):
    def __init__(self, model, *, iterations: int = 2, soft_body_relaxation: float = 0.9, soft_contact_relaxation: float = 0.9,
                 joint_linear_relaxation: float = 0.7, joint_angular_relaxation: float = 0.4,
                 joint_linear_compliance: float = 0.0, joint_angular_compliance: float = 0.0,
                 rigid_contact_relaxation: float = 0.8, rigid_contact_con_weighting: bool = True,
                 angular_damping: float = 0.0, enable_restitution: bool = False, deterministic=None,
                 envs_per_block: int = 0):
        super().__init__(model)
        self.dm.require_fit("SolverXPBD")
        self.iterations = iterations
        self.soft_body_relaxation = soft_body_relaxation
        self.soft_contact_relaxation = soft_contact_relaxation
        self.joint_linear_relaxation = joint_linear_relaxation
        self.joint_angular_relaxation = joint_angular_relaxation
        self.joint_linear_compliance = joint_linear_compliance
        self.joint_angular_compliance = joint_angular_compliance
        self.rigid_contact_relaxation = rigid_contact_relaxation
        self.rigid_contact_con_weighting = rigid_contact_con_weighting
        self.angular_damping = angular_damping
        self.enable_restitution = enable_restitution
        self.compute_body_velocity_from_position_delta = False  # reference attribute (solver_xpbd.py:767): set after construction
        self.envs_per_block = int(envs_per_block)
        self._contact_impulse = None
        self._contact_impulse_capacity = 0
        self._last_dt = None
        self._joint_impulse = None

    def _params(self) -> _lib.nt_xpbd_params:
        return _lib.nt_xpbd_params(int(self.iterations), float(self.joint_linear_relaxation),
                                   float(self.joint_angular_relaxation), float(self.joint_linear_compliance),
                                   float(self.joint_angular_compliance), float(self.rigid_contact_relaxation),
                                   int(bool(self.rigid_contact_con_weighting)), float(self.angular_damping),
                                   int(bool(self.enable_restitution)),
                                   int(bool(self.compute_body_velocity_from_position_delta)))

    def step(self, state_in, state_out, control, contacts, dt: float) -> None:
        dm = self.dm
        if control is None:
            control = self._default_control()
        p = self._params()
        d_in, d_out, d_c = state_in._desc(), state_out._desc(), control._desc()
        if contacts is not None and getattr(contacts, "_flat", None) is not None and self.enable_restitution:
            # the SDF leg's rows take part in every pass: the position solve, Contacts.force (nt_flat_rows.impulse), the body-level
            # velocity update and the restitution pass (its per-row records: nt_flat_rows.restitution)
            contacts._flat.restitution_scratch()
        d_ct = contacts._desc() if contacts is not None else None
        # optional reporting (solver_xpbd.py:368-386): per-contact impulses when contacts.force was requested, per-joint
        # impulses when state_out carries body_parent_f
        rep = _lib.nt_xpbd_report()
        reporting = False
        self._contact_impulse = None
        if contacts is not None and contacts.force is not None:
            rep.contact_impulse = contacts._impulse.data_ptr()
            self._contact_impulse = contacts._impulse
            reporting = True
        self._contact_impulse_capacity = contacts.rigid_contact_max if contacts is not None else 0
        self._last_dt = float(dt)
        if state_out._parent_f is not None and self.model.env.nj > 0:
            if self._joint_impulse is None:
                import torch  # noqa: PLC0415

                t = self.model.env
                self._joint_impulse = torch.zeros((6, t.nj, t.env_stride), dtype=torch.float32, device=dm.device)
            rep.joint_impulse = self._joint_impulse.data_ptr()
            reporting = True
        _lib.check(dm.lib.nt_xpbd_step(C.byref(dm.desc), C.byref(p), C.byref(d_in), C.byref(d_out), C.byref(d_c),
                                       C.byref(d_ct) if d_ct is not None else None, float(dt), self.envs_per_block,
                                       C.byref(rep) if reporting else None, dm.stream()), "nt_xpbd_step")

    def update_contacts(self, contacts, state=None) -> None:
        """Fill ``contacts.force`` from the impulses of the last ``step`` (solver_xpbd.py:864-921)."""
        if contacts.force is None:
            raise ValueError("contacts.force is not allocated. Call model.request_contact_attributes('force') before "
                             "creating the Contacts object.")
        if self._contact_impulse is None:
            raise ValueError("No contact impulse data available. Call step() before update_contacts().")
        if contacts.rigid_contact_max != self._contact_impulse_capacity or contacts._impulse is not self._contact_impulse:
            raise ValueError("Contacts mismatch: pass the same Contacts instance to both step() and update_contacts().")
        dm = self.dm
        d_ct = contacts._desc()
        _lib.check(dm.lib.nt_contacts_export_force(C.byref(dm.desc), C.byref(d_ct), self._contact_impulse.data_ptr(),
                                                   float(self._last_dt), int(contacts.rigid_contact_max),
                                                   contacts.force.data_ptr(), contacts._scan.data_ptr(), dm.stream()),
                   "nt_contacts_export_force")
        # the rows of the SDF legs follow the slot contacts in the flat arrays (collide.py:1999): force = impulse / dt
        # (convert_contact_impulse_to_force, xpbd/kernels.py:2464-2494)
        f = getattr(contacts, "_flat", None)
        if f is not None and f.impulse is not None:
            contacts._exported()
            live, n0 = contacts._flat_live, contacts._flat_n0
            k = min(int(live.numel()), int(contacts.force.shape[0]) - n0)
            if k > 0:
                contacts.force[n0:n0 + k] = f.impulse[live[:k]] * (1.0 / float(self._last_dt))
        # CollisionPipeline(deterministic=True) re-orders the rigid_contact_* rows by the contact key: force[i] must stay the
        # force of rigid_contact_shape0/1[i] (the reference sorts before the solver runs, so its indices agree by construction)
        order = contacts.export_order()
        if order is not None:
            n = order.numel()
            contacts.force[:n] = contacts.force[:n][order]

    def rollout(self, state_0, state_1, control, contacts, dt: float, substeps: int, collide_params=None):
        """substeps x {clear_forces; collide; step; swap} in one launch.  Returns the state object holding the
        result (state_0 for an even number of substeps, state_1 for odd -- the reference loop's swap)."""
        dm = self.dm
        if control is None:
            control = self._default_control()
        cp = collide_params if collide_params is not None else _lib.nt_collide_params(0, self.envs_per_block)
        leg = getattr(contacts, "_sdf_leg", None)
        if contacts.force is not None or state_0._parent_f is not None or state_1._parent_f is not None or leg is not None:
            # the reporting outputs only exist in the per-substep kernel, and the SDF leg of collide() is a chain of launches
            # of its own: run the reference loop launch by launch
            for _ in range(int(substeps)):
                state_0.clear_forces()
                d_s, d_ct = state_0._desc(), contacts._desc()
                if leg is not None:
                    leg.export_pointers(d_ct)
                _lib.check(dm.lib.nt_collide(C.byref(dm.desc), C.byref(d_s), C.byref(d_ct), C.byref(cp), dm.stream()),
                           "nt_collide")
                if leg is not None:
                    leg.collide(state_0, contacts._flat, dm.stream())
                contacts._generation += 1
                self.step(state_0, state_1, control, contacts, dt)
                state_0, state_1 = state_1, state_0
            return state_0
        p = self._params()
        contacts.prepare_rollout()  # (pair-heavy scenes: the rollout's own contact records, allocated on first use)
        d0, d1, d_c, d_ct = state_0._desc(), state_1._desc(), control._desc(), contacts._desc()
        _lib.check(dm.lib.nt_xpbd_rollout(C.byref(dm.desc), C.byref(p), C.byref(cp), C.byref(d0), C.byref(d1), C.byref(d_c),
                                          C.byref(d_ct), float(dt), int(substeps), dm.stream()), "nt_xpbd_rollout")
        contacts._generation += 1
        return state_1 if substeps % 2 else state_0

    def _default_control(self):
        if not hasattr(self, "_control"):
            self._control = self.model.control()
        return self._control
