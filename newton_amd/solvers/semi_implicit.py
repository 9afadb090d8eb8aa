"""SolverSemiImplicit -- drop-in for newton.solvers.SolverSemiImplicit
(newton/_src/solvers/semi_implicit/solver_semi_implicit.py:73-217), rigid bodies only.

Penalty joints (``eval_body_joints``) + penalty contacts (``eval_body_contact``) + ``integrate_bodies`` run as ONE launch
of the gfx950 kernel ``semi_implicit_step_kernel`` through the C ABI ``nt_semi_implicit_step``.
Difference to the reference, documented: ``state_in.body_f`` is never modified (the reference accumulates contact
forces into it when the model has no joints, solver_semi_implicit.py:160-163).
"""
from __future__ import annotations

import ctypes as C

from .. import _lib
from .solver import SolverBase


class SolverSemiImplicit(SolverBase):
    def __init__(self, model, *, angular_damping: float = 0.05, friction_smoothing: float = 1.0, joint_attach_ke: float = 1.0e4,
                 joint_attach_kd: float = 1.0e2, enable_tri_contact: bool = True, envs_per_block: int = 0):
        super().__init__(model)
        self.dm.require_fit("SolverSemiImplicit")
        self.angular_damping = angular_damping
        self.friction_smoothing = friction_smoothing
        self.joint_attach_ke = joint_attach_ke
        self.joint_attach_kd = joint_attach_kd
        self.enable_tri_contact = enable_tri_contact
        self.envs_per_block = int(envs_per_block)

    def step(self, state_in, state_out, control, contacts, dt: float) -> None:
        dm = self.dm
        if control is None:
            if not hasattr(self, "_control"):
                self._control = self.model.control()
            control = self._control
        p = _lib.nt_semi_implicit_params(float(self.angular_damping), float(self.friction_smoothing),
                                         float(self.joint_attach_ke), float(self.joint_attach_kd))
        d_in, d_out, d_c = state_in._desc(), state_out._desc(), control._desc()
        d_in = self._state_desc_with_sdf_forces(state_in, contacts, self.friction_smoothing)
        d_ct = contacts._desc() if contacts is not None else None
        _lib.check(dm.lib.nt_semi_implicit_step(C.byref(dm.desc), C.byref(p), C.byref(d_in), C.byref(d_out), C.byref(d_c),
                                                C.byref(d_ct) if d_ct is not None else None, float(dt),
                                                self.envs_per_block, dm.stream()), "nt_semi_implicit_step")
