"""SolverBase (newton/_src/solvers/solver.py:190-450): same constructor / step / notify surface."""
from __future__ import annotations

from ..enums import ModelFlags


class SolverBase:
    def __new__(cls, model=None, *args, **kwargs):
        # worlds with different topologies: one solver per world group behind the same surface (hetero.py).  `model` defaults to
        # None so that copy / deepcopy / pickle, which call cls.__new__(cls) without arguments, keep working
        if model is not None and getattr(model, "is_heterogeneous", False):
            from ..hetero import GroupedSolver  # noqa: PLC0415

            return GroupedSolver(cls, model, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, model):
        self.model = model
        self.dm = model.device_model()  # raises loudly when the HIP extension / GPU is missing

    @property
    def device(self):
        return self.model.device

    def step(self, state_in, state_out, control, contacts, dt):
        raise NotImplementedError()

    def _state_desc_with_sdf_forces(self, state_in, contacts, friction_smoothing):
        """Penalty solvers (SemiImplicit / Featherstone): nt_state of `state_in` whose body_f additionally carries the penalty
        wrenches of the SDF leg's contact rows (eval_body_contact, kernels_contact.py:381-556, summed per body in row order).
        state_in.body_f itself is not modified: the sum goes into a solver-owned copy the step kernel reads instead."""
        d_in = state_in._desc()
        flat = getattr(contacts, "_flat", None) if contacts is not None else None
        if flat is None:
            return d_in
        import torch  # noqa: PLC0415

        buf = getattr(self, "_sdf_body_f", None)
        if buf is None:
            buf = self._sdf_body_f = torch.empty_like(state_in._soa["body_f"])
        buf.copy_(state_in._soa["body_f"])
        contacts._sdf_leg.add_forces(state_in, flat, buf, friction_smoothing, self.dm.stream())
        d_in.body_f = buf.data_ptr()
        return d_in

    def notify_model_changed(self, flags):
        """solver.py:394-440.  Per-env parameter arrays are re-uploaded from the host model."""
        if int(flags) & int(ModelFlags.BODY_PROPERTIES | ModelFlags.BODY_INERTIAL_PROPERTIES |
                            ModelFlags.JOINT_PROPERTIES | ModelFlags.JOINT_DOF_PROPERTIES |
                            ModelFlags.SHAPE_PROPERTIES | ModelFlags.MODEL_PROPERTIES):
            self.model.notify_model_changed()

    def update_contacts(self, contacts, state=None):
        raise NotImplementedError()

    def reset(self, state, world_mask=None, flags=None):
        """solver.py:344-375: default is a no-op."""
        return None
