"""SolverBase (newton/_src/solvers/solver.py:190-450): same constructor / step / notify surface."""
from __future__ import annotations

from ..enums import ModelFlags


class SolverBase:
    def __new__(cls, model, *args, **kwargs):
        # worlds with different topologies: one solver per world group behind the same surface (hetero.py)
        if getattr(model, "is_heterogeneous", False):
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
