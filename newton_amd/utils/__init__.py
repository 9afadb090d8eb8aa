"""newton_amd.utils -- host-side helpers around the hot path (newton/_src/utils)."""
from .selection import ArticulationView

__all__ = ["ArticulationView"]
