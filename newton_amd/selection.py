"""newton.selection (newton/selection.py): ArticulationView."""
from .utils.selection import ArticulationView

__all__ = ["ArticulationView"]
