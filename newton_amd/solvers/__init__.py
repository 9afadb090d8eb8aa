"""newton_amd.solvers -- MI355X-native drop-ins for newton.solvers (newton/_src/solvers/__init__.py:35-58)."""
from .featherstone import SolverFeatherstone
from .semi_implicit import SolverSemiImplicit
from .solver import SolverBase
from .xpbd import SolverXPBD

__all__ = ["SolverBase", "SolverFeatherstone", "SolverSemiImplicit", "SolverXPBD"]
