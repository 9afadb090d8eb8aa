"""newton_amd -- MI355X (gfx950) native batched rigid-body stepping behind Newton's API.

Facade mirrors newton/__init__.py:88-132 for the hot path only:
Model / State / Control / Contacts / ModelBuilder / CollisionPipeline / eval_fk / solvers.
"""
from . import builder as _builder_mod
from . import solvers
from . import geometry, graph, selection, utils, viewer
from .articulation import eval_fk
from .builder import JointDofConfig, ModelBuilder, ShapeConfig
from .collide import CollisionPipeline, ContactMatcher, Contacts
from .enums import BodyFlags, GeoType, JointType, ModelFlags, ShapeFlags, StateFlags
from .mesh import Heightfield, Mesh
from .model import Model
from .state import Control, State

__version__ = "0.1.0"


def __getattr__(name):
    if name == "use_coord_layout_targets":
        return _builder_mod.use_coord_layout_targets
    raise AttributeError(name)


def set_use_coord_layout_targets(value: bool):
    """newton.use_coord_layout_targets (newton/__init__.py:15-40)."""
    _builder_mod.use_coord_layout_targets = bool(value)


__all__ = ["BodyFlags", "CollisionPipeline", "ContactMatcher", "Contacts", "Control", "GeoType", "Heightfield", "JointDofConfig", "JointType", "Mesh", "Model",
           "ModelBuilder", "ModelFlags", "ShapeConfig", "ShapeFlags", "State", "StateFlags", "eval_fk", "geometry", "selection", "viewer",
           "solvers",
           "set_use_coord_layout_targets"]
