"""Enumerations shared with the reference (values must match bit-for-bit).

Reference: newton/_src/sim/enums.py:16-37,72-87,136-142,183-212; newton/_src/geometry/types.py:78-111;
newton/_src/geometry/flags.py:32-44.
"""
from enum import IntEnum, IntFlag


class JointType(IntEnum):
    PRISMATIC = 0
    REVOLUTE = 1
    BALL = 2
    FIXED = 3
    FREE = 4
    DISTANCE = 5
    D6 = 6
    CABLE = 7
    ROD = 7

    def dof_count(self, num_axes: int):
        """(dofs, coords) of a joint of this type; newton/_src/sim/enums.py JointType.dof_count."""
        return {
            JointType.PRISMATIC: (1, 1),
            JointType.REVOLUTE: (1, 1),
            JointType.BALL: (3, 4),
            JointType.FIXED: (0, 0),
            JointType.FREE: (6, 7),
            JointType.DISTANCE: (6, 7),
            JointType.D6: (num_axes, num_axes),
        }[JointType(int(self))]


class GeoType(IntEnum):
    NONE = 0
    PLANE = 1
    HFIELD = 2
    SPHERE = 3
    CAPSULE = 4
    ELLIPSOID = 5
    CYLINDER = 6
    BOX = 7
    MESH = 8
    CONE = 9
    CONVEX_MESH = 10
    GAUSSIAN = 11


class BodyFlags(IntFlag):
    DYNAMIC = 1 << 0
    KINEMATIC = 1 << 1
    PROXY = 1 << 2


class ShapeFlags(IntFlag):
    VISIBLE = 1 << 0
    COLLIDE_SHAPES = 1 << 1
    COLLIDE_PARTICLES = 1 << 2
    SITE = 1 << 3
    HYDROELASTIC = 1 << 4


class ModelFlags(IntFlag):
    JOINT_PROPERTIES = 1 << 0
    JOINT_DOF_PROPERTIES = 1 << 1
    BODY_PROPERTIES = 1 << 2
    BODY_INERTIAL_PROPERTIES = 1 << 3
    SHAPE_PROPERTIES = 1 << 4
    MODEL_PROPERTIES = 1 << 5
    CONSTRAINT_PROPERTIES = 1 << 6
    TENDON_PROPERTIES = 1 << 7
    ACTUATOR_PROPERTIES = 1 << 8


class StateFlags(IntFlag):
    NONE = 0
    JOINT_Q = 1 << 0
    JOINT_QD = 1 << 1
    BODY_Q = 1 << 2
    BODY_QD = 1 << 3
    PARTICLE_Q = 1 << 4
    PARTICLE_QD = 1 << 5


MAXVAL = 1e10  # newton/_src/core/types.py:71-72
