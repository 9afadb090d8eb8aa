"""Minimal ModelBuilder: the host-side subset needed to feed the hot path with real scenes.

This is NOT a port of newton/_src/sim/builder.py (13 kLoC, out of scope -- SURVEY.md section 2.1); it restates
only the rules that decide what the flat Model arrays contain for rigid scenes:

* body / joint / shape bookkeeping and defaults          builder.py:491-590 (ShapeConfig), 762-825 (JointDofConfig)
* add_link / add_body                                    builder.py:4340-4490
* add_joint + typed helpers                              builder.py:4493-5125
* add_shape + mass accumulation                          builder.py:6498-6716, 9885-9917
* add_ground_plane                                       builder.py:6718-6812
* replicate (one world per copy)                         builder.py:2599-2659
* finalize: flat arrays, world starts, contact pairs     builder.py:11032-11232, 12816-13060
* collision filtering rules                              builder.py:1722-1737, 6638-6700; broad_phase_common.py:220-268
"""
from __future__ import annotations

import copy
from dataclasses import dataclass

import numpy as np

from . import _np_math as nm
from .enums import MAXVAL, BodyFlags, GeoType, JointType, ShapeFlags
from .inertia import compute_inertia_shape, compute_shape_radius, transform_inertia
from .mesh import Mesh, deduplicate_vertices  # noqa: F401

# module flag mirrored from newton/__init__.py:15 (coord-shaped joint_target_q when True)
use_coord_layout_targets = False


def _axis_vec(axis):
    if isinstance(axis, (int, np.integer)):
        v = np.zeros(3)
        v[int(axis)] = 1.0
        return v
    if isinstance(axis, str):
        v = np.zeros(3)
        v["xyz".index(axis.lower())] = 1.0
        return v
    v = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(v)
    return v / n if n > 0 else v


def _clone(x):
    """Cheap deep copy of the builder's per-entity records (numbers, tuples, nested lists, numpy arrays)."""
    if isinstance(x, np.ndarray):
        return x.copy()
    if isinstance(x, list):
        return [_clone(e) for e in x]
    if isinstance(x, (int, float, str, tuple, bool, type(None), np.generic)):
        return x
    return copy.deepcopy(x)


def _box_edge_tables(half_extents):
    """(edge_centers, edge_halves) of the 12 edges of a box with the given half extents: what the reference gets from
    Mesh.create_box(1, 1, 1)._build_collision_edges(...) with shape_scale applied (builder.py:11389-11412,12088-12116); the
    first edge that touches a corner owns it."""
    h = np.asarray(half_extents, dtype=np.float32)
    corners = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float32) * h
    edges = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]
    centers, halves, seen = [], [], set()
    for a, b in edges:
        v0, v1 = corners[a], corners[b]
        half = (v1 - v0) * np.float32(0.5)
        own = 4.0 + (a not in seen) + 2.0 * (b not in seen)
        seen.update((a, b))
        centers.append([*((v0 + v1) * np.float32(0.5)), np.linalg.norm(half)])
        halves.append([*half, own])
    return np.asarray(centers, np.float32), np.asarray(halves, np.float32)


@dataclass
class ShapeConfig:
    """Per-shape collision / material settings (builder.py:491-590 defaults)."""

    density: float = 1000.0
    ke: float = 2.5e3
    kd: float = 100.0
    kf: float = 1000.0
    ka: float = 0.0
    mu: float = 1.0
    restitution: float = 0.0
    mu_torsional: float = 0.005
    mu_rolling: float = 0.0001
    margin: float = 0.0
    gap: float | None = None
    is_solid: bool = True
    collision_group: int = 1
    collision_filter_parent: bool = True
    has_shape_collision: bool = True
    has_particle_collision: bool = True
    is_visible: bool = True
    kh: float = 1.0e10
    # texture-SDF / hydroelastic options (builder.py:545-592).  Primitive shapes (BOX) get a generated SDF when a resolution or a
    # voxel size is set; mesh-backed shapes use the SDF attached with Mesh.build_sdf()
    sdf_narrow_band_range: tuple = (-0.1, 0.1)
    sdf_target_voxel_size: float | None = None
    sdf_max_resolution: int | None = None
    force_sdf: bool = False
    sdf_texture_format: str = "uint16"
    is_hydroelastic: bool = False
    sdf_padding: float | None = None

    def configure_sdf(self, *, max_resolution=None, target_voxel_size=None, is_hydroelastic: bool = False, kh: float = 1.0e10,
                      texture_format=None, force_sdf: bool = False):
        """SDF and hydroelastic options in one place (builder.py:594-642)."""
        if max_resolution is not None and target_voxel_size is not None:
            raise ValueError("configure_sdf accepts either max_resolution or target_voxel_size, not both.")
        self.force_sdf = force_sdf
        if max_resolution is not None:
            self.sdf_max_resolution, self.sdf_target_voxel_size = max_resolution, None
        if target_voxel_size is not None:
            self.sdf_target_voxel_size, self.sdf_max_resolution = target_voxel_size, None
        self.is_hydroelastic = is_hydroelastic
        self.kh = kh
        if texture_format is not None:
            self.sdf_texture_format = texture_format

    def validate(self, shape_type=None):
        """builder.py:644-710 (the checks that concern the options carried here)."""
        if self.sdf_texture_format not in ("float32", "uint16", "uint8"):
            raise ValueError(f"Unknown sdf_texture_format {self.sdf_texture_format!r}. Expected one of ['float32', 'uint16', 'uint8'].")
        if self.sdf_target_voxel_size is not None and not (np.isfinite(self.sdf_target_voxel_size) and self.sdf_target_voxel_size > 0.0):
            raise ValueError(f"sdf_target_voxel_size must be finite and > 0 (got {self.sdf_target_voxel_size}).")
        if self.sdf_padding is not None and not (np.isfinite(self.sdf_padding) and self.sdf_padding >= 0.0):
            raise ValueError(f"sdf_padding must be finite and >= 0 (got {self.sdf_padding}).")
        inner, outer = self.sdf_narrow_band_range
        if not (np.isfinite(inner) and np.isfinite(outer) and inner < 0.0 < outer):
            raise ValueError("sdf_narrow_band_range must contain finite values satisfying inner < 0 < outer "
                             f"(got {self.sdf_narrow_band_range}).")
        if self.sdf_max_resolution is not None and self.sdf_target_voxel_size is not None:
            raise ValueError("Set only one of sdf_max_resolution or sdf_target_voxel_size, not both.")
        if self.sdf_max_resolution is not None:
            if self.sdf_max_resolution <= 0:
                raise ValueError(f"sdf_max_resolution must be > 0 (got {self.sdf_max_resolution}).")
            if self.sdf_max_resolution % 8 != 0:
                raise ValueError(f"sdf_max_resolution must be divisible by 8 (got {self.sdf_max_resolution}).")

    @property
    def flags(self) -> int:
        f = 0
        if self.is_visible:
            f |= ShapeFlags.VISIBLE
        if self.has_shape_collision:
            f |= ShapeFlags.COLLIDE_SHAPES
        if self.has_particle_collision:
            f |= ShapeFlags.COLLIDE_PARTICLES
        if self.is_hydroelastic:
            f |= ShapeFlags.HYDROELASTIC
        return int(f)

    def copy(self):
        return copy.copy(self)


class JointDofConfig:
    """One joint axis (builder.py:762-825)."""

    def __init__(self, *, axis=0, limit_lower=-MAXVAL, limit_upper=MAXVAL, limit_ke=1e4, limit_kd=1e1, target_pos=0.0,
                 target_vel=0.0, target_ke=0.0, target_kd=0.0, damping=0.0, armature=0.0, effort_limit=1e6,
                 velocity_limit=1e6, friction=0.0):
        self.axis = _axis_vec(axis)
        self.limit_lower = limit_lower
        self.limit_upper = limit_upper
        self.limit_ke = limit_ke
        self.limit_kd = limit_kd
        self.target_pos = target_pos
        self.target_vel = target_vel
        self.target_ke = target_ke
        self.target_kd = target_kd
        self.damping = damping
        self.armature = armature
        self.effort_limit = effort_limit
        self.velocity_limit = velocity_limit
        self.friction = friction
        if self.target_pos > self.limit_upper or self.target_pos < self.limit_lower:
            self.target_pos = 0.5 * (self.limit_lower + self.limit_upper)

    @classmethod
    def create_unlimited(cls, axis):
        return cls(axis=axis, limit_lower=-MAXVAL, limit_upper=MAXVAL)

    def copy(self):
        return copy.copy(self)


class ModelBuilder:
    ShapeConfig = ShapeConfig
    JointDofConfig = JointDofConfig

    def __init__(self, up_axis: int = 2, gravity=-9.81):
        """``gravity``: a 3-vector [m/s^2] like the reference (builder.py gravity=(gx, gy, gz)), or a scalar along ``up_axis``."""
        self.up_axis = up_axis
        if np.ndim(gravity) == 0:
            self._gravity_scalar, self._gravity_vec = float(gravity), None
        else:
            g = np.asarray(gravity, dtype=np.float64)
            if g.shape != (3,):
                raise ValueError(f"gravity must be a scalar or have shape (3,), got {g.shape}")
            self._gravity_scalar, self._gravity_vec = float(g[up_axis]), g
        self.default_shape_cfg = ShapeConfig()
        self.default_joint_cfg = JointDofConfig()
        self._requested_state_attributes: set[str] = set()
        self._requested_contact_attributes: set[str] = set()
        self.rigid_gap = 0.1  # builder.py:1596
        self.current_world = -1
        self.world_count = 0
        self.world_gravity: list = []

        # bodies
        self.body_q, self.body_qd, self.body_mass, self.body_inertia = [], [], [], []
        self.body_inv_mass, self.body_inv_inertia, self.body_com = [], [], []
        self.body_flags, self.body_world, self.body_label, self.body_lock_inertia = [], [], [], []
        self.body_shapes: dict[int, list[int]] = {-1: []}
        # joints
        self.joint_type, self.joint_parent, self.joint_child = [], [], []
        self.joint_X_p, self.joint_X_c, self.joint_label = [], [], []
        self.joint_dof_dim, self.joint_enabled, self.joint_world, self.joint_articulation = [], [], [], []
        self.joint_collision_filter_parent = []
        self.joint_q_start, self.joint_qd_start, self.joint_target_q_start = [], [], []
        self.joint_q, self.joint_qd, self.joint_f, self.joint_target_q, self.joint_target_qd = [], [], [], [], []
        self.joint_axis, self.joint_limit_lower, self.joint_limit_upper = [], [], []
        self.joint_limit_ke, self.joint_limit_kd, self.joint_target_ke, self.joint_target_kd = [], [], [], []
        self.joint_damping, self.joint_armature, self.joint_effort_limit, self.joint_velocity_limit = [], [], [], []
        self.joint_friction = []
        self.joint_parents: dict[int, list] = {}
        self.joint_children: dict[int, list] = {}
        # articulations
        self.articulation_start, self.articulation_world, self.articulation_label = [], [], []
        # shapes
        self.shape_body, self.shape_type, self.shape_scale, self.shape_transform = [], [], [], []
        self.shape_flags, self.shape_margin, self.shape_gap, self.shape_world, self.shape_label = [], [], [], [], []
        self.shape_collision_group, self.shape_collision_radius = [], []
        self.shape_source: list = []  # Mesh for CONVEX_MESH shapes, else None (shared, never copied)
        # per-shape SDF options retained until finalize() (builder.py:1322-1333)
        self.shape_sdf_narrow_band_range, self.shape_sdf_target_voxel_size, self.shape_sdf_max_resolution = [], [], []
        self.shape_sdf_texture_format, self.shape_sdf_padding = [], []
        self.shape_material_ke, self.shape_material_kd, self.shape_material_kf, self.shape_material_ka = [], [], [], []
        self.shape_material_mu, self.shape_material_restitution = [], []
        self.shape_material_mu_torsional, self.shape_material_mu_rolling, self.shape_material_kh = [], [], []
        self.shape_collision_filter_pairs: set[tuple[int, int]] = set()

    # ------------------------------------------------------------------ properties
    @property
    def body_count(self):
        return len(self.body_mass)

    @property
    def joint_count(self):
        return len(self.joint_type)

    @property
    def shape_count(self):
        return len(self.shape_type)

    @property
    def joint_dof_count(self):
        return len(self.joint_qd)

    @property
    def joint_coord_count(self):
        return len(self.joint_q)

    @property
    def articulation_count(self):
        return len(self.articulation_start)

    @property
    def up_vector(self):
        v = [0.0, 0.0, 0.0]
        v[self.up_axis] = 1.0
        return tuple(v)

    def _gravity_vector(self):
        if self._gravity_vec is not None:
            return self._gravity_vec.copy()
        return np.asarray(self.up_vector) * self._gravity_scalar

    # ------------------------------------------------------------------ worlds
    def begin_world(self):
        if self.current_world != -1:
            raise RuntimeError("already in a world context; call end_world() first")
        self.current_world = self.world_count
        self.world_gravity.append(self._gravity_vector())

    def end_world(self):
        if self.current_world == -1:
            raise RuntimeError("not in a world context")
        self.world_count += 1
        self.current_world = -1

    def add_shape_collision_filter_pair(self, a: int, b: int):
        self.shape_collision_filter_pairs.add((min(a, b), max(a, b)))

    # ------------------------------------------------------------------ bodies
    def add_link(self, *, xform=None, com=None, inertia=None, mass=0.0, label=None, lock_inertia=False,
                 is_kinematic=False) -> int:
        xform = nm.transform() if xform is None else np.asarray(xform, dtype=np.float64)
        com = np.zeros(3) if com is None else np.asarray(com, dtype=np.float64)
        inertia = np.zeros((3, 3)) if inertia is None else np.asarray(inertia, dtype=np.float64).reshape(3, 3)
        body_id = self.body_count
        self.body_inertia.append(inertia)
        self.body_mass.append(float(mass))
        self.body_com.append(com)
        self.body_lock_inertia.append(lock_inertia)
        self.body_flags.append(int(BodyFlags.KINEMATIC) if is_kinematic else int(BodyFlags.DYNAMIC))
        self.body_inv_mass.append(1.0 / mass if mass > 0.0 else 0.0)
        self.body_inv_inertia.append(np.linalg.inv(inertia) if inertia.any() else inertia.copy())
        self.body_q.append(xform)
        self.body_qd.append(np.zeros(6))
        self.body_label.append(label or f"body_{body_id}")
        self.body_shapes[body_id] = []
        self.body_world.append(self.current_world)
        return body_id

    def add_body(self, *, xform=None, com=None, inertia=None, mass=0.0, label=None, lock_inertia=False,
                 is_kinematic=False) -> int:
        """Free-floating body = link + FREE joint + single-joint articulation (builder.py:4426-4490)."""
        b = self.add_link(xform=xform, com=com, inertia=inertia, mass=mass, label=label, lock_inertia=lock_inertia,
                          is_kinematic=is_kinematic)
        j = self.add_joint_free(child=b, label=f"{label}_free_joint" if label else None)
        self.add_articulation([j])
        return b

    # ------------------------------------------------------------------ joints
    @staticmethod
    def _default_filter_parent(joint_type, parent):
        # builder.py:1732-1737: non-fixed joints to world do not filter
        return not (parent == -1 and joint_type != JointType.FIXED)

    def add_joint(self, joint_type, parent, child, *, linear_axes=None, angular_axes=None, label=None, parent_xform=None,
                  child_xform=None, collision_filter_parent=None, enabled=True) -> int:
        linear_axes = linear_axes or []
        angular_axes = angular_axes or []
        if collision_filter_parent is None:
            collision_filter_parent = self._default_filter_parent(joint_type, parent)
        parent_xform = nm.transform() if parent_xform is None else np.asarray(parent_xform, dtype=np.float64)
        child_xform = nm.transform() if child_xform is None else np.asarray(child_xform, dtype=np.float64)
        if child < 0 or child >= self.body_count:
            raise ValueError(f"Child body index {child} is out of range")
        if parent != -1 and (parent < 0 or parent >= self.body_count):
            raise ValueError(f"Parent body index {parent} is out of range")

        self.joint_type.append(int(joint_type))
        joint_idx = self.joint_count - 1
        self.joint_parent.append(parent)
        self.joint_parents.setdefault(child, []).append((parent, joint_idx))
        self.joint_children.setdefault(parent, []).append((child, joint_idx))
        self.joint_child.append(child)
        self.joint_X_p.append(parent_xform)
        self.joint_X_c.append(child_xform)
        self.joint_label.append(label or f"joint_{self.joint_count}")
        self.joint_dof_dim.append((len(linear_axes), len(angular_axes)))
        self.joint_enabled.append(bool(enabled))
        self.joint_collision_filter_parent.append(collision_filter_parent)
        self.joint_world.append(self.current_world)
        self.joint_articulation.append(-1)

        for dim in (*linear_axes, *angular_axes):
            self.joint_axis.append(dim.axis)
            self.joint_target_qd.append(dim.target_vel)
            self.joint_target_ke.append(dim.target_ke)
            self.joint_target_kd.append(dim.target_kd)
            self.joint_damping.append(dim.damping)
            self.joint_limit_ke.append(dim.limit_ke)
            self.joint_limit_kd.append(dim.limit_kd)
            self.joint_armature.append(dim.armature)
            self.joint_effort_limit.append(dim.effort_limit)
            self.joint_velocity_limit.append(dim.velocity_limit)
            self.joint_friction.append(dim.friction)
            self.joint_limit_lower.append(dim.limit_lower if np.isfinite(dim.limit_lower) else -MAXVAL)
            self.joint_limit_upper.append(dim.limit_upper if np.isfinite(dim.limit_upper) else MAXVAL)

        jt = JointType(int(joint_type))
        dof_count, coord_count = jt.dof_count(len(linear_axes) + len(angular_axes))
        q_start, qd_start = len(self.joint_q), len(self.joint_qd)
        target_q_offset = len(self.joint_target_q)
        tq_count = coord_count if use_coord_layout_targets else dof_count
        self.joint_q.extend([0.0] * coord_count)
        self.joint_target_q.extend([0.0] * tq_count)
        self.joint_qd.extend([0.0] * dof_count)
        self.joint_f.extend([0.0] * dof_count)
        if jt in (JointType.FREE, JointType.DISTANCE, JointType.BALL):
            self.joint_q[-1] = 1.0
            if use_coord_layout_targets:
                self.joint_target_q[-1] = 1.0
        if jt not in (JointType.FREE, JointType.DISTANCE, JointType.BALL, JointType.FIXED):
            for i, dim in enumerate((*linear_axes, *angular_axes)):
                self.joint_target_q[target_q_offset + i] = dim.target_pos
        self.joint_q_start.append(q_start)
        self.joint_qd_start.append(qd_start)
        self.joint_target_q_start.append(target_q_offset)

        if collision_filter_parent:
            for child_shape in self.body_shapes[child]:
                if not self.shape_flags[child_shape] & ShapeFlags.COLLIDE_SHAPES:
                    continue
                for parent_shape in self.body_shapes[parent]:
                    if not self.shape_flags[parent_shape] & ShapeFlags.COLLIDE_SHAPES:
                        continue
                    self.add_shape_collision_filter_pair(parent_shape, child_shape)
        return joint_idx

    def _dof(self, axis, **overrides):
        d = self.default_joint_cfg
        kw = dict(axis=axis, limit_lower=d.limit_lower, limit_upper=d.limit_upper, limit_ke=d.limit_ke,
                  limit_kd=d.limit_kd, target_pos=d.target_pos, target_vel=d.target_vel, target_ke=d.target_ke,
                  target_kd=d.target_kd, damping=d.damping, armature=d.armature, effort_limit=d.effort_limit,
                  velocity_limit=d.velocity_limit, friction=d.friction)
        for k, v in overrides.items():
            if v is not None:
                kw[k] = v
        return JointDofConfig(**kw)

    def add_joint_revolute(self, parent, child, *, parent_xform=None, child_xform=None, axis=None, label=None,
                           collision_filter_parent=None, enabled=True, **dof_kwargs) -> int:
        ax = axis if isinstance(axis, JointDofConfig) else self._dof(0 if axis is None else axis, **dof_kwargs)
        return self.add_joint(JointType.REVOLUTE, parent, child, parent_xform=parent_xform, child_xform=child_xform,
                              angular_axes=[ax], label=label, collision_filter_parent=collision_filter_parent,
                              enabled=enabled)

    def add_joint_prismatic(self, parent, child, *, parent_xform=None, child_xform=None, axis=None, label=None,
                            collision_filter_parent=None, enabled=True, **dof_kwargs) -> int:
        ax = axis if isinstance(axis, JointDofConfig) else self._dof(0 if axis is None else axis, **dof_kwargs)
        return self.add_joint(JointType.PRISMATIC, parent, child, parent_xform=parent_xform, child_xform=child_xform,
                              linear_axes=[ax], label=label, collision_filter_parent=collision_filter_parent,
                              enabled=enabled)

    def add_joint_ball(self, parent, child, *, parent_xform=None, child_xform=None, label=None,
                       collision_filter_parent=None, enabled=True) -> int:
        return self.add_joint(JointType.BALL, parent, child, parent_xform=parent_xform, child_xform=child_xform,
                              angular_axes=[self._dof(0), self._dof(1), self._dof(2)], label=label,
                              collision_filter_parent=collision_filter_parent, enabled=enabled)

    def add_joint_fixed(self, parent, child, *, parent_xform=None, child_xform=None, label=None,
                        collision_filter_parent=None, enabled=True) -> int:
        return self.add_joint(JointType.FIXED, parent, child, parent_xform=parent_xform, child_xform=child_xform,
                              label=label, collision_filter_parent=collision_filter_parent, enabled=enabled)

    def add_joint_distance(self, parent, child, *, parent_xform=None, child_xform=None, min_distance=-1.0, max_distance=1.0,
                           label=None, collision_filter_parent=None, enabled=True) -> int:
        """Distance joint (builder.py:5128-5187): anchor distance kept in [min, max] (a negative bound is inactive); only
        SolverXPBD supports it."""
        ax = JointDofConfig(axis=0, limit_lower=min_distance, limit_upper=max_distance)
        return self.add_joint(JointType.DISTANCE, parent, child, parent_xform=parent_xform, child_xform=child_xform, label=label,
                              linear_axes=[ax, JointDofConfig.create_unlimited(1), JointDofConfig.create_unlimited(2)],
                              angular_axes=[JointDofConfig.create_unlimited(a) for a in range(3)],
                              collision_filter_parent=collision_filter_parent, enabled=enabled)

    def add_joint_d6(self, parent, child, *, linear_axes=None, angular_axes=None, parent_xform=None, child_xform=None,
                     label=None, collision_filter_parent=None, enabled=True) -> int:
        return self.add_joint(JointType.D6, parent, child, parent_xform=parent_xform, child_xform=child_xform,
                              linear_axes=linear_axes, angular_axes=angular_axes, label=label,
                              collision_filter_parent=collision_filter_parent, enabled=enabled)

    def add_joint_free(self, child, *, parent_xform=None, child_xform=None, parent=-1, label=None,
                       collision_filter_parent=None, enabled=True) -> int:
        """builder.py:5065-5125: joint_q initialised so FK reproduces body_q[child]."""
        j = self.add_joint(JointType.FREE, parent, child, parent_xform=parent_xform, child_xform=child_xform, label=label,
                           collision_filter_parent=collision_filter_parent, enabled=enabled,
                           linear_axes=[JointDofConfig.create_unlimited(a) for a in range(3)],
                           angular_axes=[JointDofConfig.create_unlimited(a) for a in range(3)])
        q_start = self.joint_q_start[j]
        parent_body_xform = nm.transform_identity() if parent == -1 else self.body_q[parent]
        parent_anchor_world = nm.transform_mul(parent_body_xform, self.joint_X_p[j])
        jq = nm.transform_mul(nm.transform_mul(nm.transform_inverse(parent_anchor_world), self.body_q[child]),
                              self.joint_X_c[j])
        self.joint_q[q_start:q_start + 7] = list(jq)
        return j

    def add_articulation(self, joints, label=None):
        """builder.py:3076-3182 (joints must be contiguous and ascending)."""
        joints = list(joints)
        if not joints:
            raise ValueError("articulation needs at least one joint")
        if joints != list(range(joints[0], joints[0] + len(joints))):
            raise ValueError("articulation joints must be contiguous and ascending")
        child_to_parent = {}
        for j in joints:  # builder.py:3151-3163
            child, parent = self.joint_child[j], self.joint_parent[j]
            if child_to_parent.setdefault(child, parent) != parent:
                raise ValueError(f"Body {child} has multiple parents in this articulation: loop-closing joints must not "
                                 "be part of an articulation.")
            if parent != -1 and int(self.body_flags[child]) & int(BodyFlags.KINEMATIC):
                raise ValueError(f"Body {child} ('{self.body_label[child]}') is kinematic but is attached to parent body "
                                 f"{parent}. Only root bodies (whose joint parent is the world) can be kinematic.")
        aid = self.articulation_count
        self.articulation_start.append(joints[0])
        self.articulation_world.append(self.current_world)
        self.articulation_label.append(label or f"articulation_{aid}")
        for j in joints:
            self.joint_articulation[j] = aid
        return aid

    # ------------------------------------------------------------------ shapes
    def _update_body_mass(self, i, m, inertia, p, q):
        if i == -1:
            return
        new_mass = self.body_mass[i] + m
        if new_mass == 0.0:
            return
        new_com = (self.body_com[i] * self.body_mass[i] + p * m) / new_mass
        com_offset = new_com - self.body_com[i]
        shape_offset = new_com - p
        new_inertia = transform_inertia(self.body_mass[i], self.body_inertia[i], com_offset, nm.quat_identity()) + \
            transform_inertia(m, inertia, shape_offset, q)
        self.body_mass[i] = new_mass
        self.body_inertia[i] = new_inertia
        self.body_com[i] = new_com
        self.body_inv_mass[i] = 1.0 / new_mass if new_mass > 0.0 else 0.0
        self.body_inv_inertia[i] = np.linalg.inv(new_inertia) if new_inertia.any() else new_inertia.copy()

    def add_shape(self, *, body, type, xform=None, cfg=None, scale=None, is_static=False, label=None, src=None) -> int:
        cfg = self.default_shape_cfg if cfg is None else cfg
        xform = nm.transform() if xform is None else np.asarray(xform, dtype=np.float64)
        scale = (1.0, 1.0, 1.0) if scale is None else scale
        if type in (GeoType.SPHERE, GeoType.BOX, GeoType.CAPSULE, GeoType.CYLINDER, GeoType.ELLIPSOID, GeoType.PLANE,
                    GeoType.CONE):
            scale = tuple(abs(float(s)) for s in scale)
            if type == GeoType.CYLINDER and scale[2] != 0.0 and scale[2] < scale[1]:  # (builder.py:6590-6591)
                raise ValueError(f"Cylinder barrel radius must be zero or at least the half-height; got scale={scale}.")
        self.shape_body.append(body)
        shape = self.shape_count  # shape_type not yet appended
        if cfg.has_shape_collision:
            for other in self.body_shapes[body]:
                if self.shape_flags[other] & ShapeFlags.COLLIDE_SHAPES:
                    self.add_shape_collision_filter_pair(other, shape)
        self.body_shapes[body].append(shape)
        self.shape_label.append(label or f"shape_{shape}")
        self.shape_transform.append(xform)
        self.shape_flags.append(cfg.flags)
        self.shape_type.append(int(type))
        self.shape_scale.append(tuple(float(s) for s in scale))
        self.shape_margin.append(cfg.margin)
        self.shape_material_ke.append(cfg.ke)
        self.shape_material_kd.append(cfg.kd)
        self.shape_material_kf.append(cfg.kf)
        self.shape_material_ka.append(cfg.ka)
        self.shape_material_mu.append(cfg.mu)
        self.shape_material_restitution.append(cfg.restitution)
        self.shape_material_mu_torsional.append(cfg.mu_torsional)
        self.shape_material_mu_rolling.append(cfg.mu_rolling)
        self.shape_material_kh.append(cfg.kh)
        self.shape_gap.append(cfg.gap if cfg.gap is not None else self.rigid_gap)
        self.shape_collision_group.append(cfg.collision_group)
        self.shape_collision_radius.append(compute_shape_radius(type, scale, src))
        self.shape_source.append(src)
        self.shape_world.append(self.current_world)
        cfg.validate(type)
        if src is not None and (cfg.sdf_max_resolution is not None or cfg.sdf_target_voxel_size is not None):
            raise ValueError("Mesh-backed shapes do not use cfg.sdf_* for SDF generation. "
                             "Build and attach an SDF on the mesh via mesh.build_sdf().")  # builder.py:6544-6558
        if cfg.is_hydroelastic and src is not None and getattr(src, "sdf", None) is None:
            raise ValueError("Hydroelastic mesh-backed shapes require mesh.sdf. "
                             "Call mesh.build_sdf() before adding a mesh-backed hydroelastic shape.")
        self.shape_sdf_narrow_band_range.append(tuple(cfg.sdf_narrow_band_range))
        self.shape_sdf_target_voxel_size.append(cfg.sdf_target_voxel_size)
        self.shape_sdf_max_resolution.append(cfg.sdf_max_resolution)
        self.shape_sdf_texture_format.append(cfg.sdf_texture_format)
        self.shape_sdf_padding.append(cfg.sdf_padding)

        if cfg.has_shape_collision and cfg.collision_filter_parent:
            for parent_body, joint_idx in self.joint_parents.get(body, ()):
                if not self.joint_collision_filter_parent[joint_idx]:
                    continue
                for ps in self.body_shapes[parent_body]:
                    if self.shape_flags[ps] & ShapeFlags.COLLIDE_SHAPES:
                        self.add_shape_collision_filter_pair(ps, shape)
            for child_body, joint_idx in self.joint_children.get(body, ()):
                if not self.joint_collision_filter_parent[joint_idx]:
                    continue
                for cs in self.body_shapes[child_body]:
                    if self.shape_flags[cs] & ShapeFlags.COLLIDE_SHAPES:
                        self.add_shape_collision_filter_pair(shape, cs)

        if not is_static and cfg.density > 0.0 and body >= 0 and not self.body_lock_inertia[body]:
            m, c, inertia = compute_inertia_shape(type, scale, cfg.density, src)
            com_body = nm.transform_point(xform, c)
            self._update_body_mass(body, m, inertia, com_body, xform[3:])
        return shape

    def add_shape_plane(self, plane=(0.0, 0.0, 1.0, 0.0), *, xform=None, width=10.0, length=10.0, body=-1, cfg=None,
                        label=None) -> int:
        if xform is None:
            normal = np.asarray(plane[:3], dtype=np.float64)
            norm = np.linalg.norm(normal)
            normal = normal / norm
            pos = -(plane[3] / norm) * normal
            rot = nm.quat_between_vectors([0.0, 0.0, 1.0], normal)
            xform = nm.transform(pos, rot)
        return self.add_shape(body=body, type=GeoType.PLANE, xform=xform, cfg=cfg, scale=(width, length, 0.0),
                              is_static=True, label=label)

    def add_ground_plane(self, *, height=0.0, cfg=None, label=None) -> int:
        return self.add_shape_plane(plane=(*self.up_vector, -height), width=0.0, length=0.0, cfg=cfg,
                                    label=label or "ground_plane")

    def add_shape_sphere(self, body, *, xform=None, radius=1.0, cfg=None, label=None) -> int:
        return self.add_shape(body=body, type=GeoType.SPHERE, xform=xform, cfg=cfg, scale=(radius, 0.0, 0.0), label=label)

    def add_shape_ellipsoid(self, body, *, xform=None, rx=1.0, ry=0.75, rz=0.5, cfg=None, label=None) -> int:
        return self.add_shape(body=body, type=GeoType.ELLIPSOID, xform=xform, cfg=cfg, scale=(rx, ry, rz), label=label)

    def add_shape_box(self, body, *, xform=None, hx=0.5, hy=0.5, hz=0.5, cfg=None, label=None) -> int:
        return self.add_shape(body=body, type=GeoType.BOX, xform=xform, cfg=cfg, scale=(hx, hy, hz), label=label)

    def add_shape_capsule(self, body, *, xform=None, radius=1.0, half_height=0.5, cfg=None, label=None) -> int:
        return self.add_shape(body=body, type=GeoType.CAPSULE, xform=xform, cfg=cfg, scale=(radius, half_height, 0.0),
                              label=label)

    def add_shape_cylinder(self, body, *, xform=None, radius=1.0, half_height=0.5, barrel_radius=0.0, cfg=None, label=None) -> int:
        """Cylinder along Z (builder.py:7041-7100).  ``barrel_radius`` [m]: radius of the symmetric circular arc revolved about the
        axis to form the side; 0.0 = straight sides, otherwise at least ``half_height``."""
        return self.add_shape(body=body, type=GeoType.CYLINDER, xform=xform, cfg=cfg, scale=(radius, half_height, barrel_radius),
                              label=label)

    def add_shape_cone(self, body, *, xform=None, radius=1.0, half_height=0.5, cfg=None, label=None) -> int:
        """Cone along +Z: base at -half_height, apex at +half_height (builder.py:7102-7156)."""
        return self.add_shape(body=body, type=GeoType.CONE, xform=xform, cfg=cfg, scale=(radius, half_height, 0.0),
                              label=label)

    def add_shape_convex_hull(self, body, *, xform=None, mesh=None, scale=None, cfg=None, label=None) -> int:
        """Convex collision shape from the vertex set of ``mesh`` (builder.py:7201-7241); the support map scans every vertex."""
        if mesh is None:
            raise ValueError("add_shape_convex_hull() requires a Mesh")
        return self.add_shape(body=body, type=GeoType.CONVEX_MESH, xform=xform, cfg=cfg, scale=scale, label=label, src=mesh)

    def add_shape_mesh(self, body, *, xform=None, mesh=None, scale=None, cfg=None, label=None) -> int:
        """Triangle-mesh collision shape (builder.py:7158-7198).  Collides with infinite planes through its vertices
        (narrow_phase.py:618-631,1744-1992) and, with an SDF attached (``mesh.build_sdf``), with other SDF shapes through the
        mesh-SDF leg; against convex primitives through its triangles (narrow_phase.py:633-638,1455-1665: the triangle leg)."""
        if mesh is None:
            raise ValueError("add_shape_mesh() requires a Mesh")
        return self.add_shape(body=body, type=GeoType.MESH, xform=xform, cfg=cfg, scale=scale, label=label, src=mesh)

    def add_shape_heightfield(self, *, xform=None, heightfield=None, scale=None, cfg=None, label=None) -> int:
        """Static heightfield terrain (builder.py add_shape_heightfield; geometry/types.py:2240-2340).  Collides with convex primitives
        and hulls cell by cell through the triangle leg (narrow_phase.py:553-583, utils/heightfield.py:280-462).  The cells ignore the
        shape scale in the reference's kernels (get_triangle_shape_from_heightfield reads hx / hy / min_z / max_z only): scale must be 1."""
        if heightfield is None:
            raise ValueError("add_shape_heightfield() requires a Heightfield")
        if scale is not None and tuple(float(x) for x in scale) != (1.0, 1.0, 1.0):
            raise NotImplementedError("heightfield shapes take scale (1, 1, 1): the collision kernels read the field's own extents")
        if 2 * (heightfield.nrow - 1) * (heightfield.ncol - 1) >= (1 << 18):
            raise NotImplementedError("heightfields with 2^18 or more triangles are not supported (the triangle index is part of the "
                                      "22-bit contact fingerprint of the reduction)")
        return self.add_shape(body=-1, type=GeoType.HFIELD, xform=xform, cfg=cfg, scale=(1.0, 1.0, 1.0), label=label, src=heightfield)

    def _finalize_sdf(self, m) -> None:
        """Texture-SDF resources of the finalized model (builder.py:11690-11960 compact SDF table + per-shape index, :12050-12116
        collision-edge tables, :11544-11611 local AABBs and voxel grids of the contact reduction).  Mesh-backed shapes use the SDF
        attached with Mesh.build_sdf(); BOX shapes with a resolution / voxel size (or the hydroelastic flag) get a generated one with
        the scale baked in and the 12 edges of the unit box as collision edges.  Host construction (newton_amd/sdf.py) -- the
        reference needs a CUDA device for the same tables."""
        from . import sdf as S  # noqa: PLC0415
        from .mesh import mesh_edge_tables  # noqa: PLC0415

        n = self.shape_count
        sdf_index = -np.ones(n, dtype=np.int32)
        edge_range = np.zeros((n, 2), dtype=np.int32)
        voxel_res = np.zeros((n, 3), dtype=np.int32)
        table, cache, edge_cache, ecs, ehs = [], {}, {}, [], []
        fmt = {"float32": S.QuantizationMode.FLOAT32, "uint16": S.QuantizationMode.UINT16, "uint8": S.QuantizationMode.UINT8}
        lo_all, hi_all = m.shape_collision_aabb_lower, m.shape_collision_aabb_upper
        for i in range(n):
            ty, src, flags = self.shape_type[i], self.shape_source[i], self.shape_flags[i]
            if not flags & ShapeFlags.COLLIDE_SHAPES:
                continue
            scale = tuple(float(x) for x in self.shape_scale[i])
            hydro = bool(flags & ShapeFlags.HYDROELASTIC)
            key = edges_of = None
            if ty in (GeoType.MESH, GeoType.CONVEX_MESH) and src is not None and getattr(src, "sdf", None) is not None:
                key = ("mesh_sdf", id(src.sdf))
                make = lambda src=src: src.sdf  # noqa: E731
                # the SDF carries the scale it was built with; the edges are scaled per shape (shape_scale on an unbaked SDF)
                edges_of = (id(src), scale if not src.sdf.scale_baked else src.sdf_scale)
                edge_args = (src.vertices, np.asarray(src.indices).reshape(-1, 3), edges_of[1])
                if src.sdf.scale_baked and tuple(src.sdf_scale) != scale:
                    raise ValueError(f"shape {i}: mesh.sdf was built with scale {tuple(src.sdf_scale)} baked in but the shape uses "
                                     f"scale {scale}; rebuild it with mesh.build_sdf(scale={scale})")
            elif (ty == GeoType.BOX and (self.shape_sdf_max_resolution[i] is not None or self.shape_sdf_target_voxel_size[i] is not None)) \
                    or (hydro and ty in (GeoType.BOX, GeoType.SPHERE, GeoType.CAPSULE, GeoType.CYLINDER, GeoType.ELLIPSOID, GeoType.CONE)):
                # generated primitive SDF, scale baked (builder.py:11822-11870): boxes on request, any primitive when hydroelastic
                res, vox = self.shape_sdf_max_resolution[i], self.shape_sdf_target_voxel_size[i]
                if res is None and vox is None:
                    res = 64
                pad = self.shape_sdf_padding[i]
                if pad is None:
                    pad = self.shape_gap[i] + (self.shape_margin[i] if hydro else 0.0)
                band = tuple(self.shape_sdf_narrow_band_range[i])
                key = ("primitive_generated", int(ty), pad, band, vox, res, scale, self.shape_sdf_texture_format[i])
                make = lambda ty=ty, scale=scale, pad=pad, band=band, res=res, vox=vox, i=i: S.create_texture_sdf_from_primitive(  # noqa: E731
                    int(ty), scale, margin=pad, narrow_band_range=band, max_resolution=res, target_voxel_size=vox,
                    quantization_mode=fmt[self.shape_sdf_texture_format[i]], scale_baked=True)
                if ty == GeoType.BOX:  # the 12 edges of the unit box carry the mesh-SDF edge contacts of planar-faced shapes
                    edges_of = ("unit_box", scale)
                    edge_args = (None, None, scale)
                ext = S.primitive_extents(int(ty), scale)
                lo_all[i], hi_all[i] = np.asarray(ext[0], np.float32), np.asarray(ext[1], np.float32)
            if key is None:
                if ty in (GeoType.MESH, GeoType.HFIELD) and src is not None:  # the reduction's voxel grid of a mesh / heightfield without an SDF
                    voxel_res[i] = S.voxel_resolution_from_aabb(lo_all[i], hi_all[i])
                continue
            if key not in cache:
                cache[key] = len(table)
                table.append(make())
            sdf_index[i] = cache[key]
            voxel_res[i] = S.voxel_resolution_from_aabb(lo_all[i], hi_all[i])
            if edges_of is None:
                continue
            if edges_of not in edge_cache:
                if edges_of[0] == "unit_box":
                    ec, eh = _box_edge_tables(edge_args[2])
                else:
                    ec, eh = mesh_edge_tables(edge_args[0], edge_args[1], scale=edge_args[2])
                edge_cache[edges_of] = (sum(len(e) for e in ecs), len(ec))
                ecs.append(ec)
                ehs.append(eh)
            edge_range[i] = edge_cache[edges_of]
            if ty == GeoType.MESH and src is not None:  # local AABB of triangle meshes (CONVEX_MESH has it from the hull table)
                v = np.asarray(src.vertices, np.float64) * np.asarray(scale, np.float64)
                lo_all[i], hi_all[i] = v.min(axis=0), v.max(axis=0)
            voxel_res[i] = S.voxel_resolution_from_aabb(lo_all[i], hi_all[i])
        m._shape_sdf_index = sdf_index
        m._texture_sdf_data = table
        m.shape_edge_range = edge_range
        m.mesh_edge_centers = np.concatenate(ecs).astype(np.float32) if ecs else np.zeros((0, 4), np.float32)
        m.mesh_edge_halves = np.concatenate(ehs).astype(np.float32) if ehs else np.zeros((0, 4), np.float32)
        m._shape_voxel_resolution = voxel_res

    # ------------------------------------------------------------------ importers
    def add_urdf(self, source, **kwargs):
        from .urdf import parse_urdf  # noqa: PLC0415

        return parse_urdf(self, source, **kwargs)

    # ------------------------------------------------------------------ composition
    _BODY_LISTS = ["body_q", "body_qd", "body_mass", "body_inertia", "body_inv_mass", "body_inv_inertia", "body_com",
                   "body_flags", "body_label", "body_lock_inertia"]
    _DOF_LISTS = ["joint_qd", "joint_f", "joint_target_qd", "joint_axis", "joint_limit_lower", "joint_limit_upper",
                  "joint_limit_ke", "joint_limit_kd", "joint_target_ke", "joint_target_kd", "joint_damping",
                  "joint_armature", "joint_effort_limit", "joint_velocity_limit", "joint_friction"]
    _JOINT_LISTS = ["joint_type", "joint_X_p", "joint_X_c", "joint_label", "joint_dof_dim", "joint_enabled",
                    "joint_collision_filter_parent"]
    _SHAPE_LISTS = ["shape_type", "shape_scale", "shape_transform", "shape_flags", "shape_margin", "shape_gap",
                    "shape_label", "shape_collision_group", "shape_collision_radius", "shape_material_ke",
                    "shape_material_kd", "shape_material_kf", "shape_material_ka", "shape_material_mu",
                    "shape_material_restitution", "shape_material_mu_torsional", "shape_material_mu_rolling",
                    "shape_material_kh", "shape_sdf_narrow_band_range", "shape_sdf_target_voxel_size", "shape_sdf_max_resolution",
                    "shape_sdf_texture_format", "shape_sdf_padding"]

    def request_state_attributes(self, *attributes: str) -> None:
        """Extended State attributes to allocate (builder.py request_state_attributes; state.py:77): 'body_parent_f'."""
        from .model import Model  # noqa: PLC0415

        Model._check_requested(attributes, Model.EXTENDED_STATE_ATTRIBUTES, "state")
        self._requested_state_attributes.update(attributes)

    def request_contact_attributes(self, *attributes: str) -> None:
        """Extended Contacts attributes to allocate (contacts.py:170-226): 'force'."""
        from .model import Model  # noqa: PLC0415

        Model._check_requested(attributes, Model.EXTENDED_CONTACT_ATTRIBUTES, "contact")
        self._requested_contact_attributes.update(attributes)

    def add_builder(self, other: ModelBuilder, xform=None, world: int | None = None):
        """Append a copy of ``other`` (builder.py:4261-4313); entities go to ``world`` (default: current world)."""
        self._requested_state_attributes |= other._requested_state_attributes
        self._requested_contact_attributes |= other._requested_contact_attributes
        w = self.current_world if world is None else world
        b0, j0, s0 = self.body_count, self.joint_count, self.shape_count
        q0, qd0, tq0, a0 = len(self.joint_q), len(self.joint_qd), len(self.joint_target_q), self.articulation_count
        for name in self._BODY_LISTS + self._DOF_LISTS + self._JOINT_LISTS + self._SHAPE_LISTS:
            getattr(self, name).extend(_clone(x) for x in getattr(other, name))
        self.joint_q.extend(other.joint_q)
        self.joint_target_q.extend(other.joint_target_q)
        self.body_world.extend([w] * other.body_count)
        self.joint_world.extend([w] * other.joint_count)
        self.shape_world.extend([w] * other.shape_count)
        self.shape_source.extend(other.shape_source)
        self.joint_parent.extend([p + b0 if p >= 0 else -1 for p in other.joint_parent])
        self.joint_child.extend([c + b0 for c in other.joint_child])
        self.joint_q_start.extend([q + q0 for q in other.joint_q_start])
        self.joint_qd_start.extend([q + qd0 for q in other.joint_qd_start])
        self.joint_target_q_start.extend([q + tq0 for q in other.joint_target_q_start])
        self.joint_articulation.extend([a + a0 if a >= 0 else -1 for a in other.joint_articulation])
        self.articulation_start.extend([s + j0 for s in other.articulation_start])
        self.articulation_world.extend([w] * other.articulation_count)
        self.articulation_label.extend(other.articulation_label)
        self.shape_body.extend([b + b0 if b >= 0 else -1 for b in other.shape_body])
        for b, shapes in other.body_shapes.items():
            key = b + b0 if b >= 0 else -1
            self.body_shapes.setdefault(key, []).extend(s + s0 for s in shapes)
        for child, lst in other.joint_parents.items():
            self.joint_parents.setdefault(child + b0, []).extend((p + b0 if p >= 0 else -1, j + j0) for p, j in lst)
        for parent, lst in other.joint_children.items():
            key = parent + b0 if parent >= 0 else -1
            self.joint_children.setdefault(key, []).extend((c + b0, j + j0) for c, j in lst)
        for a, b in other.shape_collision_filter_pairs:
            self.shape_collision_filter_pairs.add((a + s0, b + s0))
        if xform is not None:
            xform = np.asarray(xform, dtype=np.float64)
            for b in range(b0, self.body_count):
                self.body_q[b] = nm.transform_mul(xform, self.body_q[b])
            for j in range(j0, self.joint_count):
                if self.joint_parent[j] == -1:
                    self.joint_X_p[j] = nm.transform_mul(xform, self.joint_X_p[j])
                    if self.joint_type[j] == JointType.FREE:
                        qs = self.joint_q_start[j]
                        self.joint_q[qs:qs + 7] = list(nm.transform_mul(xform, np.asarray(self.joint_q[qs:qs + 7])))
            for s in range(s0, self.shape_count):
                if self.shape_body[s] == -1:
                    self.shape_transform[s] = nm.transform_mul(xform, self.shape_transform[s])

    def add_world(self, other: ModelBuilder, xform=None):
        self.begin_world()
        self.add_builder(other, xform=xform)
        self.end_world()

    def replicate(self, other: ModelBuilder, world_count: int, spacing=(0.0, 0.0, 0.0)):
        """One world per copy (builder.py:2599-2659).  Non-zero spacing is not supported (the reference itself
        recommends keeping all worlds at the origin)."""
        if any(float(s) != 0.0 for s in spacing):
            raise NotImplementedError("replicate(spacing != 0) is not supported; use viewer offsets instead")
        if self.current_world != -1:
            raise RuntimeError("Cannot begin a new world: already in world context")
        for _ in range(world_count):
            self.add_world(other)

    # ------------------------------------------------------------------ finalize
    @staticmethod
    def _test_group_pair(a, b):
        if a == 0 or b == 0:
            return False
        if a > 0:
            return a == b or b < 0
        return a != b

    def _find_shape_contact_pairs(self):
        """builder.py:12816-13060: globals-vs-globals, then per world (global, local) pairs followed by local pairs."""
        S = self.shape_count
        world = np.asarray(self.shape_world, dtype=np.int64) if S else np.zeros(0, dtype=np.int64)
        colliding = [(self.shape_flags[i] & ShapeFlags.COLLIDE_SHAPES) != 0 for i in range(S)]
        filt = self.shape_collision_filter_pairs
        globals_ = [i for i in range(S) if world[i] == -1 and colliding[i]]
        pairs = []
        for i1, a in enumerate(globals_):
            for b in globals_[i1 + 1:]:
                if self._test_group_pair(self.shape_collision_group[a], self.shape_collision_group[b]):
                    p = (min(a, b), max(a, b))
                    if p not in filt:
                        pairs.append(p)
        # shape indices grouped by world in one pass (ascending inside each world)
        coll_idx = np.flatnonzero(np.asarray(colliding, dtype=bool)) if S else np.zeros(0, dtype=np.int64)
        order = coll_idx[np.argsort(world[coll_idx], kind="stable")]
        bounds = np.searchsorted(world[order], np.arange(self.world_count + 1))
        for w in range(self.world_count):
            local = order[bounds[w]:bounds[w + 1]].tolist()
            for g in globals_:
                for l in local:
                    if self._test_group_pair(self.shape_collision_group[g], self.shape_collision_group[l]):
                        p = (min(g, l), max(g, l))
                        if p not in filt:
                            pairs.append(p)
            for i1, a in enumerate(local):
                for b in local[i1 + 1:]:
                    if self._test_group_pair(self.shape_collision_group[a], self.shape_collision_group[b]):
                        p = (a, b)
                        if p not in filt:
                            pairs.append(p)
        return np.asarray(pairs, dtype=np.int32).reshape(-1, 2)

    def finalize(self, device=None):
        from .model import Model  # noqa: PLC0415

        if self.current_world != -1:
            raise RuntimeError("finalize() called inside a world context")
        f32, i32 = np.float32, np.int32
        m = Model(device)
        m._requested_state_attributes = set(self._requested_state_attributes)
        m._requested_contact_attributes = set(self._requested_contact_attributes)
        m.world_count = max(self.world_count, 0)
        m.body_count, m.joint_count, m.shape_count = self.body_count, self.joint_count, self.shape_count
        m.joint_dof_count, m.joint_coord_count = self.joint_dof_count, self.joint_coord_count
        m.articulation_count = self.articulation_count
        m.up_axis = self.up_axis

        def arr(lst, dtype, shape):
            a = np.asarray(lst, dtype=dtype)
            return a.reshape(shape) if a.size else np.zeros(shape, dtype=dtype)

        B, J, S, D = m.body_count, m.joint_count, m.shape_count, m.joint_dof_count
        m.body_q = arr(self.body_q, f32, (B, 7))
        m.body_qd = arr(self.body_qd, f32, (B, 6))
        m.body_com = arr(self.body_com, f32, (B, 3))
        m.body_mass = arr(self.body_mass, f32, (B,))
        m.body_inertia = arr(self.body_inertia, f32, (B, 3, 3))
        # finalize() always recomputes the inverses from the (possibly user-edited) mass / inertia lists:
        # validate_and_correct_inertia_kernel, newton/_src/geometry/inertia.py:965-1113 (symmetrise, then
        # inv_mass = 1/m, inv_inertia = inverse(I) for m > 0, zero otherwise).  The eigenvalue / triangle-inequality
        # repairs of that kernel only trigger for non-physical inertias and are not restated here.
        inv_mass, inv_inertia = [], []
        for mass, I in zip(self.body_mass, self.body_inertia):
            I = 0.5 * (np.asarray(I, dtype=np.float64) + np.asarray(I, dtype=np.float64).T)
            if mass > 0.0:
                inv_mass.append(1.0 / mass)
                inv_inertia.append(np.linalg.inv(I) if I.any() else np.zeros((3, 3)))  # point mass without shapes
            else:
                inv_mass.append(0.0)
                inv_inertia.append(np.zeros((3, 3)))
        m.body_inv_mass = arr(inv_mass, f32, (B,))
        m.body_inv_inertia = arr(inv_inertia, f32, (B, 3, 3))
        m.body_flags = arr(self.body_flags, i32, (B,))
        m.body_world = arr(self.body_world, i32, (B,))
        m.body_label = list(self.body_label)
        g = [np.asarray(v, dtype=np.float64) for v in self.world_gravity] + [self._gravity_vector()]
        m.gravity = arr(g, f32, (m.world_count + 1, 3))

        m.joint_type = arr(self.joint_type, i32, (J,))
        m.joint_enabled = arr(self.joint_enabled, np.bool_, (J,))
        m.joint_parent = arr(self.joint_parent, i32, (J,))
        m.joint_child = arr(self.joint_child, i32, (J,))
        m.joint_X_p = arr(self.joint_X_p, f32, (J, 7))
        m.joint_X_c = arr(self.joint_X_c, f32, (J, 7))
        m.joint_q_start = arr(self.joint_q_start, i32, (J,))
        child_to_joint = {int(c): i for i, c in enumerate(self.joint_child)}  # builder.py:12341-12348
        m.joint_ancestor = arr([child_to_joint.get(int(p), -1) for p in self.joint_parent], i32, (J,))
        m.joint_qd_start = arr(self.joint_qd_start, i32, (J,))
        m.joint_target_q_start = arr(self.joint_target_q_start, i32, (J,))
        m.joint_dof_dim = arr(self.joint_dof_dim, i32, (J, 2))
        m.joint_articulation = arr(self.joint_articulation, i32, (J,))
        m.joint_world = arr(self.joint_world, i32, (J,))
        m.joint_label = list(self.joint_label)
        m.joint_q = arr(self.joint_q, f32, (m.joint_coord_count,))
        m.joint_qd = arr(self.joint_qd, f32, (D,))
        m.joint_f = arr(self.joint_f, f32, (D,))
        m.joint_target_q = arr(self.joint_target_q, f32, (len(self.joint_target_q),))
        m.joint_target_qd = arr(self.joint_target_qd, f32, (D,))
        m.joint_axis = arr(self.joint_axis, f32, (D, 3))
        for name in ("joint_limit_lower", "joint_limit_upper", "joint_limit_ke", "joint_limit_kd", "joint_target_ke",
                     "joint_target_kd", "joint_damping", "joint_armature", "joint_effort_limit", "joint_velocity_limit",
                     "joint_friction"):
            setattr(m, name, arr(getattr(self, name), f32, (D,)))
        m.articulation_start = arr(self.articulation_start, i32, (m.articulation_count,))
        ends = list(self.articulation_start[1:]) + [J]
        m.articulation_end = arr(ends if m.articulation_count else [], i32, (m.articulation_count,))
        m.articulation_world = arr(self.articulation_world, i32, (m.articulation_count,))
        m.articulation_label = list(self.articulation_label)

        m.shape_transform = arr(self.shape_transform, f32, (S, 7))
        m.shape_body = arr(self.shape_body, i32, (S,))
        m.shape_type = arr(self.shape_type, i32, (S,))
        m.shape_scale = arr(self.shape_scale, f32, (S, 3))
        m.shape_flags = arr(self.shape_flags, i32, (S,))
        m.shape_world = arr(self.shape_world, i32, (S,))
        m.shape_collision_group = arr(self.shape_collision_group, i32, (S,))
        m.shape_label = list(self.shape_label)
        # convex-hull vertex tables: each distinct Mesh once (exact-duplicate vertices removed, first-occurrence order),
        # local AABBs with the per-shape scale baked in (builder.py:11575-11612)
        m.shape_source = list(self.shape_source)
        uniq, starts, counts, points = {}, [], [], []
        lo_all, hi_all = np.zeros((S, 3), dtype=f32), np.zeros((S, 3), dtype=f32)
        for i, src in enumerate(self.shape_source):
            if self.shape_type[i] not in (GeoType.CONVEX_MESH, GeoType.MESH, GeoType.HFIELD) or src is None:
                starts.append(-1)
                counts.append(0)
                # local AABB of a primitive (builder.py:11601-11652): what the heightfield midphase reads of the partner shape
                ty, (sx, sy, sz) = self.shape_type[i], (float(x) for x in self.shape_scale[i])
                ext = {GeoType.SPHERE: (sx, sx, sx), GeoType.BOX: (sx, sy, sz), GeoType.ELLIPSOID: (sx, sy, sz), GeoType.CAPSULE: (sx, sx, sy + sx),
                       GeoType.CONE: (sx, sx, sy)}.get(ty)
                if ty == GeoType.CYLINDER:
                    r = sx + ((sy * sy) / (sz + np.sqrt(sz * sz - sy * sy)) if sz > 0.0 else 0.0)
                    ext = (r, r, sy)
                if ext is not None:
                    lo_all[i], hi_all[i] = -np.asarray(ext, f32), np.asarray(ext, f32)
                continue
            if id(src) not in uniq:
                v = deduplicate_vertices(src)
                uniq[id(src)] = (sum(len(p) for p in points), len(v))
                points.append(v)
            st, ct = uniq[id(src)]
            starts.append(st)
            counts.append(ct)
            v = points[[k for k, key in enumerate(uniq) if key == id(src)][0]].astype(np.float64)
            sc = np.asarray(self.shape_scale[i], dtype=f32).astype(np.float64)  # the device multiplies in fp32
            lo, hi = v.min(axis=0) * sc, v.max(axis=0) * sc
            lo_all[i], hi_all[i] = np.minimum(lo, hi), np.maximum(lo, hi)
        # triangle meshes: wp.Mesh.points as they are (the vertex index is the contact fingerprint of the mesh-plane leg,
        # narrow_phase.py:1969), each distinct Mesh once
        # ... and wp.Mesh.indices (vertex ids relative to the mesh's first vertex): the triangle leg scans them (mesh vs convex
        # primitive, collision_core.py:1218-1276; the triangle index is part of the contact fingerprint)
        vuniq, vranges, vpoints = {}, np.zeros((S, 2), dtype=i32), []
        tuniq, tranges, tindices = {}, np.zeros((S, 2), dtype=i32), []
        for i, src in enumerate(self.shape_source):
            if self.shape_type[i] != GeoType.MESH or src is None:
                continue
            if id(src) not in vuniq:
                v = np.asarray(src.vertices, dtype=f32).reshape(-1, 3)
                vuniq[id(src)] = (sum(len(p) for p in vpoints), len(v))
                vpoints.append(v)
                tri = np.asarray(src.indices, dtype=i32).reshape(-1, 3)
                tuniq[id(src)] = (sum(len(p) for p in tindices), len(tri))
                tindices.append(tri)
            vranges[i] = vuniq[id(src)]
            tranges[i] = tuniq[id(src)]
        m.mesh_vertex_range = vranges
        m.mesh_vertices = (np.concatenate(vpoints) if vpoints else np.zeros((0, 3))).astype(f32).reshape(-1, 3)
        m.mesh_triangle_range = tranges
        m.mesh_indices = (np.concatenate(tindices) if tindices else np.zeros((0, 3))).astype(i32).reshape(-1, 3)
        # heightfields: HeightfieldData per distinct Heightfield + the concatenated normalised elevation grids (builder.py finalize,
        # utils/heightfield.py:141-156)
        huniq, hidx, hdata, helev = {}, -np.ones(S, dtype=i32), [], []
        for i, src in enumerate(self.shape_source):
            if self.shape_type[i] != GeoType.HFIELD or src is None:
                continue
            if id(src) not in huniq:
                huniq[id(src)] = len(hdata)
                hdata.append((sum(len(e) for e in helev), src.nrow, src.ncol, src.hx, src.hy, src.min_z, src.max_z))
                helev.append(np.asarray(src.data, dtype=f32).reshape(-1))
            hidx[i] = huniq[id(src)]
        m.shape_heightfield_index = hidx
        m.heightfield_data = hdata  # (data_offset, nrow, ncol, hx, hy, min_z, max_z)
        m.heightfield_elevations = (np.concatenate(helev) if helev else np.zeros(0)).astype(f32)
        m.heightfield_count = len(hdata)
        m.shape_mesh_start = np.asarray(starts, dtype=i32).reshape(S)
        m.shape_mesh_count = np.asarray(counts, dtype=i32).reshape(S)
        m.mesh_points = (np.concatenate(points) if points else np.zeros((0, 3))).astype(f32).reshape(-1, 3)
        m.shape_collision_aabb_lower, m.shape_collision_aabb_upper = lo_all, hi_all
        for name in ("shape_margin", "shape_gap", "shape_collision_radius", "shape_material_ke", "shape_material_kd",
                     "shape_material_kf", "shape_material_ka", "shape_material_mu", "shape_material_restitution",
                     "shape_material_mu_torsional", "shape_material_mu_rolling", "shape_material_kh"):
            setattr(m, name, arr(getattr(self, name), f32, (S,)))
        self._finalize_sdf(m)
        m.shape_collision_filter_pairs = set(self.shape_collision_filter_pairs)
        m.shape_contact_pairs = self._find_shape_contact_pairs()
        m.shape_contact_pair_count = len(m.shape_contact_pairs)
        m.rigid_contact_max = 0
        m._build_env_template()
        return m
