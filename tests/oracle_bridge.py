"""ctypes bridge to oracle/liboracle.so (TEST INFRASTRUCTURE ONLY -- never imported by newton_amd/).

Builds the oracle's o_model / o_state / o_contacts views over a host Model's numpy arrays and exposes
``collide`` / ``xpbd_step`` / ``eval_fk`` with Newton-shaped inputs and outputs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)


class o_model(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("body_count", "joint_count", "shape_count", "dof_count", "coord_count", "world_count",
                                       "pair_count")] + [
        ("body_com", _f), ("body_mass", _f), ("body_inertia", _f), ("body_inv_mass", _f), ("body_inv_inertia", _f),
        ("body_flags", _i), ("body_world", _i), ("gravity", _f),
        ("joint_type", _i), ("joint_enabled", _i), ("joint_parent", _i), ("joint_child", _i), ("joint_X_p", _f),
        ("joint_X_c", _f), ("joint_q_start", _i), ("joint_qd_start", _i), ("joint_target_q_start", _i),
        ("joint_dof_dim", _i), ("joint_articulation", _i), ("joint_axis", _f), ("joint_limit_lower", _f),
        ("joint_limit_upper", _f), ("joint_limit_ke", _f), ("joint_limit_kd", _f), ("joint_target_ke", _f),
        ("joint_target_kd", _f), ("joint_armature", _f), ("joint_damping", _f),
        ("articulation_count", C.c_int), ("articulation_start", _i), ("articulation_end", _i),
        ("shape_transform", _f), ("shape_body", _i), ("shape_type", _i), ("shape_scale", _f), ("shape_margin", _f),
        ("shape_gap", _f), ("shape_flags", _i), ("shape_world", _i), ("shape_collision_group", _i),
        ("shape_collision_radius", _f), ("shape_material_ke", _f), ("shape_material_kd", _f), ("shape_material_kf", _f),
        ("shape_material_ka", _f), ("shape_material_mu", _f), ("shape_material_mu_torsional", _f),
        ("shape_material_mu_rolling", _f), ("shape_material_restitution", _f), ("shape_contact_pairs", _i), ("joint_ancestor", _i),
        ("mesh_points", _f), ("shape_mesh_start", _i), ("shape_mesh_count", _i), ("shape_collision_aabb_lower", _f),
        ("shape_collision_aabb_upper", _f), ("filter_pair_count", C.c_int), ("shape_collision_filter_pairs", _i),
    ]


class o_state(C.Structure):
    _fields_ = [("body_q", _f), ("body_qd", _f), ("body_f", _f), ("joint_q", _f), ("joint_qd", _f), ("body_parent_f", _f)]


class o_control(C.Structure):
    _fields_ = [("joint_f", _f), ("joint_target_q", _f), ("joint_target_qd", _f)]


class o_contacts(C.Structure):
    _fields_ = [("rigid_contact_max", C.c_int), ("rigid_contact_count", _i), ("shape0", _i), ("shape1", _i), ("point0", _f),
                ("point1", _f), ("offset0", _f), ("offset1", _f), ("normal", _f), ("margin0", _f), ("margin1", _f),
                ("tids", _i), ("stiffness", _f), ("damping", _f), ("friction_scale", _f)]


class o_xpbd_params(C.Structure):
    _fields_ = [("iterations", C.c_int), ("joint_linear_relaxation", C.c_float), ("joint_angular_relaxation", C.c_float),
                ("joint_linear_compliance", C.c_float), ("joint_angular_compliance", C.c_float),
                ("rigid_contact_relaxation", C.c_float), ("rigid_contact_con_weighting", C.c_int),
                ("angular_damping", C.c_float), ("enable_restitution", C.c_int),
                ("compute_body_velocity_from_position_delta", C.c_int)]


class o_semi_implicit_params(C.Structure):
    _fields_ = [("angular_damping", C.c_float), ("friction_smoothing", C.c_float), ("joint_attach_ke", C.c_float),
                ("joint_attach_kd", C.c_float), ("enable_tri_contact", C.c_int)]


class o_featherstone_params(C.Structure):
    _fields_ = [("angular_damping", C.c_float), ("friction_smoothing", C.c_float), ("update_mass_matrix", C.c_int),
                ("mass_matrix_cache", C.POINTER(C.c_float))]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))]
        if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            build()
        _lib = C.CDLL(LIB)
        _lib.o_collide.restype = C.c_int
        _lib.o_probe_primitive.restype = C.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


class OracleModel:
    """Keeps contiguous copies alive and exposes the o_model struct."""

    def __init__(self, model):
        self.model = model
        self._keep = {}

        def f32(name, arr=None):
            a = np.ascontiguousarray(getattr(model, name) if arr is None else arr, dtype=np.float32)
            if a.size == 0:
                a = np.zeros(1, dtype=np.float32)
            self._keep[name] = a
            return _fp(a)

        def i32(name, arr=None):
            a = np.ascontiguousarray(getattr(model, name) if arr is None else arr, dtype=np.int32)
            if a.size == 0:
                a = np.zeros(1, dtype=np.int32)
            self._keep[name] = a
            return _ip(a)

        m = o_model()
        m.body_count, m.joint_count, m.shape_count = model.body_count, model.joint_count, model.shape_count
        m.dof_count, m.coord_count = model.joint_dof_count, model.joint_coord_count
        m.world_count, m.pair_count = model.world_count, model.shape_contact_pair_count
        kin = (np.asarray(model.body_flags) & 2) != 0
        inv_mass = np.where(kin, 0.0, model.body_inv_mass).astype(np.float32)
        inv_inertia = np.where(kin[:, None, None], 0.0, model.body_inv_inertia).astype(np.float32)
        m.body_com, m.body_mass, m.body_inertia = f32("body_com"), f32("body_mass"), f32("body_inertia")
        m.body_inv_mass, m.body_inv_inertia = f32("body_inv_mass", inv_mass), f32("body_inv_inertia", inv_inertia)
        m.body_flags, m.body_world, m.gravity = i32("body_flags"), i32("body_world"), f32("gravity")
        m.joint_type = i32("joint_type")
        m.joint_enabled = i32("joint_enabled", np.asarray(model.joint_enabled, dtype=np.int32))
        m.joint_parent, m.joint_child = i32("joint_parent"), i32("joint_child")
        m.joint_X_p, m.joint_X_c = f32("joint_X_p"), f32("joint_X_c")
        m.joint_q_start, m.joint_qd_start = i32("joint_q_start"), i32("joint_qd_start")
        m.joint_target_q_start, m.joint_dof_dim = i32("joint_target_q_start"), i32("joint_dof_dim")
        m.joint_articulation = i32("joint_articulation")
        for n in ("joint_axis", "joint_limit_lower", "joint_limit_upper", "joint_limit_ke", "joint_limit_kd",
                  "joint_target_ke", "joint_target_kd", "joint_armature", "joint_damping"):
            setattr(m, n, f32(n))
        m.articulation_count = model.articulation_count
        m.articulation_start, m.articulation_end = i32("articulation_start"), i32("articulation_end")
        m.shape_transform, m.shape_body, m.shape_type = f32("shape_transform"), i32("shape_body"), i32("shape_type")
        m.shape_scale, m.shape_margin, m.shape_gap = f32("shape_scale"), f32("shape_margin"), f32("shape_gap")
        m.shape_flags, m.shape_world = i32("shape_flags"), i32("shape_world")
        m.shape_collision_group, m.shape_collision_radius = i32("shape_collision_group"), f32("shape_collision_radius")
        for n in ("shape_material_ke", "shape_material_kd", "shape_material_kf", "shape_material_ka", "shape_material_mu",
                  "shape_material_mu_torsional", "shape_material_mu_rolling", "shape_material_restitution"):
            setattr(m, n, f32(n))
        m.shape_contact_pairs = i32("shape_contact_pairs")
        m.joint_ancestor = i32("joint_ancestor")
        m.mesh_points = f32("mesh_points")
        m.shape_mesh_start = i32("shape_mesh_start")
        m.shape_mesh_count = i32("shape_mesh_count")
        m.shape_collision_aabb_lower = f32("shape_collision_aabb_lower")
        m.shape_collision_aabb_upper = f32("shape_collision_aabb_upper")
        fp = np.asarray(sorted(getattr(model, "shape_collision_filter_pairs", [])), dtype=np.int32).reshape(-1, 2)
        m.filter_pair_count = len(fp)
        m.shape_collision_filter_pairs = i32("shape_collision_filter_pairs", fp)
        self.struct = m


class OracleContacts:
    def __init__(self, cmax):
        self.max = int(cmax)
        n = max(self.max, 1)
        self.count = np.zeros(1, dtype=np.int32)
        self.shape0 = np.full(n, -1, dtype=np.int32)
        self.shape1 = np.full(n, -1, dtype=np.int32)
        self.point0 = np.zeros((n, 3), dtype=np.float32)
        self.point1 = np.zeros((n, 3), dtype=np.float32)
        self.offset0 = np.zeros((n, 3), dtype=np.float32)
        self.offset1 = np.zeros((n, 3), dtype=np.float32)
        self.normal = np.zeros((n, 3), dtype=np.float32)
        self.margin0 = np.zeros(n, dtype=np.float32)
        self.margin1 = np.zeros(n, dtype=np.float32)
        self.tids = np.full(n, -1, dtype=np.int32)
        s = o_contacts()
        s.rigid_contact_max = self.max
        s.rigid_contact_count = _ip(self.count)
        s.shape0, s.shape1 = _ip(self.shape0), _ip(self.shape1)
        s.point0, s.point1 = _fp(self.point0), _fp(self.point1)
        s.offset0, s.offset1 = _fp(self.offset0), _fp(self.offset1)
        s.normal, s.margin0, s.margin1, s.tids = _fp(self.normal), _fp(self.margin0), _fp(self.margin1), _ip(self.tids)
        self.struct = s
        self.stiffness = self.damping = self.friction_scale = None

    def set_properties(self, stiffness, damping, friction_scale):
        """Optional per-contact overrides (Contacts.rigid_contact_stiffness / _damping / _friction), flat append order."""
        n = max(self.max, 1)
        self.stiffness = np.ascontiguousarray(np.resize(np.asarray(stiffness, dtype=np.float32), n))
        self.damping = np.ascontiguousarray(np.resize(np.asarray(damping, dtype=np.float32), n))
        self.friction_scale = np.ascontiguousarray(np.resize(np.asarray(friction_scale, dtype=np.float32), n))
        self.struct.stiffness, self.struct.damping = _fp(self.stiffness), _fp(self.damping)
        self.struct.friction_scale = _fp(self.friction_scale)


class OracleState:
    def __init__(self, model, body_q=None, body_qd=None, body_f=None):
        B = model.body_count
        self.body_q = np.array(model.body_q if body_q is None else body_q, dtype=np.float32).reshape(B, 7).copy()
        self.body_qd = np.array(model.body_qd if body_qd is None else body_qd, dtype=np.float32).reshape(B, 6).copy()
        self.body_f = np.zeros((B, 6), dtype=np.float32) if body_f is None else np.array(body_f, dtype=np.float32).reshape(B, 6).copy()
        self.joint_q = np.array(model.joint_q, dtype=np.float32).copy()
        self.joint_qd = np.array(model.joint_qd, dtype=np.float32).copy()
        if self.joint_q.size == 0:
            self.joint_q = np.zeros(1, dtype=np.float32)
        if self.joint_qd.size == 0:
            self.joint_qd = np.zeros(1, dtype=np.float32)
        requested = getattr(model, "get_requested_state_attributes", lambda: [])()
        self.body_parent_f = np.zeros((B, 6), dtype=np.float32) if "body_parent_f" in requested else None

    @property
    def struct(self):
        s = o_state()
        s.body_q, s.body_qd, s.body_f = _fp(self.body_q), _fp(self.body_qd), _fp(self.body_f)
        s.joint_q, s.joint_qd = _fp(self.joint_q), _fp(self.joint_qd)
        if self.body_parent_f is not None:
            s.body_parent_f = _fp(self.body_parent_f)
        return s


class Oracle:
    """Newton-shaped facade over the C oracle for one host Model."""

    def __init__(self, model):
        self.model = model
        self.om = OracleModel(model)
        self.L = lib()

    def control(self, joint_f=None, joint_target_q=None, joint_target_qd=None):
        m = self.model
        self._cf = np.ascontiguousarray(m.joint_f if joint_f is None else joint_f, dtype=np.float32)
        self._ctq = np.ascontiguousarray(m.joint_target_q if joint_target_q is None else joint_target_q, dtype=np.float32)
        self._ctqd = np.ascontiguousarray(m.joint_target_qd if joint_target_qd is None else joint_target_qd, dtype=np.float32)
        for n in ("_cf", "_ctq", "_ctqd"):
            if getattr(self, n).size == 0:
                setattr(self, n, np.zeros(1, dtype=np.float32))
        c = o_control()
        c.joint_f, c.joint_target_q, c.joint_target_qd = _fp(self._cf), _fp(self._ctq), _fp(self._ctqd)
        return c

    def contacts(self, cmax=None):
        if cmax is None:
            cmax = max(1000, self.model.shape_contact_pair_count * 5)
        return OracleContacts(cmax)

    def collide(self, body_q, contacts: OracleContacts, broad_phase="explicit"):
        bq = np.ascontiguousarray(body_q, dtype=np.float32)
        cap = max(self.model.shape_contact_pair_count, self.model.shape_count * 8, 1)
        pairs = np.zeros((cap, 2), dtype=np.int32)
        S = self.model.shape_count
        lo = np.zeros((S, 3), dtype=np.float32)
        hi = np.zeros((S, 3), dtype=np.float32)
        bp = {"explicit": 0, "nxn": 1, "sap": 2}[broad_phase]
        n = self.L.o_collide(C.byref(self.om.struct), _fp(bq), bp, C.byref(contacts.struct), _ip(pairs), cap, _fp(lo), _fp(hi))
        return pairs[:n].copy(), lo, hi

    def xpbd_step(self, s_in: OracleState, s_out: OracleState, control, contacts, dt, **params):
        p = o_xpbd_params(params.get("iterations", 2), params.get("joint_linear_relaxation", 0.7),
                          params.get("joint_angular_relaxation", 0.4), params.get("joint_linear_compliance", 0.0),
                          params.get("joint_angular_compliance", 0.0), params.get("rigid_contact_relaxation", 0.8),
                          int(params.get("rigid_contact_con_weighting", True)), params.get("angular_damping", 0.0),
                          int(params.get("enable_restitution", False)),
                          int(params.get("compute_body_velocity_from_position_delta", False)))
        si, so = s_in.struct, s_out.struct
        force = params.get("contact_force_out")  # [Cmax, 6] float32: what update_contacts would write into contacts.force
        self.L.o_xpbd_step_report(C.byref(self.om.struct), C.byref(p), C.byref(si), C.byref(so), C.byref(control),
                                  C.byref(contacts.struct) if contacts is not None else None, C.c_float(dt),
                                  _fp(force) if force is not None else None)

    def xpbd_rollout(self, s0: OracleState, s1: OracleState, control, contacts, dt, substeps, **params):
        """substeps x {clear_forces; collide; xpbd step; swap} in one foreign call; returns the state holding the result."""
        p = o_xpbd_params(params.get("iterations", 2), params.get("joint_linear_relaxation", 0.7),
                          params.get("joint_angular_relaxation", 0.4), params.get("joint_linear_compliance", 0.0),
                          params.get("joint_angular_compliance", 0.0), params.get("rigid_contact_relaxation", 0.8),
                          int(params.get("rigid_contact_con_weighting", True)), params.get("angular_damping", 0.0),
                          int(params.get("enable_restitution", False)),
                          int(params.get("compute_body_velocity_from_position_delta", False)))
        a, b = s0.struct, s1.struct
        self.L.o_xpbd_rollout(C.byref(self.om.struct), C.byref(p), C.byref(a), C.byref(b), C.byref(control),
                              C.byref(contacts.struct), C.c_float(dt), int(substeps))
        return s1 if substeps % 2 else s0

    def semi_implicit_step(self, s_in: OracleState, s_out: OracleState, control, contacts, dt, angular_damping=0.05,
                           friction_smoothing=1.0, joint_attach_ke=1.0e4, joint_attach_kd=1.0e2):
        p = o_semi_implicit_params(angular_damping, friction_smoothing, joint_attach_ke, joint_attach_kd, 0)
        si, so = s_in.struct, s_out.struct
        self.L.o_semi_implicit_step(C.byref(self.om.struct), C.byref(p), C.byref(si), C.byref(so), C.byref(control),
                                    C.byref(contacts.struct) if contacts is not None else None, C.c_float(dt))

    def featherstone_step(self, s_in: OracleState, s_out: OracleState, control, contacts, dt, angular_damping=0.05,
                          friction_smoothing=1.0, update_mass_matrix_interval=1):
        """SolverFeatherstone.step: advances joint_q/joint_qd and rebuilds body_q/body_qd (s_in.body_q is refreshed by FK).
        update_mass_matrix_interval > 1: this facade keeps the step counter and the factor cache like the reference solver."""
        p = o_featherstone_params(angular_damping, friction_smoothing, 1, None)
        if update_mass_matrix_interval > 1:
            if not hasattr(self, "_fs_cache"):
                self._fs_cache, self._fs_step = np.zeros(max(self.model.joint_dof_count, 1) ** 2, dtype=np.float32), 0
            p.update_mass_matrix = 1 if self._fs_step % update_mass_matrix_interval == 0 else 0
            p.mass_matrix_cache = _fp(self._fs_cache)
            self._fs_step += 1
        si, so = s_in.struct, s_out.struct
        self.L.o_featherstone_step(C.byref(self.om.struct), C.byref(p), C.byref(si), C.byref(so), C.byref(control),
                                   C.byref(contacts.struct) if contacts is not None else None, C.c_float(dt))

    def eval_fk(self, joint_q, joint_qd):
        m = self.model
        bq = np.array(m.body_q, dtype=np.float32).copy()
        bqd = np.array(m.body_qd, dtype=np.float32).copy()
        jq = np.ascontiguousarray(joint_q, dtype=np.float32)
        jqd = np.ascontiguousarray(joint_qd, dtype=np.float32)
        self.L.o_eval_fk(C.byref(self.om.struct), _fp(jq), _fp(jqd), _fp(bq), _fp(bqd))
        return bq, bqd
