"""TEST INFRASTRUCTURE ONLY -- drives tests/emu/_build/libnewton_emu.so (the gfx950 kernel sources compiled for the host, see
build.py) through the same C ABI the product uses, with numpy arrays where the product passes device pointers.

Mirrors the small part of newton_amd.model.DeviceModel / state.State / collide.Contacts that lays the data out
(env-major SoA); the parameter packing itself is the product's own ``pack_param_arrays``."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from newton_amd import _lib as L  # noqa: E402  (struct layouts + signatures only; the product library is NOT loaded)
from newton_amd.model import choose_contact_scratch, pack_param_arrays, params_uniform  # noqa: E402

_emu = None


def lib():
    global _emu
    if _emu is None:
        import build  # noqa: PLC0415

        _emu = C.CDLL(build.build())
        for name, (restype, argtypes) in L.SYMBOLS.items():
            if hasattr(_emu, name):
                fn = getattr(_emu, name)
                fn.restype, fn.argtypes = restype, argtypes
    return _emu


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p).value


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed in the emulated library: status {rc}")


class EmuModel:
    """nt_model descriptor over host arrays."""

    TOPOLOGY = ("body_flags", "joint_type", "joint_enabled", "joint_parent", "joint_child", "joint_q_start", "joint_qd_start",
                "joint_tq_start", "joint_lin_count", "joint_ang_count", "shape_body", "shape_type", "shape_flags",
                "shape_group", "pair_a", "pair_b", "body_joint_start", "body_joint_list", "body_pair_start", "body_pair_list",
                "art_start", "shape_mesh_start", "shape_mesh_count", "gshape_id")

    def __init__(self, model):
        self.model, self.t = model, model.env
        t = self.t
        self.keep = {}

        def i32(a):
            x = np.ascontiguousarray(a, dtype=np.int32)
            return x if x.size else np.zeros(1, dtype=np.int32)

        def f32(a):
            x = np.ascontiguousarray(a, dtype=np.float32)
            return x if x.size else np.zeros(1, dtype=np.float32)

        d = L.nt_model()
        d.env_count, d.env_stride = t.env_count, t.env_stride
        d.nb, d.nj, d.nd, d.nc, d.ntq, d.ns, d.ng, d.np, d.cpp = t.nb, t.nj, t.nd, t.nc, t.ntq, t.ns, t.ng, t.np, t.cpp
        d.np_analytic, d.na, d.max_art_dofs, d.shape_local0 = t.np_analytic, t.na, t.max_art_dofs, t.shape_local0
        for k in self.TOPOLOGY:  # (the tiles' shape-type table: triangle meshes as pre-computed-AABB shapes, like DeviceModel)
            self.keep[k] = i32(getattr(t, "tile_shape_type" if k == "shape_type" else k))
        self.keep["mesh_points"], self.keep["shape_mesh_bounds"] = f32(t.mesh_points), f32(t.shape_mesh_bounds)
        packed = pack_param_arrays(model, t)
        d.params_uniform = params_uniform(packed, t.env_count)
        for k, v in packed.items():
            self.keep[k] = f32(v)
        for k, v in self.keep.items():
            setattr(d, k, _ptr(v))
        choose_contact_scratch(lib(), d)
        self.desc = d

    # env-major SoA <-> Newton's flat AoS
    def to_soa(self, aos, ncomp, n):
        t = self.t
        out = np.zeros((ncomp, max(n, 1), t.env_stride), dtype=np.float32)
        if n:
            out[:, :n, :t.env_count] = np.asarray(aos, dtype=np.float32).reshape(t.env_count, n, ncomp).transpose(2, 1, 0)
        return out

    def to_aos(self, soa, ncomp, n):
        t = self.t
        a = soa[:, :n, :t.env_count].transpose(2, 1, 0).reshape(t.env_count * n, ncomp)
        return a.copy() if ncomp > 1 else a.reshape(-1).copy()


class EmuState:
    def __init__(self, em: EmuModel, body_q=None, body_qd=None, joint_q=None, joint_qd=None, body_f=None, parent_f=False):
        m, t = em.model, em.t
        self.em = em
        self.body_q = em.to_soa(m.body_q if body_q is None else body_q, 7, t.nb)
        self.body_qd = em.to_soa(m.body_qd if body_qd is None else body_qd, 6, t.nb)
        self.body_f = em.to_soa(np.zeros((t.env_count * t.nb, 6)) if body_f is None else body_f, 6, t.nb)
        self.joint_q = em.to_soa(m.joint_q if joint_q is None else joint_q, 1, t.nc)
        self.joint_qd = em.to_soa(m.joint_qd if joint_qd is None else joint_qd, 1, t.nd)
        self.body_parent_f = np.zeros((6, max(t.nb, 1), t.env_stride), dtype=np.float32) if parent_f else None

    def desc(self):
        d = L.nt_state()
        d.body_q, d.body_qd, d.body_f = _ptr(self.body_q), _ptr(self.body_qd), _ptr(self.body_f)
        d.joint_q, d.joint_qd = _ptr(self.joint_q), _ptr(self.joint_qd)
        if self.body_parent_f is not None:
            d.body_parent_f = _ptr(self.body_parent_f)
        return d

    def aos(self, name):
        t = self.em.t
        ncomp, n = {"body_q": (7, t.nb), "body_qd": (6, t.nb), "body_f": (6, t.nb), "joint_q": (1, t.nc), "joint_qd": (1, t.nd),
                    "body_parent_f": (6, t.nb)}[name]
        return self.em.to_aos(getattr(self, name), ncomp, n)


class EmuControl:
    def __init__(self, em: EmuModel, joint_f=None, joint_target_q=None, joint_target_qd=None):
        m, t = em.model, em.t
        self.joint_f = em.to_soa(m.joint_f if joint_f is None else joint_f, 1, t.nd)
        self.joint_target_q = em.to_soa(m.joint_target_q if joint_target_q is None else joint_target_q, 1, t.ntq)
        self.joint_target_qd = em.to_soa(m.joint_target_qd if joint_target_qd is None else joint_target_qd, 1, t.nd)

    def desc(self):
        d = L.nt_control()
        d.joint_f, d.joint_target_q, d.joint_target_qd = _ptr(self.joint_f), _ptr(self.joint_target_q), _ptr(self.joint_target_qd)
        return d


class EmuContacts:
    def __init__(self, em: EmuModel):
        t = em.t
        self.em = em
        ns = max(t.np * t.cpp, 1)
        self.shape0 = np.full((ns, t.env_stride), -1, dtype=np.int32)
        self.shape1 = np.full((ns, t.env_stride), -1, dtype=np.int32)
        self.data = np.zeros((L.NT_CONTACT_FLOATS, ns, t.env_stride), dtype=np.float32)
        self.env_count = np.zeros(t.env_stride, dtype=np.int32)
        self.pair_hit = np.zeros((max(t.np, 1), t.env_stride), dtype=np.uint8)
        self.scan = np.zeros(4 * (t.env_stride + 1), dtype=np.int32)
        self.cw = np.zeros((15, ns, t.env_stride), dtype=np.float32) if em.desc.contact_scratch_in_hbm else None
        self.cr = np.zeros((t.env_count, ns, 32), dtype=np.float32) if em.desc.contact_scratch_in_hbm else None
        self.rigid_contact_max = t.env_count * t.np * t.cpp
        self.prop = None  # optional per-slot stiffness / damping / friction scale [3][ns][ES]

    def desc(self):
        d = L.nt_contacts()
        d.shape0, d.shape1, d.data = _ptr(self.shape0), _ptr(self.shape1), _ptr(self.data)
        d.env_count, d.pair_hit = _ptr(self.env_count), _ptr(self.pair_hit)
        if self.cw is not None:
            d.cw = _ptr(self.cw)
        if self.cr is not None:
            d.cr = _ptr(self.cr)
        if self.prop is not None:
            d.prop = _ptr(self.prop)
        return d

    def export(self):
        """Newton's flat arrays in the reference's append order (nt_contacts_export)."""
        cap = max(self.rigid_contact_max, 1)
        out = {"count": np.zeros(1, np.int32), "shape0": np.full(cap, -1, np.int32), "shape1": np.full(cap, -1, np.int32)}
        for k in ("point0", "point1", "offset0", "offset1", "normal"):
            out[k] = np.zeros((cap, 3), np.float32)
        out["margin0"], out["margin1"] = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        d = self.desc()
        check(lib().nt_contacts_export(C.byref(self.em.desc), C.byref(d), self.rigid_contact_max, _ptr(out["count"]),
                                       _ptr(out["shape0"]), _ptr(out["shape1"]), _ptr(out["point0"]), _ptr(out["point1"]),
                                       _ptr(out["offset0"]), _ptr(out["offset1"]), _ptr(out["normal"]), _ptr(out["margin0"]),
                                       _ptr(out["margin1"]), _ptr(self.scan), None), "nt_contacts_export")
        return out


def xpbd_params(iterations=2, joint_linear_relaxation=0.7, joint_angular_relaxation=0.4, joint_linear_compliance=0.0,
                joint_angular_compliance=0.0, rigid_contact_relaxation=0.8, rigid_contact_con_weighting=True, angular_damping=0.0,
                enable_restitution=False, compute_body_velocity_from_position_delta=False):
    return L.nt_xpbd_params(iterations, joint_linear_relaxation, joint_angular_relaxation, joint_linear_compliance,
                            joint_angular_compliance, rigid_contact_relaxation, int(rigid_contact_con_weighting),
                            angular_damping, int(enable_restitution), int(compute_body_velocity_from_position_delta))


def collide(em, state, contacts, epb=0):
    ds, dc, p = state.desc(), contacts.desc(), L.nt_collide_params(0, epb)
    check(lib().nt_collide(C.byref(em.desc), C.byref(ds), C.byref(dc), C.byref(p), None), "nt_collide")


def xpbd_step(em, s_in, s_out, control, contacts, dt, epb=0, report=None, **kw):
    p = xpbd_params(**kw)
    di, do_, dc = s_in.desc(), s_out.desc(), control.desc()
    dct = contacts.desc() if contacts is not None else None
    check(lib().nt_xpbd_step(C.byref(em.desc), C.byref(p), C.byref(di), C.byref(do_), C.byref(dc),
                             C.byref(dct) if dct is not None else None, float(dt), epb,
                             C.byref(report) if report is not None else None, None), "nt_xpbd_step")


def xpbd_rollout(em, s0, s1, control, contacts, dt, substeps, epb=0, **kw):
    p, cp = xpbd_params(**kw), L.nt_collide_params(0, epb)
    d0, d1, dc, dct = s0.desc(), s1.desc(), control.desc(), contacts.desc()
    check(lib().nt_xpbd_rollout(C.byref(em.desc), C.byref(p), C.byref(cp), C.byref(d0), C.byref(d1), C.byref(dc), C.byref(dct),
                                float(dt), int(substeps), None), "nt_xpbd_rollout")
    return s1 if substeps % 2 else s0


def semi_implicit_step(em, s_in, s_out, control, contacts, dt, epb=0, angular_damping=0.05, friction_smoothing=1.0,
                       joint_attach_ke=1.0e4, joint_attach_kd=1.0e2):
    p = L.nt_semi_implicit_params(angular_damping, friction_smoothing, joint_attach_ke, joint_attach_kd)
    di, do_, dc = s_in.desc(), s_out.desc(), control.desc()
    dct = contacts.desc() if contacts is not None else None
    check(lib().nt_semi_implicit_step(C.byref(em.desc), C.byref(p), C.byref(di), C.byref(do_), C.byref(dc),
                                      C.byref(dct) if dct is not None else None, float(dt), epb, None), "nt_semi_implicit_step")


def featherstone_step(em, s_in, s_out, control, contacts, dt, epb=0, angular_damping=0.05, friction_smoothing=1.0, dense=False):
    p = L.nt_featherstone_params(angular_damping, friction_smoothing)
    p.dense_mass_matrix = int(dense)
    di, do_, dc = s_in.desc(), s_out.desc(), control.desc()
    dct = contacts.desc() if contacts is not None else None
    check(lib().nt_featherstone_step(C.byref(em.desc), C.byref(p), C.byref(di), C.byref(do_), C.byref(dc),
                                     C.byref(dct) if dct is not None else None, float(dt), epb, None), "nt_featherstone_step")


def featherstone_rollout(em, s0, s1, control, contacts, dt, substeps, epb=0, angular_damping=0.05, friction_smoothing=1.0,
                         dense=False):
    p, cp = L.nt_featherstone_params(angular_damping, friction_smoothing), L.nt_collide_params(0, epb)
    p.dense_mass_matrix = int(dense)
    d0, d1, dc, dct = s0.desc(), s1.desc(), control.desc(), contacts.desc()
    check(lib().nt_featherstone_rollout(C.byref(em.desc), C.byref(p), C.byref(cp), C.byref(d0), C.byref(d1), C.byref(dc),
                                        C.byref(dct), float(dt), int(substeps), None), "nt_featherstone_rollout")
    return s1 if substeps % 2 else s0
