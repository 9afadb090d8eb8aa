"""State and Control containers (newton/_src/sim/state.py:113-262, control.py:31-68).

Host models (device 'cpu') hold numpy AoS arrays.  GPU models hold env-major SoA torch tensors
(``_soa[name]`` with shape [comp, slots_per_env, env_stride]); the Newton-shaped AoS arrays
(``body_q[B,7]`` ...) are produced on read and packed on write through nt_unpack_aos / nt_pack_aos.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _torch():
    import torch  # noqa: PLC0415

    return torch


class _SoAContainer:
    """Shared AoS <-> SoA plumbing. ``_FIELDS``: name -> (ncomp, slots attr on EnvTemplate, model default attr)."""

    _FIELDS: dict = {}

    def __init__(self, model):
        self.model = model
        self.requires_grad = False
        self._soa = {}
        self._host = {}
        t = model.env
        if model.is_gpu:
            torch = _torch()
            dm = model.device_model()
            for name, (ncomp, slots, default) in self._FIELDS.items():
                n = getattr(t, slots)
                self._soa[name] = torch.zeros((ncomp, max(n, 1), t.env_stride), dtype=torch.float32, device=dm.device)
                if default is not None and n > 0:
                    self._set(name, getattr(model, default))
        else:
            for name, (ncomp, slots, default) in self._FIELDS.items():
                n = getattr(t, slots) * t.env_count
                shape = (n, ncomp) if ncomp > 1 else (n,)
                src = getattr(model, default) if default is not None else None
                self._host[name] = np.array(src, dtype=np.float32).reshape(shape) if src is not None else np.zeros(shape, np.float32)

    def _get(self, name):
        if not self.model.is_gpu:
            return self._host[name]
        torch = _torch()
        ncomp, slots, _ = self._FIELDS[name]
        t = self.model.env
        n = getattr(t, slots)
        dm = self.model.device_model()
        shape = (t.env_count * n, ncomp) if ncomp > 1 else (t.env_count * n,)
        out = torch.empty(shape, dtype=torch.float32, device=dm.device)
        if n > 0:
            _lib.check(dm.lib.nt_unpack_aos(self._soa[name].data_ptr(), out.data_ptr(), ncomp, n, t.env_count, t.env_stride,
                                            dm.stream()), "nt_unpack_aos")
        return out

    def _set(self, name, value):
        ncomp, slots, _ = self._FIELDS[name]
        t = self.model.env
        n = getattr(t, slots)
        if not self.model.is_gpu:
            shape = (t.env_count * n, ncomp) if ncomp > 1 else (t.env_count * n,)
            self._host[name] = np.array(value, dtype=np.float32).reshape(shape)
            return
        torch = _torch()
        dm = self.model.device_model()
        if not isinstance(value, torch.Tensor):
            value = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32))
        value = value.to(device=dm.device, dtype=torch.float32).contiguous()
        if value.numel() != t.env_count * n * ncomp:
            raise ValueError(f"{name}: expected {t.env_count * n * ncomp} values, got {value.numel()}")
        if n > 0:
            _lib.check(dm.lib.nt_pack_aos(value.data_ptr(), self._soa[name].data_ptr(), ncomp, n, t.env_count, t.env_stride,
                                          dm.stream()), "nt_pack_aos")
            torch.cuda.current_stream(dm.device).synchronize()  # `value` may be a temporary


def pack_soa(model, value, ncomp: int, n: int):
    """Newton flat AoS array ([E*n, ncomp] or [E*n]) -> a fresh env-major SoA device tensor [ncomp, n, ES]."""
    torch = _torch()
    t = model.env
    dm = model.device_model()
    out = torch.zeros((ncomp, max(n, 1), t.env_stride), dtype=torch.float32, device=dm.device)
    if not isinstance(value, torch.Tensor):
        value = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32))
    value = value.to(device=dm.device, dtype=torch.float32).contiguous()
    if value.numel() != t.env_count * n * ncomp:
        raise ValueError(f"expected {t.env_count * n * ncomp} values, got {value.numel()}")
    if n > 0:
        _lib.check(dm.lib.nt_pack_aos(value.data_ptr(), out.data_ptr(), ncomp, n, t.env_count, t.env_stride, dm.stream()),
                   "nt_pack_aos")
        torch.cuda.current_stream(dm.device).synchronize()  # `value` may be a temporary
    return out


def _aos_property(name):
    def getter(self):
        return self._get(name)

    def setter(self, value):
        self._set(name, value)

    return property(getter, setter)


class State(_SoAContainer):
    """Time-varying simulation state (state.py:113-171): body_q [B,7], body_qd [B,6], body_f [B,6], joint_q, joint_qd.

    On a GPU model the resident arrays are env-major SoA and READING an attribute (``state.body_q``) materialises a fresh AoS
    copy: in-place edits of that copy (``state.body_q[i] = x``, ``.copy_()``) do NOT reach the device state.  Write whole
    arrays through the setter (``state.body_q = new_q``), or use ``State.reset`` / ``ArticulationView`` setters for masked
    updates; host models hold plain numpy arrays that are writable in place."""

    _FIELDS = {
        "body_q": (7, "nb", "body_q"),
        "body_qd": (6, "nb", "body_qd"),
        "body_f": (6, "nb", None),
        "joint_q": (1, "nc", "joint_q"),
        "joint_qd": (1, "nd", "joint_qd"),
    }
    body_q = _aos_property("body_q")
    body_qd = _aos_property("body_qd")
    body_f = _aos_property("body_f")
    joint_q = _aos_property("joint_q")
    joint_qd = _aos_property("joint_qd")

    def __init__(self, model):
        super().__init__(model)
        # extended attribute (state.py:77,156-163): allocated only when requested on the model / builder
        self._parent_f = None
        if "body_parent_f" in model.get_requested_state_attributes():
            t = model.env
            if model.is_gpu:
                self._parent_f = _torch().zeros((6, max(t.nb, 1), t.env_stride), dtype=_torch().float32,
                                                device=model.device_model().device)
            else:
                self._parent_f = np.zeros((t.env_count * t.nb, 6), dtype=np.float32)
        self.particle_count = 0

    @property
    def body_parent_f(self):
        """[body_count, 6] incoming joint wrench per body (world frame, at the COM), or None when not requested."""
        if self._parent_f is None or not self.model.is_gpu:
            return self._parent_f
        t = self.model.env
        dm = self.model.device_model()
        out = _torch().empty((t.env_count * t.nb, 6), dtype=_torch().float32, device=dm.device)
        _lib.check(dm.lib.nt_unpack_aos(self._parent_f.data_ptr(), out.data_ptr(), 6, t.nb, t.env_count, t.env_stride,
                                        dm.stream()), "nt_unpack_aos")
        return out

    @property
    def body_count(self):
        return self.model.body_count

    def clear_forces(self):
        """state.py:189-200"""
        if not self.model.is_gpu:
            self._host["body_f"][...] = 0.0
            return
        dm = self.model.device_model()
        d = self._desc()
        _lib.check(dm.lib.nt_clear_forces(C.byref(dm.desc), C.byref(d), dm.stream()), "nt_clear_forces")

    def assign(self, other: State):
        """state.py:202-262"""
        if self.model.is_gpu:
            for k in self._soa:
                self._soa[k].copy_(other._soa[k])
        else:
            for k in self._host:
                self._host[k] = other._host[k].copy()
        if self._parent_f is not None and other._parent_f is not None:  # extended attribute travels with the state
            if self.model.is_gpu:
                self._parent_f.copy_(other._parent_f)
            else:
                self._parent_f[...] = other._parent_f

    def reset(self, source: State, world_mask=None):
        """RL-style reset: copy ``source`` (e.g. a default state) into the worlds selected by ``world_mask``
        (bool, shape (world_count,) or the reference's (world_count + 1,) with the trailing global-entity slot;
        None = every world).  newton/_src/solvers/solver.py:344-375, core/reset.py:13-60."""
        t = self.model.env
        if world_mask is None:
            return self.assign(source)
        if self.model.is_gpu:
            torch = _torch()
            dm = self.model.device_model()
            mask = torch.as_tensor(world_mask, device=dm.device).to(torch.uint8).contiguous()
        else:
            mask = np.asarray(world_mask).astype(np.uint8)
        if mask.ndim != 1 or mask.shape[0] not in (t.env_count, t.env_count + 1):
            raise ValueError(f"'world_mask' length {mask.shape[0]} must equal model.world_count + 1 ({t.env_count + 1})")
        mask = mask[: t.env_count]
        if not self.model.is_gpu:
            sel = mask.astype(bool)
            for name, (ncomp, slots, _) in self._FIELDS.items():
                n = getattr(t, slots)
                if n == 0:
                    continue
                dst = self._host[name].reshape(t.env_count, -1)
                dst[sel] = source._host[name].reshape(t.env_count, -1)[sel]
            return None
        mask = mask.contiguous()
        d, s_ = self._desc(), source._desc()
        _lib.check(dm.lib.nt_state_reset(C.byref(dm.desc), C.byref(d), C.byref(s_), mask.data_ptr(), dm.stream()),
                   "nt_state_reset")
        _torch().cuda.current_stream(dm.device).synchronize()  # `mask` may be a temporary
        return None

    def _desc(self) -> _lib.nt_state:
        d = _lib.nt_state()
        d.body_q = self._soa["body_q"].data_ptr()
        d.body_qd = self._soa["body_qd"].data_ptr()
        d.body_f = self._soa["body_f"].data_ptr()
        d.joint_q = self._soa["joint_q"].data_ptr()
        d.joint_qd = self._soa["joint_qd"].data_ptr()
        if self._parent_f is not None:
            d.body_parent_f = self._parent_f.data_ptr()
        return d


class Control(_SoAContainer):
    """Control inputs (control.py:31-68): joint_f [D], joint_target_q [coords or D], joint_target_qd [D]."""

    _FIELDS = {
        "joint_f": (1, "nd", "joint_f"),
        "joint_target_q": (1, "ntq", "joint_target_q"),
        "joint_target_qd": (1, "nd", "joint_target_qd"),
    }
    joint_f = _aos_property("joint_f")
    joint_target_q = _aos_property("joint_target_q")
    joint_target_qd = _aos_property("joint_target_qd")

    def clear(self, model=None):
        """control.py:76-105: zero joint_f and joint_target_qd; joint_target_q is restored from ``model.joint_target_q`` when a
        model is passed (zeroing it would corrupt the quaternion slots of FREE / BALL / DISTANCE joints under the coord
        layout) and zero-filled otherwise."""
        for name in ("joint_f", "joint_target_qd"):
            if self.model.is_gpu:
                self._soa[name].zero_()
            else:
                self._host[name][...] = 0.0
        if model is not None and getattr(model, "joint_target_q", None) is not None:
            self.joint_target_q = model.joint_target_q
        elif self.model.is_gpu:
            self._soa["joint_target_q"].zero_()
        else:
            self._host["joint_target_q"][...] = 0.0

    def _desc(self) -> _lib.nt_control:
        d = _lib.nt_control()
        d.joint_f = self._soa["joint_f"].data_ptr()
        d.joint_target_q = self._soa["joint_target_q"].data_ptr()
        d.joint_target_qd = self._soa["joint_target_qd"].data_ptr()
        return d
