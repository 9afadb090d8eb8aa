"""TEST INFRASTRUCTURE ONLY -- pytest plugin that lets the `-m gpu` test files run on a machine WITHOUT a GPU, against the
emulated kernel library (build.py): the product's Python path (DeviceModel, State, Contacts, CollisionPipeline, the solver
classes) is exercised unchanged, its "cuda" tensors are redirected to host memory and libnewton_hip.so is replaced by
libnewton_emu.so by assigning newton_amd._lib.LIB_PATH before the first load().

    python -m pytest -p emu_plugin -m gpu tests/test_gpu_parity_xpbd.py     (with tests/emu on PYTHONPATH)
    python tests/emu/run_gpu_tests_emulated.py                              (the curated subset)

This is a dry run of the GPU suite's logic (Python glue + kernel sources), not a substitute for running it on an MI355X."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import build  # noqa: E402

_EMU_LIB = build.build()

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from newton_amd import _lib as _product_loader  # noqa: E402

assert _product_loader._lib is None, "emu_plugin must be imported before the product library is loaded"
_product_loader.LIB_PATH = _EMU_LIB


def _cpu(dev):
    if dev is None:
        return None
    return "cpu" if str(dev).startswith(("cuda", "hip")) else dev


def _wrap_factory(fn):
    def wrapped(*args, **kwargs):
        if "device" in kwargs:
            kwargs["device"] = _cpu(kwargs["device"])
        return fn(*args, **kwargs)

    return wrapped


for _name in ("zeros", "empty", "full", "ones", "as_tensor", "tensor", "arange", "rand", "randn", "zeros_like", "empty_like",
              "full_like"):
    setattr(torch, _name, _wrap_factory(getattr(torch, _name)))

_real_to = torch.Tensor.to


def _to(self, *args, **kwargs):
    args = tuple(_cpu(a) if isinstance(a, (str, torch.device)) else a for a in args)
    if "device" in kwargs:
        kwargs["device"] = _cpu(kwargs["device"])
    return _real_to(self, *args, **kwargs)


torch.Tensor.to = _to
torch.Tensor.cuda = lambda self, *a, **k: self
torch.Tensor.is_cuda = property(lambda self: True)  # the product's argument checks ask for device tensors
_real_device = torch.device


class _Stream:
    cuda_stream = 0

    def synchronize(self):
        pass


class _Event:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        import time

        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)

    def synchronize(self):
        pass


torch.cuda.Event = _Event
torch.cuda.is_available = lambda: True
torch.cuda._newton_emulated = True  # tests that only make sense on the device (full-size scenes, hipGraph capture) skip on it
torch.cuda.current_stream = lambda device=None: _Stream()
torch.cuda.synchronize = lambda device=None: None
torch.cuda.set_device = lambda device: None


class _DeviceCtx:  # `with torch.cuda.device(d):` -- one (emulated) device, nothing to switch
    def __init__(self, device=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


torch.cuda.device = _DeviceCtx

import newton_amd.model as _model  # noqa: E402

_real_dm_init = _model.DeviceModel.__init__


def _dm_init(self, model):
    _real_dm_init(self, model)
    self.device = _real_device("cpu")


# DeviceModel.__init__ allocates through the wrapped factories (device kwarg) and `.to(self.device)`: make the stored device
# a host device from the start
_orig_torch_device = torch.device


def _device(*args, **kwargs):
    args = tuple(_cpu(a) if isinstance(a, str) else a for a in args)
    return _orig_torch_device(*args, **kwargs)


_model_torch = _model._torch


class _TorchProxy:
    """torch, with torch.device("cuda:0") -> device("cpu")."""

    def __getattr__(self, name):
        if name == "device":
            return _device
        return getattr(torch, name)


_model._torch = lambda: _TorchProxy()
