"""HIP-graph capture of a caller's frame, the counterpart of `wp.ScopedCapture` / `wp.capture_launch` around `simulate()` in the
reference's examples (newton/examples/basic/example_basic_urdf.py:117-141): every call of this package launches on torch's
current stream with caller-owned buffers and without a host read, so a whole frame -- clear_forces / collide / step per
substep -- records into one hipGraph and replays as ONE host call.

    graph = newton_amd.graph.capture(simulate)      # runs simulate() once to warm up, once more under capture
    for _ in range(frames):
        graph.launch()

`simulate` must leave the Python-side state as it found it (e.g. an even number of state swaps, as the examples do): replay
repeats the recorded launches on the recorded buffers.  Pass the frame's Contacts objects as `contacts=`: a replay re-runs the
recorded collide launches without going through `CollisionPipeline.collide`, so `launch()` invalidates their cached flat views
(`rigid_contact_count`, `rigid_contact_shape0`, ...) the way `collide()` does."""
from __future__ import annotations


class CapturedGraph:
    """backend="torch": torch.cuda.CUDAGraph (its capture also redirects allocations made inside `fn`); backend="abi": the C ABI's
    own helper (nt_graph_capture_begin / _end / _launch, include/newton_hip.h) on a side stream -- what a host without torch binds;
    `fn` must not allocate device memory or synchronise (a frame of collide / step / rollout calls does neither)."""

    def __init__(self, fn, warmup: int = 1, device=None, contacts=(), backend: str = "torch"):
        import torch  # noqa: PLC0415

        self._contacts = tuple(contacts) if isinstance(contacts, (tuple, list)) else (contacts,)
        if backend not in ("torch", "abi"):
            raise ValueError(f"backend must be 'torch' or 'abi', got {backend!r}")
        self._abi = None
        for c in self._contacts:  # (buffers the fused rollout allocates on first use: a capture must not allocate)
            if hasattr(c, "prepare_rollout"):
                c.prepare_rollout()
        if backend == "abi":
            self._init_abi(fn, warmup, device)
            return

        if not torch.cuda.is_available():
            raise RuntimeError("newton_amd.graph.capture needs a GPU (hipGraph capture)")
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):  # lazy allocations / hipFuncSetAttribute calls happen here, outside the capture
            for _ in range(max(int(warmup), 0)):
                fn()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()

    def _init_abi(self, fn, warmup, device):
        import ctypes as C  # noqa: PLC0415

        import torch  # noqa: PLC0415

        from . import _lib  # noqa: PLC0415

        if not torch.cuda.is_available():
            raise RuntimeError("newton_amd.graph.capture needs a GPU (hipGraph capture)")
        lib = _lib.load()
        self._lib, self._stream = lib, torch.cuda.Stream(device=device)
        self._stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(self._stream):  # the entry points launch on torch's current stream: make it the capture stream
            for _ in range(max(int(warmup), 1)):  # (at least once: function attributes of the big-LDS kernels are set on first use)
                fn()
            self._stream.synchronize()
            h = C.c_void_p(self._stream.cuda_stream)
            _lib.check(lib.nt_graph_capture_begin(h), "nt_graph_capture_begin")
            try:
                fn()
            finally:
                g = C.c_void_p()
                status = lib.nt_graph_capture_end(h, C.byref(g))
            _lib.check(status, "nt_graph_capture_end")
        self._abi = g
        torch.cuda.current_stream(device).wait_stream(self._stream)

    def __del__(self):
        if getattr(self, "_abi", None):
            self._lib.nt_graph_destroy(self._abi)
            self._abi = None

    def launch(self) -> None:
        if self._abi is not None:
            import ctypes as C  # noqa: PLC0415

            import torch  # noqa: PLC0415

            from . import _lib  # noqa: PLC0415

            cur = torch.cuda.current_stream(self._stream.device)
            _lib.check(self._lib.nt_graph_launch(self._abi, C.c_void_p(cur.cuda_stream)), "nt_graph_launch")
        else:
            self.graph.replay()
        for c in self._contacts:  # the replayed collide launches rewrote the device buffers behind the cached exports
            c.invalidate_views()


def capture(fn, warmup: int = 1, device=None, contacts=(), backend: str = "torch") -> CapturedGraph:
    """Record `fn()` (a frame of collide / step calls) into a hipGraph; `.launch()` replays it and invalidates the cached flat
    views of `contacts` (one Contacts or a sequence of them).  backend="abi" records through the C ABI's nt_graph_* helper."""
    return CapturedGraph(fn, warmup=warmup, device=device, contacts=contacts, backend=backend)
