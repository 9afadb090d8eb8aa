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
    def __init__(self, fn, warmup: int = 1, device=None, contacts=()):
        import torch  # noqa: PLC0415

        self._contacts = tuple(contacts) if isinstance(contacts, (tuple, list)) else (contacts,)

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

    def launch(self) -> None:
        self.graph.replay()
        for c in self._contacts:  # the replayed collide launches rewrote the device buffers behind the cached exports
            c.invalidate_views()


def capture(fn, warmup: int = 1, device=None, contacts=()) -> CapturedGraph:
    """Record `fn()` (a frame of collide / step calls) into a hipGraph; `.launch()` replays it and invalidates the cached flat
    views of `contacts` (one Contacts or a sequence of them)."""
    return CapturedGraph(fn, warmup=warmup, device=device, contacts=contacts)
