"""Headless viewers of the reference's example loop: ``ViewerNull`` (newton/_src/viewer/viewer_null.py:18-264) and the state
recorder part of ``ViewerFile`` (newton/_src/viewer/viewer_file.py:1112-1533): record States frame by frame, save / load a
recording, play a frame back into a State -- the checkpoint / replay tool for parity debugging.
Rendering-side calls (log_mesh, log_lines, ...) are accepted and ignored."""
from __future__ import annotations

import time as _time

import numpy as np

_STATE_FIELDS = ("body_q", "body_qd", "joint_q", "joint_qd")


def _host(x):
    return np.array(x.detach().cpu().numpy() if hasattr(x, "detach") else x, dtype=np.float32)


def _sync(model):
    if model is not None and getattr(model, "is_gpu", False):
        import torch  # noqa: PLC0415

        torch.cuda.synchronize()


class ViewerNull:
    """No-op viewer that runs for a fixed number of frames, optionally measuring frames per second after a warm-up."""

    def __init__(self, num_frames: int = 1000, benchmark: bool = False, benchmark_timeout: float | None = None,
                 benchmark_start_frame: int = 3):
        self.num_frames = num_frames
        self.frame_count = 0
        self.benchmark = benchmark or benchmark_timeout is not None
        self.benchmark_timeout = benchmark_timeout
        self.benchmark_start_frame = benchmark_start_frame
        self._bench_start_time = None
        self._bench_frames = 0
        self._bench_elapsed = 0.0
        self.model = None
        self.time = 0.0

    def set_model(self, model):
        self.model = model

    def begin_frame(self, time: float):
        self.time = time

    def end_frame(self):
        self.frame_count += 1
        if self.benchmark:
            if self.frame_count == self.benchmark_start_frame:
                _sync(self.model)
                self._bench_start_time = _time.perf_counter()
            elif self._bench_start_time is not None:
                _sync(self.model)
                self._bench_frames = self.frame_count - self.benchmark_start_frame
                self._bench_elapsed = _time.perf_counter() - self._bench_start_time

    def is_running(self) -> bool:
        if self.frame_count >= self.num_frames:
            return False
        return not (self.benchmark_timeout is not None and self._bench_start_time is not None
                    and self._bench_elapsed >= self.benchmark_timeout)

    def benchmark_result(self):
        if not self.benchmark:
            return None
        if self._bench_frames == 0 or self._bench_elapsed == 0.0:
            return {"fps": 0.0, "frames": 0, "elapsed": 0.0}
        return {"fps": self._bench_frames / self._bench_elapsed, "frames": self._bench_frames, "elapsed": self._bench_elapsed}

    def close(self):
        pass

    # rendering / logging hooks of ViewerBase: accepted, ignored
    def log_state(self, state):
        pass

    def log_contacts(self, contacts, state):
        pass

    def log_mesh(self, *args, **kwargs):
        pass

    def log_instances(self, *args, **kwargs):
        pass

    def log_lines(self, *args, **kwargs):
        pass

    def log_points(self, *args, **kwargs):
        pass

    def log_array(self, name, array):
        pass

    def log_scalar(self, name, value, *, clear: bool = False, smoothing: int = 1):
        pass

    def apply_forces(self, state):
        pass


class ViewerFile(ViewerNull):
    """Records every logged State; ``save_recording`` writes one ``.npz`` (frames x state arrays + the frame times),
    ``load_recording`` reads it back and ``load_state(state, frame_id)`` replays a frame into a State of the same model."""

    def __init__(self, output_path: str | None = None, auto_save: bool = True, max_history_size: int | None = None):
        super().__init__(num_frames=2**31 - 1)
        self.output_path = output_path
        self.auto_save = auto_save
        self.max_history_size = max_history_size
        self._frames: list[dict] = []
        self._times: list[float] = []

    def log_state(self, state):
        self.record(state)

    def record(self, state):
        self._frames.append({k: _host(getattr(state, k)) for k in _STATE_FIELDS})
        self._times.append(float(self.time))
        if self.max_history_size is not None and len(self._frames) > self.max_history_size:  # ring buffer behaviour
            del self._frames[0], self._times[0]

    def get_frame_count(self) -> int:
        return len(self._frames)

    def has_model(self) -> bool:
        return self.model is not None

    def playback(self, state, frame_id: int):
        self.load_state(state, frame_id)

    def load_state(self, state, frame_id: int):
        if not 0 <= frame_id < len(self._frames):
            raise IndexError(f"frame {frame_id} out of range (recording has {len(self._frames)} frames)")
        for k, v in self._frames[frame_id].items():
            setattr(state, k, v)

    def save_recording(self, file_path: str | None = None, verbose: bool = False):
        path = file_path or self.output_path
        if path is None:
            raise ValueError("no file path given")
        arrays = {k: np.stack([f[k] for f in self._frames]) if self._frames else np.zeros((0,), np.float32) for k in _STATE_FIELDS}
        np.savez_compressed(path, times=np.asarray(self._times, dtype=np.float64), **arrays)
        if verbose:
            print(f"saved {len(self._frames)} frames to {path}")

    def load_recording(self, file_path: str | None = None, verbose: bool = False):
        path = file_path or self.output_path
        if path is None:
            raise ValueError("no file path given")
        data = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        n = len(data["times"])
        self._times = [float(t) for t in data["times"]]
        self._frames = [{k: data[k][i] for k in _STATE_FIELDS} for i in range(n)]
        if verbose:
            print(f"loaded {n} frames from {path}")

    def close(self):
        if self.auto_save and self.output_path is not None and self._frames:
            self.save_recording()
