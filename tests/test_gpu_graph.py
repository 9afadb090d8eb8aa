"""hipGraph capture of a frame (the reference's examples wrap simulate() in wp.ScopedCapture, example_basic_urdf.py:117-141):
replaying the captured frame gives bit-identical states to the same calls issued one by one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _needs_device():
    import torch

    if not torch.cuda.is_available() or getattr(torch.cuda, "_newton_emulated", False):
        pytest.skip("hipGraph capture needs the device (not emulated)")


def _frame(model, pipe, solver, st, ctrl, contacts, dt, substeps):
    def simulate():
        for _ in range(substeps):
            st[0].clear_forces()
            pipe.collide(st[0], contacts)
            solver.step(st[0], st[1], ctrl, contacts, dt)
            st[0], st[1] = st[1], st[0]
    return simulate


@pytest.mark.parametrize("backend", ["torch", "abi"])
@pytest.mark.parametrize("scene", ["quadruped", "sdf"])
def test_captured_frame_replays_bit_identically(scene, backend):
    """backend "abi": the C ABI's own capture helper (nt_graph_capture_begin / _end / _launch), what a host without torch binds."""
    _needs_device()
    import torch

    import newton_amd as nt
    import scenes

    if scene == "quadruped":
        model = scenes.quadruped_scene(64, seed=3, device="cuda:0")
        pipe = nt.CollisionPipeline(model)
    else:
        from sdf_pipeline_checker import sdf_scene

        model = sdf_scene(4, 6, device="cuda:0", walls=True, seed=9)
        pipe = nt.CollisionPipeline(model, broad_phase="sap")
    dt, substeps, frames = 1.0e-3, 4, 3
    out = {}
    for mode in ("eager", "graph"):
        solver = nt.solvers.SolverXPBD(model, iterations=2)
        st, ctrl, contacts = [model.state(), model.state()], model.control(), pipe.contacts()
        simulate = _frame(model, pipe, solver, st, ctrl, contacts, dt, substeps)
        if mode == "eager":
            for _ in range(frames):
                simulate()
        else:
            if backend == "torch":
                g = nt.graph.capture(simulate, warmup=0)  # the capture pass itself is frame 1 (it launches nothing: replay does)
                n_replays = frames
            else:  # the ABI helper runs the frame once for real before recording it (kernel attributes are set on first use)
                g = nt.graph.capture(simulate, warmup=1, backend="abi", contacts=contacts)
                n_replays = frames - 1
            for _ in range(n_replays):
                g.launch()
        torch.cuda.synchronize()
        out[mode] = (st[0].body_q.cpu().numpy().copy(), st[0].body_qd.cpu().numpy().copy())
    assert np.isfinite(out["eager"][0]).all()
    assert np.abs(out["eager"][1]).max() > 0.0
    assert np.array_equal(out["eager"][0], out["graph"][0]) and np.array_equal(out["eager"][1], out["graph"][1])
