"""ViewerFile on device States (newton/_src/viewer/viewer_file.py:1176-1260 record, :1479-1533 playback):
record the frames of a quadruped rollout on the MI355X, save / load the recording, replay frames into fresh device States -- the
replayed state is bit-identical to the recorded one, and stepping from a replayed frame reproduces the recorded continuation bit
for bit (the checkpoint use of the recorder for parity debugging)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_record_save_load_replay_on_device_states(tmp_path):
    import torch

    import newton_amd as nt
    from scenes import quadruped_scene
    from test_gpu_parity_xpbd import _lower_quadrupeds

    E, frames, dt = 48, 6, 1e-3
    model = quadruped_scene(E, device="cuda:0", seed=4)
    _lower_quadrupeds(nt, model, 0.25)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    s0, s1 = model.state(), model.state()
    path = str(tmp_path / "run.npz")
    rec = nt.viewer.ViewerFile(path, auto_save=True)
    rec.set_model(model)
    live = []
    for f in range(frames):
        rec.begin_frame(f * 10 * dt)
        out = solver.rollout(s0, s1, None, contacts, dt, 10)  # even substep count: the result is back in s0
        assert out is s0
        rec.log_state(s0)
        rec.end_frame()
        live.append({k: getattr(s0, k).cpu().numpy().copy() for k in ("body_q", "body_qd", "joint_q", "joint_qd")})
    rec.close()  # auto_save
    torch.cuda.synchronize()
    assert rec.get_frame_count() == frames and rec.has_model()

    play = nt.viewer.ViewerFile()
    play.load_recording(path)
    assert play.get_frame_count() == frames
    a = model.state()
    for f in (0, frames - 2, frames - 1):
        play.load_state(a, f)
        for k, want in live[f].items():
            assert np.array_equal(getattr(a, k).cpu().numpy(), want), (f, k)
    with pytest.raises(IndexError):
        play.load_state(a, frames)

    # resume from a replayed frame: one more frame == the recorded next frame, bit for bit
    b = model.state()
    play.playback(a, frames - 2)
    c2 = pipe.contacts()
    out = solver.rollout(a, b, None, c2, dt, 10)
    for k, want in live[frames - 1].items():
        assert np.array_equal(getattr(out, k).cpu().numpy(), want), k

    # ring-buffer recording keeps the newest frames
    ring = nt.viewer.ViewerFile(max_history_size=2)
    for f in range(4):
        play.load_state(a, f)
        ring.begin_frame(float(f))
        ring.log_state(a)
        ring.end_frame()
    assert ring.get_frame_count() == 2
    ring.load_state(b, 1)
    assert np.array_equal(b.body_q.cpu().numpy(), live[3]["body_q"])
