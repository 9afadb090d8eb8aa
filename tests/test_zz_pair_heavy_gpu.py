"""Pair-heavy scenes on the GPU (config C5 without the SDF / hydroelastic contact models): 64 convex hulls in a five-wall bin =
2 336 candidate pairs per environment; the per-contact solver records live in HBM (nt_model.contact_scratch_in_hbm) and every
environment gets its own workgroup.  HIP vs the oracle, plus fused rollout == launch-by-launch loop.

The kernels' logic for this mode is covered on the CPU by tests/test_emu_parity.py (same sources, emulated); this file is the
on-device check and deliberately sorts last in the suite."""
import numpy as np
import pytest

from test_gpu_parity_xpbd import _compare_contacts, _rel, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_hulls,n_env", [(40, 5), (64, 3)])
def test_hull_bin_collide_step_and_rollout(n_hulls, n_env):
    from oracle_bridge import OracleState
    from scenes import hull_bin_scene

    nt, model, o = _setup(hull_bin_scene, n_env, n_hulls=n_hulls)
    assert model.device_model().desc.contact_scratch_in_hbm == 1
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    s0, s1 = model.state(), model.state()
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    dt = 1.0 / 600.0
    pipe.collide(s0, contacts)
    pairs, _, _ = o.collide(os0.body_q, oc)
    _compare_contacts(model, contacts, oc, pairs)
    assert int(oc.count[0]) > 5 * n_hulls * n_env
    solver.step(s0, s1, None, contacts, dt)
    o.xpbd_step(os0, os1, o.control(), oc, dt)
    assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 1e-5
    assert _rel(s1.body_qd.cpu().numpy(), os1.body_qd) <= 2e-4
    # fused rollout == launch-by-launch loop, bitwise
    r0, r1 = model.state(), model.state()
    out = solver.rollout(r0, r1, None, contacts, dt, 3)
    a, b = model.state(), model.state()
    for _ in range(3):
        a.clear_forces()
        pipe.collide(a, contacts)
        solver.step(a, b, None, contacts, dt)
        a, b = b, a
    assert np.array_equal(out.body_q.cpu().numpy(), a.body_q.cpu().numpy())


def test_scene_too_large_for_lds_is_rejected_with_numbers():
    from scenes import hull_bin_scene

    import newton_amd as nt

    model = hull_bin_scene(1, 150, device="cuda:0")  # 11 925 pairs: the topology tables alone exceed the LDS
    with pytest.raises(NotImplementedError, match="KB of LDS"):
        nt.CollisionPipeline(model)
    with pytest.raises(NotImplementedError, match="candidate pairs"):
        nt.solvers.SolverXPBD(model)
