"""Device checks for the features added after the last on-hardware run of the suite (they are covered on the CPU by the emulated
kernels, tests/test_emu_parity.py and tests/emu/run_gpu_tests_emulated.py); kept in one late-sorting file so that the order of
the on-device run is: long-validated tests first, newest last."""
import numpy as np
import pytest

from pair_scenes import RECENT_CONVEX_CASES
from test_gpu_parity_xpbd import _compare_contacts, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(RECENT_CONVEX_CASES))
def test_recent_convex_pair_contacts(name):
    """Cylinder / cone rolling on a hull face: the axial post-processing treats convex meshes as discrete shapes
    (collide.py contact reduction for curved-vs-polyhedral pairs)."""
    import newton_amd as nt
    from oracle_bridge import Oracle
    from pair_scenes import CONVEX_CASES, pair_model

    model = pair_model(CONVEX_CASES[name], device="cuda:0")
    o = Oracle(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    pipe.collide(model.state(), contacts)
    oc = o.contacts()
    pairs, _, _ = o.collide(model.body_q, oc)
    _compare_contacts(model, contacts, oc, pairs)


def test_deterministic_mode_orders_contacts_by_sort_key():
    """CollisionPipeline(deterministic=True): the flat contact arrays are sorted by the reference's contact key
    (shape0, shape1, sub-contact index: contact_data.py:60-90); same contacts as the default append order."""
    from scenes import mixed_primitive_scene

    nt, model, o = _setup(mixed_primitive_scene, 6)
    s0 = model.state()
    plain, det = nt.CollisionPipeline(model), nt.CollisionPipeline(model, deterministic=True)
    c0, c1 = plain.contacts(), det.contacts()
    plain.collide(s0, c0)
    det.collide(s0, c1)
    n = int(c0.rigid_contact_count.cpu().numpy()[0])
    assert n > 20 and int(c1.rigid_contact_count.cpu().numpy()[0]) == n
    a0, a1 = c0.rigid_contact_shape0.cpu().numpy()[:n], c0.rigid_contact_shape1.cpu().numpy()[:n]
    order = np.lexsort((np.arange(n), a1, a0))  # stable: sub-contact order inside a pair is the export order
    assert not np.array_equal(order, np.arange(n))  # the append order (analytic first, then convex) is a different order
    for name in ("shape0", "shape1", "point0", "point1", "normal", "margin0"):
        want = getattr(c0, "rigid_contact_" + name).cpu().numpy()[:n][order]
        assert np.array_equal(getattr(c1, "rigid_contact_" + name).cpu().numpy()[:n], want), name
    key = c1.rigid_contact_shape0.cpu().numpy()[:n].astype(np.int64) * (1 << 20) + c1.rigid_contact_shape1.cpu().numpy()[:n]
    assert np.all(np.diff(key) >= 0)


def test_eval_fk_selection_on_device():
    """Same selections through the device path: full FK into a scratch State (one eval_fk_kernel launch), then the selected
    body rows are copied."""
    import newton_amd as nt

    env = nt.ModelBuilder()
    for k in range(2):
        link = env.add_link(is_kinematic=(k == 1), mass=1.0)
        env.add_shape_box(link, hx=0.1, hy=0.1, hz=0.1)
        j = env.add_joint_revolute(-1, link, axis=(0, 0, 1), parent_xform=[float(k), 0.0, 0.0, 0.0, 0.0, 0.0, 1.0],
                                   child_xform=[-0.5, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
        env.add_articulation([j])
    scene = nt.ModelBuilder()
    scene.replicate(env, 5)
    model = scene.finalize(device="cuda:0")
    rng = np.random.default_rng(0)
    q = rng.uniform(-1.0, 1.0, size=10).astype(np.float32)
    qd = rng.normal(size=10).astype(np.float32)
    full_q, full_qd = nt.articulation.eval_fk_numpy(model, q, qd)
    base = model.state()
    base_q = base.body_q.cpu().numpy().copy()
    mask = rng.random(10) < 0.5
    mask[0], mask[1] = True, False
    for kw, sel in ((dict(mask=mask), mask), (dict(indices=np.flatnonzero(mask)), mask),
                    (dict(body_flag_filter=nt.BodyFlags.KINEMATIC), np.arange(10) % 2 == 1)):
        s = model.state()
        nt.eval_fk(model, q, qd, s, **kw)
        got_q, got_qd = s.body_q.cpu().numpy(), s.body_qd.cpu().numpy()
        assert np.allclose(got_q[sel], full_q[sel], atol=1e-6) and np.allclose(got_qd[sel], full_qd[sel], atol=1e-5)
        assert np.array_equal(got_q[~sel], base_q[~sel])


def _fuzz_names():
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import fuzz_reference_cases as fc

    return [f"fuzz/{('xpbd' if i % 10 < 6 else 'semi' if i % 10 < 8 else 'fs')}_{fc.SEED0 + i}" for i in range(fc.N_CASES)]


@pytest.mark.parametrize("name", _fuzz_names())
def test_hip_path_against_fuzz_reference_vectors(name):
    """The HIP path against the executed reference on the seeded RANDOM scenes of tests/golden/fuzz_reference_cases.py (the checker's
    twin: tests/test_fuzz_reference_vectors.py), teacher-forced step by step.  Random joint trees over every joint type, mixed
    colliders, random solver options, speeds of several hundred rad/s: contact counts exact, state within the single-step contract
    scaled by the state's own magnitude (a nearly satisfied joint row amplifies rounding in later iterations; XPBD velocities are
    position differences / dt)."""
    import os

    import torch

    import fuzz_reference_cases as fc
    import newton_amd as nt
    import reference_cases as rc
    from test_zx_round2_gpu import _to_device

    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_reference_vectors.npz"))
    if name in {s.split(":")[0] for s in ref["skipped"].tolist()}:
        pytest.skip("the reference run needs an un-vendored Warp builtin")
    case = fc.cases()[name]
    kind = case.get("solver", "xpbd")
    case_dev = dict(case)
    scene = case["scene"]
    case_dev["scene"] = lambda: _to_device(scene())
    model = rc.prepare(case_dev)
    if kind == "xpbd":
        solver = nt.solvers.SolverXPBD(model, **case["kw"])
    elif kind == "semi_implicit":
        solver = nt.solvers.SolverSemiImplicit(model, **case["kw"])
    else:
        solver = nt.solvers.SolverFeatherstone(model, mass_matrix="dense", **case["kw"])
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s0, s1 = model.state(), model.state()
    worst = np.zeros(4)
    for k in range(case["steps"]):
        s0.body_q, s0.body_qd = torch.from_numpy(ref[f"{name}/body_q{k}"]), torch.from_numpy(ref[f"{name}/body_qd{k}"])
        if kind == "featherstone":
            s0.joint_q, s0.joint_qd = torch.from_numpy(ref[f"{name}/joint_q{k}"]), torch.from_numpy(ref[f"{name}/joint_qd{k}"])
        s0.clear_forces()
        pipe.collide(s0, contacts)
        assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == int(ref[f"{name}/contacts{k}"][0])
        solver.step(s0, s1, None, contacts, case["dt"])
        torch.cuda.synchronize()
        q, qd = s1.body_q.cpu().numpy(), s1.body_qd.cpu().numpy()
        q_ref, qd_ref = ref[f"{name}/body_q{k + 1}"], ref[f"{name}/body_qd{k + 1}"]
        rot = np.minimum(np.abs(q[:, 3:] - q_ref[:, 3:]).max(axis=1), np.abs(q[:, 3:] + q_ref[:, 3:]).max(axis=1)).max()
        worst = np.maximum(worst, [np.abs(q[:, :3] - q_ref[:, :3]).max(), rot, np.abs(qd[:, :3] - qd_ref[:, :3]).max(),
                                   np.abs(qd[:, 3:] - qd_ref[:, 3:]).max()])
    vmax = max(1.0, float(np.abs(ref[f"{name}/body_qd{case['steps']}"]).max()))
    print(name, "HIP vs reference run: pos %.3g rot %.3g lin vel %.3g ang vel %.3g (max |qd| %.3g)" % (*worst, vmax))
    pos_tol = 5e-5 if kind == "xpbd" else 1e-5
    vel_tol = (5e-5 / case["dt"] if kind == "xpbd" else 1e-3) * vmax
    assert worst[0] <= pos_tol and worst[1] <= pos_tol and worst[2] <= vel_tol and worst[3] <= 10.0 * vel_tol, worst
