"""Round-2 device tests of host-visible behaviour fixed after the advisor's review: Contacts.force rows follow the
deterministic (key-sorted) contact order, runtime flag edits reach the device through notify_model_changed, Control.clear
matches the reference (control.py:76-105)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_contact_force_rows_follow_the_deterministic_contact_order():
    """CollisionPipeline(deterministic=True) sorts the flat contact arrays by (shape0, shape1); Contacts.force[i] must be
    the force of rigid_contact_shape0/1[i] (the reference sorts before the solver, so its indices agree by construction)."""
    import torch
    from scenes import mixed_primitive_scene

    import newton_amd as nt

    def run(deterministic):
        model = mixed_primitive_scene(6, device="cuda:0")
        model.request_contact_attributes("force")
        pipe = nt.CollisionPipeline(model, deterministic=deterministic)
        contacts = pipe.contacts()
        solver = nt.solvers.SolverXPBD(model, iterations=2)
        s0, s1 = model.state(), model.state()
        for _ in range(150):  # 0.15 s: every body has fallen its 2-10 cm and presses on the ground
            s0.clear_forces()
            pipe.collide(s0, contacts)
            solver.step(s0, s1, None, contacts, 1e-3)
            s0, s1 = s1, s0
        solver.update_contacts(contacts)
        torch.cuda.synchronize()
        n = int(contacts.rigid_contact_count.cpu().numpy()[0])
        return (contacts.rigid_contact_shape0.cpu().numpy()[:n], contacts.rigid_contact_shape1.cpu().numpy()[:n],
                contacts.rigid_contact_point0.cpu().numpy()[:n], contacts.force.cpu().numpy()[:n])

    a0, a1, ap, af = run(False)
    d0, d1, dp, df = run(True)
    assert len(a0) == len(d0) > 0 and np.abs(af).max() > 0.0
    key = d0.astype(np.int64) * (1 << 32) + d1
    assert np.all(np.diff(key) >= 0) and not np.array_equal(a0, d0)  # really re-ordered (envs interleave per pair)
    # match rows through (shape0, shape1, point0), which identify a contact uniquely
    order = np.lexsort((np.arange(len(a0)), a1, a0))  # stable sort of the raw rows by key == the deterministic order
    assert np.array_equal(a0[order], d0) and np.array_equal(a1[order], d1) and np.array_equal(ap[order], dp)
    assert np.array_equal(af[order], df)


def test_runtime_flag_edits_reach_the_device():
    """Disabling a joint at run time + notify_model_changed(JOINT_PROPERTIES): the XPBD joint rows of that joint stop."""
    from scenes import pendulum_scene

    import newton_amd as nt

    def run(disable):
        model = pendulum_scene(4, device="cuda:0", seed=3)
        solver = nt.solvers.SolverXPBD(model, iterations=2)
        s0, s1 = model.state(), model.state()
        solver.step(s0, s1, None, None, 1e-3)
        if disable:
            model.joint_enabled[1::2] = False  # second joint of every pendulum
            solver.notify_model_changed(nt.ModelFlags.JOINT_PROPERTIES)
        for _ in range(300):
            solver.step(s1, s0, None, None, 1e-3)
            s0, s1 = s1, s0
        return s1.body_q.cpu().numpy().reshape(4, 2, 7)

    from newton_amd import _np_math as nm

    def anchor_gap(q):  # distance between link 0's far end and link 1's near end (the second joint's two anchors)
        out = []
        for e in range(q.shape[0]):
            a = q[e, 0, :3] + nm.quat_rotate(q[e, 0, 3:], np.array([1.0, 0.0, 0.0]))
            b = q[e, 1, :3] + nm.quat_rotate(q[e, 1, 3:], np.array([-1.0, 0.0, 0.0]))
            out.append(np.linalg.norm(a - b))
        return np.array(out)

    held, free = anchor_gap(run(False)), anchor_gap(run(True))
    assert np.all(held < 5e-3), held       # the enabled joint keeps its anchors together
    assert np.all(free > 2e-2), free       # disabled on the device too: the second link drifts away in free fall


def test_control_clear_matches_reference_semantics():
    from scenes import quadruped_scene

    model = quadruped_scene(3, device="cuda:0")
    ctrl = model.control()
    ctrl.joint_f = np.ones(model.joint_dof_count, dtype=np.float32)
    ctrl.joint_target_qd = np.ones(model.joint_dof_count, dtype=np.float32)
    ctrl.clear(model)
    assert float(ctrl.joint_f.abs().max()) == 0.0 and float(ctrl.joint_target_qd.abs().max()) == 0.0
    assert np.array_equal(ctrl.joint_target_q.cpu().numpy(), model.joint_target_q)
    ctrl.clear()
    assert float(ctrl.joint_target_q.abs().max()) == 0.0


def test_contact_matcher_follows_the_reference_matcher_on_a_box_stack(oracle_lib):
    """ContactMatcher (device kernels nt_contacts_match / nt_contacts_save_history on the fixed slots, flat indices in the
    deterministic export order) against oracle/oracle_match.py on three consecutive frames of a settling box stack."""
    import os
    import sys

    import torch
    from scenes import box_stack_scene

    import newton_amd as nt

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_match as O

    model = box_stack_scene(8, n_boxes=4, seed=2, jitter=2e-3, device="cuda:0")
    pipe = nt.CollisionPipeline(model, deterministic=True)
    contacts = pipe.contacts()
    matcher = nt.ContactMatcher(model)
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    s0, s1 = model.state(), model.state()
    shape_body = np.asarray(model.shape_body)
    prev = None
    for frame in range(3):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        got = matcher.match(s0, contacts).cpu().numpy()
        torch.cuda.synchronize()
        n = int(contacts.rigid_contact_count.cpu().numpy()[0])
        sh0 = contacts.rigid_contact_shape0.cpu().numpy()[:n]
        sh1 = contacts.rigid_contact_shape1.cpu().numpy()[:n]
        p0 = contacts.rigid_contact_point0.cpu().numpy()[:n]
        p1 = contacts.rigid_contact_point1.cpu().numpy()[:n]
        nrm = contacts.rigid_contact_normal.cpu().numpy()[:n]
        assert len(got) == n > 0
        pair = sh0.astype(np.int64) * (1 << 32) + sh1
        assert np.all(np.diff(pair) >= 0)
        sub = np.zeros(n, dtype=np.int64)  # rank inside the pair's run == manifold slot order
        for i in range(1, n):
            sub[i] = sub[i - 1] + 1 if pair[i] == pair[i - 1] else 0
        keys = np.array([O.sort_key(a, b, k) for a, b, k in zip(sh0, sh1, sub)], dtype=np.int64)
        mid = O.midpoints(s0.body_q.cpu().numpy(), shape_body, sh0, sh1, p0, p1)
        if prev is None:
            assert np.all(got == -1)
        else:
            want = O.match(keys, mid, nrm, *prev)
            assert np.array_equal(got, want)
            assert (want >= 0).sum() > 0.5 * n
        matcher.save_sorted_state(s0, contacts)
        prev = (keys, mid, nrm)
        solver.step(s0, s1, None, contacts, 1.0 / 240.0)
        s0, s1 = s1, s0
    # a reset world forgets its history
    s0.clear_forces()
    pipe.collide(s0, contacts)
    mask = np.zeros(8, dtype=bool)
    mask[3] = True
    matcher.reset(mask)
    got = matcher.match(s0, contacts).cpu().numpy()
    n = len(got)
    sw = np.asarray(model.shape_world)
    world = np.maximum(sw[contacts.rigid_contact_shape0.cpu().numpy()[:n]], sw[contacts.rigid_contact_shape1.cpu().numpy()[:n]])
    assert np.all(got[world == 3] == -1) and np.any(got[world != 3] >= 0)


def test_descriptor_built_by_the_c_helper_steps_like_the_python_built_one():
    """nt_model_create(on_device=1) from the flat Model arrays: a fused rollout through its descriptor is bitwise the rollout
    through the descriptor the Python host builds (same tables, same device kernels)."""
    import ctypes as C

    import torch
    from scenes import quadruped_scene
    from test_model_build import _newton_arrays

    import newton_amd as nt
    from newton_amd import _lib

    model = quadruped_scene(96, device="cuda:0", seed=4)
    model.joint_q.reshape(96, -1)[:, 2] -= 0.2
    model.body_q, model.body_qd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    dm = model.device_model()

    def run(desc):
        s0, s1 = model.state(), model.state()
        ctrl = model.control()
        p = solver._params()
        d0, d1, d_c, d_ct = s0._desc(), s1._desc(), ctrl._desc(), contacts._desc()
        for _ in range(3):
            _lib.check(dm.lib.nt_xpbd_rollout(C.byref(desc), C.byref(p), None, C.byref(d0), C.byref(d1), C.byref(d_c),
                                              C.byref(d_ct), 1e-3, 10, dm.stream()), "nt_xpbd_rollout")
        torch.cuda.synchronize()
        return s0.body_q.cpu().numpy().copy(), s0.body_qd.cpu().numpy().copy()

    q_ref, qd_ref = run(dm.desc)
    src, keep = _newton_arrays(model)
    h = C.c_void_p()
    assert dm.lib.nt_model_create(C.byref(src), 1, C.byref(h)) == 0, dm.lib.nt_model_last_error()
    try:
        mine = dm.lib.nt_model_get(h).contents
        assert mine.params_uniform == dm.desc.params_uniform and mine.np == dm.desc.np
        q, qd = run(mine)
    finally:
        dm.lib.nt_model_destroy(h)
    assert np.array_equal(q.view(np.int32), q_ref.view(np.int32)) and np.array_equal(qd.view(np.int32), qd_ref.view(np.int32))
    assert np.abs(qd_ref).max() > 0.0


_REF_NAMES = ["quadruped_standing", "quadruped_impact_restitution", "pendulum", "joint_zoo", "joint_zoo_free_root",
              "box_stack_no_weighting", "box_stack_sunk_restitution", "quadruped_velocity_from_delta",
              "box_stack_velocity_from_delta_restitution", "semi/pendulum", "semi/joint_zoo", "semi/box_stack",
              "semi/quadruped", "fs/pendulum", "fs/joint_zoo", "fs/joint_zoo_free_root", "fs/quadruped", "fs/quadruped_interval3",
              "fs/joint_zoo_interval2", "fs/free_child", "fs/free_child_free_root"]


@pytest.mark.parametrize("name", _REF_NAMES)
def test_hip_path_against_reference_vectors(name):
    """The HIP path (collide + step through the C ABI) against vectors recorded from the REFERENCE's own solver source
    (tests/golden/make_xpbd_reference_vectors.py), teacher-forced step by step.  The C++ checker reproduces these vectors bit
    for bit (tests/test_reference_vectors.py); the device path differs by its documented reorderings (world-frame inverse
    inertia precomputed per body, two lanes per joint, OCML libm), hence tolerances instead of equality."""
    import os
    import sys

    import torch

    import newton_amd as nt

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import reference_cases as rc

    case = rc.cases()[name]
    ref = np.load(os.path.join(here, "golden", "xpbd_reference_vectors.npz"))
    host = rc.prepare(case)
    kind = case.get("solver", "xpbd")
    # the same model on the device
    case_dev = dict(case)
    scene = case["scene"]
    case_dev["scene"] = lambda: _to_device(scene())
    model = rc.prepare(case_dev)
    if kind == "xpbd":
        solver = nt.solvers.SolverXPBD(model, **case["kw"])
        for k_, v_ in case.get("attrs", {}).items():
            assert hasattr(solver, k_), k_
            setattr(solver, k_, v_)
    elif kind == "semi_implicit":
        solver = nt.solvers.SolverSemiImplicit(model, **case["kw"])
    else:
        solver = nt.solvers.SolverFeatherstone(model, **case["kw"])
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
        n = int(contacts.rigid_contact_count.cpu().numpy()[0])
        assert n == int(ref[f"{name}/contacts{k}"][0])
        solver.step(s0, s1, None, contacts, case["dt"])
        torch.cuda.synchronize()
        q, qd = s1.body_q.cpu().numpy(), s1.body_qd.cpu().numpy()
        q_ref, qd_ref = ref[f"{name}/body_q{k + 1}"], ref[f"{name}/body_qd{k + 1}"]
        rot = np.minimum(np.abs(q[:, 3:] - q_ref[:, 3:]).max(axis=1), np.abs(q[:, 3:] + q_ref[:, 3:]).max(axis=1)).max()
        worst = np.maximum(worst, [np.abs(q[:, :3] - q_ref[:, :3]).max(), rot, np.abs(qd[:, :3] - qd_ref[:, :3]).max(),
                                   np.abs(qd[:, 3:] - qd_ref[:, 3:]).max()])
    print(name, "HIP vs reference run: pos %.3g rot %.3g lin vel %.3g ang vel %.3g" % tuple(worst))
    scale = max(1.0, float(np.abs(ref[f"{name}/body_qd0"]).max()))
    # one step from identical inputs: positions to a few fp32 ulps of a metre; XPBD velocities are position differences / dt
    vel_tol = 4e-6 / case["dt"] if kind == "xpbd" else 2e-5 * scale
    assert worst[0] <= 4e-6 and worst[1] <= 4e-6 and worst[2] <= vel_tol and worst[3] <= 10.0 * vel_tol, worst


def _to_device(host_model):
    """Re-finalize is not needed: the scene factories take `device`; this helper exists for factories called without it."""
    import copy

    m = copy.copy(host_model)
    m.device = "cuda:0"
    m._dev = None
    return m


_COLLIDE_NAMES = ["hull_bin_a", "hull_bin_b", "pair_matrix_a", "pair_matrix_b", "mixed_primitives_a", "mixed_primitives_b", "mixed_primitives_c", "box_stack_a",
                  "box_stack_b", "quadruped_cylinders", "quadruped_box_feet"]


@pytest.mark.parametrize("name", _COLLIDE_NAMES)
def test_hip_collide_against_reference_collision_vectors(name):
    """CollisionPipeline.collide on the device against the contact arrays the REFERENCE's collision kernels produced
    (tests/golden/make_collide_reference_vectors.py): same contacts in the same append order, geometry within 1e-5."""
    import os
    import sys

    import torch

    import newton_amd as nt

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import collide_cases as cc

    ref = np.load(os.path.join(here, "golden", "collide_reference_vectors.npz"))
    host, _ = cc.cases()[name]()
    model = _to_device(host)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s = model.state()
    s.body_q = torch.from_numpy(ref[f"{name}/body_q"])
    pipe.collide(s, contacts)
    torch.cuda.synchronize()
    n = int(ref[f"{name}/count"][0])
    assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == n
    get = lambda k: getattr(contacts, "rigid_contact_" + k).cpu().numpy()[:n]  # noqa: E731
    assert np.array_equal(get("shape0"), ref[f"{name}/shape0"]) and np.array_equal(get("shape1"), ref[f"{name}/shape1"])
    err = {k: float(np.abs(get(k) - ref[f"{name}/{k}"]).max()) for k in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")}
    print(name, "HIP collide vs reference kernels:", {k: float("%.3g" % v) for k, v in err.items()})
    assert all(v <= 1e-5 for v in err.values()), err


@pytest.mark.parametrize("name", ["barrel_wide", "barrel_tight"])
def test_hip_collide_against_reference_vectors_of_barrel_cylinders(name):
    """Barrel cylinders on the device against the executed reference (make_collide_reference_vectors.py --barrel).  A plane pair of a
    barrel is analytic only while it rests on an end cap (narrow_phase.py:682-686): the tile stores it with the convex pairs and
    decides per substep, so its analytic contacts are exported inside the convex group -- compared pair by pair (the order INSIDE a
    pair is the reference's); CollisionPipeline(deterministic=True) gives the reference's sorted order either way."""
    import os
    import sys

    import torch

    import newton_amd as nt

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import collide_cases as cc

    ref = np.load(os.path.join(here, "golden", "collide_barrel_reference_vectors.npz"))
    host, _ = cc.barrel_cases()[name]()
    model = _to_device(host)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s = model.state()
    s.body_q = torch.from_numpy(ref[f"{name}/body_q"])
    pipe.collide(s, contacts)
    torch.cuda.synchronize()
    n = int(ref[f"{name}/count"][0])
    assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == n
    get = lambda k: getattr(contacts, "rigid_contact_" + k).cpu().numpy()[:n]  # noqa: E731
    key = lambda a, b: np.lexsort((np.arange(n), b, a))  # noqa: E731  (stable: by pair, append order inside the pair)
    mine, theirs = key(get("shape0"), get("shape1")), key(ref[f"{name}/shape0"], ref[f"{name}/shape1"])
    assert np.array_equal(get("shape0")[mine], ref[f"{name}/shape0"][theirs]) and np.array_equal(get("shape1")[mine], ref[f"{name}/shape1"][theirs])
    err = {k: float(np.abs(get(k)[mine] - ref[f"{name}/{k}"][theirs]).max()) for k in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")}
    print(name, "HIP collide vs reference kernels (barrel cylinders):", {k: float("%.3g" % v) for k, v in err.items()})
    assert all(v <= 1e-5 for v in err.values()), err


def test_barrel_cylinders_fused_rollout_tracks_the_oracle_and_equals_the_loop():
    """The barrel scene through SolverXPBD: the fused rollout is bitwise the collide / step loop and follows the oracle (whose contact
    order is the reference's; the on-cap contacts of a barrel sit in the convex group on the device: summation order only)."""
    import os
    import sys

    import torch

    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import collide_cases as cc

    host, _ = cc.barrel_cases()["barrel_wide"]()
    dt, n = 1e-3, 12

    def device(fused):
        model = _to_device(host)
        pipe = nt.CollisionPipeline(model)
        contacts = pipe.contacts()
        solver = nt.solvers.SolverXPBD(model, iterations=2)
        a, b = model.state(), model.state()
        if fused:
            out = solver.rollout(a, b, None, contacts, dt, n)
        else:
            for _ in range(n):
                a.clear_forces()
                pipe.collide(a, contacts)
                solver.step(a, b, None, contacts, dt)
                a, b = b, a
            out = a
        torch.cuda.synchronize()
        return out.body_q.cpu().numpy().copy(), out.body_qd.cpu().numpy().copy()

    fq, fv = device(True)
    lq, lv = device(False)
    assert np.array_equal(fq, lq) and np.array_equal(fv, lv)
    o = Oracle(host)
    a, b = OracleState(host), OracleState(host)
    for _ in range(n):
        ct = o.contacts()
        o.collide(a.body_q, ct)
        o.xpbd_step(a, b, o.control(), ct if ct.count[0] else None, dt, iterations=2)
        a, b = b, a
    dq = float(np.abs(fq - a.body_q).max())
    print("barrel scene, 12 substeps: max |dq| vs the oracle", dq)
    assert dq <= 1e-4


def test_featherstone_rollout_with_a_mass_matrix_interval_is_the_step_loop():
    """update_mass_matrix_interval = 3: the fused rollout (substep index inside the launch) and the launch-by-launch loop
    rebuild the mass matrix on the same steps and agree bit for bit; interval 1 gives a different (fresher) result."""
    import torch
    from scenes import quadruped_scene

    import newton_amd as nt

    def run(interval, fused):
        model = quadruped_scene(16, device="cuda:0", seed=3)
        solver = nt.solvers.SolverFeatherstone(model, update_mass_matrix_interval=interval)
        pipe = nt.CollisionPipeline(model)
        contacts = pipe.contacts()
        s0, s1 = model.state(), model.state()
        if fused:
            out = solver.rollout(s0, s1, None, contacts, 1e-3, 7)
            out = solver.rollout(out, s1 if out is s0 else s0, None, contacts, 1e-3, 4)
        else:
            for _ in range(11):
                s0.clear_forces()
                pipe.collide(s0, contacts)
                solver.step(s0, s1, None, contacts, 1e-3)
                s0, s1 = s1, s0
            out = s0
        torch.cuda.synchronize()
        return out.joint_q.cpu().numpy().copy(), out.joint_qd.cpu().numpy().copy()

    q_loop, qd_loop = run(3, False)
    q_fused, qd_fused = run(3, True)
    assert np.array_equal(q_loop.view(np.int32), q_fused.view(np.int32)) and np.array_equal(qd_loop.view(np.int32), qd_fused.view(np.int32))
    q1, qd1 = run(1, True)
    assert not np.array_equal(qd1, qd_fused) and np.abs(qd1 - qd_fused).max() < 1e-2


@pytest.mark.parametrize("free_root", [False, True])
def test_featherstone_free_and_distance_joints_below_the_root_on_device(oracle_lib, free_root):
    """Descendant FREE / DISTANCE joints (solver_featherstone.py:229-265,1006-1046) through the Python surface: 20 steps against
    the checker, and the fused rollout against the launch-by-launch loop bit for bit."""
    import torch
    from oracle_bridge import Oracle, OracleState
    from scenes import free_child_scene
    from tolerances import record

    import newton_amd as nt

    model = free_child_scene(8, device="cuda:0", seed=33, free_root=free_root)
    solver = nt.solvers.SolverFeatherstone(model, angular_damping=0.05)
    s0, s1 = model.state(), model.state()
    o = Oracle(model)
    os0, os1 = OracleState(model), OracleState(model)
    for _ in range(20):
        s0.clear_forces()
        solver.step(s0, s1, None, None, 1e-3)
        os0.body_f[:] = 0
        o.featherstone_step(os0, os1, o.control(), None, 1e-3, angular_damping=0.05)
        s0, s1, os0, os1 = s1, s0, os1, os0
    torch.cuda.synchronize()
    errs = {k: float(np.abs(getattr(s0, k).cpu().numpy().reshape(-1) - getattr(os0, k).reshape(-1)).max())
            for k in ("joint_q", "joint_qd", "body_q", "body_qd")}
    record("featherstone_free_child" + ("_free_root" if free_root else ""), errs,
           {"joint_q": 1e-5, "body_q": 1e-5, "joint_qd": 1e-4, "body_qd": 1e-4})
    assert errs["joint_q"] < 1e-5 and errs["body_q"] < 1e-5 and errs["joint_qd"] < 1e-4 and errs["body_qd"] < 1e-4, errs
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    a, b = model.state(), model.state()
    out = solver.rollout(a, b, None, contacts, 1e-3, 5)
    c, d = model.state(), model.state()
    for _ in range(5):
        c.clear_forces()
        pipe.collide(c, contacts)
        solver.step(c, d, None, contacts, 1e-3)
        c, d = d, c
    torch.cuda.synchronize()
    for k in ("joint_q", "joint_qd", "body_q", "body_qd"):
        assert np.array_equal(getattr(out, k).cpu().numpy().view(np.int32), getattr(c, k).cpu().numpy().view(np.int32)), k


def test_collision_pipeline_contact_matching_latest_with_report():
    """CollisionPipeline(contact_matching="latest", contact_report=True) (collide.py:1126-1129): rigid_contact_match_index is
    filled by every collide(); new / broken reports are consistent with it; the indices equal oracle_match on the exported rows."""
    import os
    import sys

    import torch
    from scenes import box_stack_scene

    import newton_amd as nt

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_match as O

    model = box_stack_scene(6, n_boxes=4, seed=2, jitter=4e-3, device="cuda:0")
    with pytest.raises(ValueError):
        nt.CollisionPipeline(model, contact_matching="always")
    with pytest.raises(ValueError):
        nt.CollisionPipeline(model, contact_report=True)
    pipe = nt.CollisionPipeline(model, contact_matching="latest", contact_report=True)
    assert pipe.deterministic
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    s0, s1 = model.state(), model.state()
    shape_body = np.asarray(model.shape_body)
    prev = None
    broken_total = 0
    for frame in range(5):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        torch.cuda.synchronize()
        n = int(contacts.rigid_contact_count.cpu().numpy()[0])
        got = contacts.rigid_contact_match_index.cpu().numpy()[:n]
        sh0, sh1 = contacts.rigid_contact_shape0.cpu().numpy()[:n], contacts.rigid_contact_shape1.cpu().numpy()[:n]
        pair = sh0.astype(np.int64) * (1 << 32) + sh1
        sub = np.zeros(n, dtype=np.int64)
        for i in range(1, n):
            sub[i] = sub[i - 1] + 1 if pair[i] == pair[i - 1] else 0
        keys = np.array([O.sort_key(a, b, k) for a, b, k in zip(sh0, sh1, sub)], dtype=np.int64)
        mid = O.midpoints(s0.body_q.cpu().numpy(), shape_body, sh0, sh1, contacts.rigid_contact_point0.cpu().numpy()[:n],
                          contacts.rigid_contact_point1.cpu().numpy()[:n])
        nrm = contacts.rigid_contact_normal.cpu().numpy()[:n]
        new_n = int(contacts.rigid_contact_new_count.cpu().numpy()[0])
        new = np.sort(contacts.rigid_contact_new_indices.cpu().numpy()[:new_n])
        assert np.array_equal(new, np.flatnonzero(got < 0))
        if prev is None:
            assert np.all(got == -1) and int(contacts.rigid_contact_broken_count.cpu().numpy()[0]) == 0
        else:
            want = O.match(keys, mid, nrm, *prev)
            assert np.array_equal(got, want)
            bn = int(contacts.rigid_contact_broken_count.cpu().numpy()[0])
            broken = np.sort(contacts.rigid_contact_broken_indices.cpu().numpy()[:bn])
            assert np.array_equal(broken, np.setdiff1d(np.arange(len(prev[0])), want[want >= 0]))
            broken_total += bn
        prev = (keys, mid, nrm)
        solver.step(s0, s1, None, contacts, 1.0 / 240.0)
        s0, s1 = s1, s0
    pipe.reset_contact_matching()
    s0.clear_forces()
    pipe.collide(s0, contacts)
    n = int(contacts.rigid_contact_count.cpu().numpy()[0])
    assert np.all(contacts.rigid_contact_match_index.cpu().numpy()[:n] == -1)


@pytest.mark.parametrize("name,frames", [("box_stack", 6), ("mixed_primitives", 8)])
def test_collision_pipeline_sticky_matching_against_the_reference_matcher(name, frames):
    """CollisionPipeline(contact_matching="sticky") on the device, frame by frame on the recorded states, against the arrays the
    reference ContactMatcher(sticky=True) left after match -> replay_matched (tests/golden/make_match_reference_vectors.py)."""
    import os

    import torch
    from scenes import box_stack_scene, mixed_primitive_scene

    import newton_amd as nt

    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "match_reference_vectors.npz"))
    host = box_stack_scene(1, n_boxes=4, seed=2, jitter=5e-3) if name == "box_stack" else mixed_primitive_scene(1, seed=4)
    model = _to_device(host)
    pipe = nt.CollisionPipeline(model, contact_matching="sticky")
    contacts = pipe.contacts()
    s = model.state()
    for k in range(frames):
        s.body_q = torch.from_numpy(ref[f"{name}/{k}/body_q"])
        pipe.collide(s, contacts)
        torch.cuda.synchronize()
        n = len(ref[f"{name}/{k}/keys"])
        assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == n
        assert np.array_equal(contacts.rigid_contact_match_index.cpu().numpy()[:n], ref[f"{name}/{k}/sticky_match"])
        for field in ("point0", "point1", "offset0", "offset1", "normal"):
            got = getattr(contacts, "rigid_contact_" + field).cpu().numpy()[:n]
            assert np.array_equal(got, ref[f"{name}/{k}/sticky_{field}"]), (k, field)


@pytest.mark.parametrize("variant", ["plain", "filtered", "immovable"])
@pytest.mark.parametrize("name", ["single_world", "multiple_worlds", "shape_flags", "per_shape_gap"])
def test_hip_broad_phase_against_the_reference_classes(name, variant):
    """BroadPhaseAllPairs / BroadPhaseSAP (device projection + in-LDS segment sort + sweep) / BroadPhaseExplicit on the device
    against the candidate lists the reference's own classes produced when executed
    (tests/golden/make_broadphase_reference_vectors.py) -- as sets: the wave-aggregated append has no fixed order."""
    import os
    import sys

    import torch

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import broadphase_cases as bc
    from test_broad_phase_standalone import _gpu_run

    from newton_amd import geometry

    ref = np.load(os.path.join(here, "golden", "broadphase_reference_vectors.npz"))
    v, key = bc.variants(name)[variant], f"{name}/{variant}"
    kw = dict(include_static_kinematic_pairs=v["include"])
    if v["shape_body"] is not None:
        kw.update(shape_body=v["shape_body"], body_flags=v["body_flags"])
    for cls_name, kind in (("BroadPhaseAllPairs", "nxn"), ("BroadPhaseSAP", "sap")):
        count, pairs, _ = _gpu_run(cls_name, v["lower"], v["upper"], v["gap"], v["group"], v["world"], v["flags"],
                                   filter_pairs=v["filter_pairs"], **kw)
        want = {tuple(p) for p in ref[f"{key}/{kind}_pairs"]}
        assert count == len(want) and {tuple(p) for p in pairs} == want, (cls_name, count, len(want))
    dev = "cuda:0"
    t = lambda a, d: torch.as_tensor(np.ascontiguousarray(a), dtype=d, device=dev)  # noqa: E731
    ep = v["explicit_pairs"]
    out = torch.full((len(ep) + 1, 2), -1, dtype=torch.int32, device=dev)
    cnt = torch.full((1,), 77, dtype=torch.int32, device=dev)
    extra = {k: t(x, torch.int32) if isinstance(x, np.ndarray) else x for k, x in kw.items()}
    geometry.BroadPhaseExplicit().launch(t(v["lower"], torch.float32), t(v["upper"], torch.float32), t(v["gap"], torch.float32),
                                         t(ep, torch.int32), len(ep), out, cnt, **extra)
    c = int(cnt.cpu().numpy()[0])
    want = {tuple(p) for p in ref[f"{key}/explicit_pairs"]}
    assert c == len(want) and {tuple(p) for p in out.cpu().numpy()[:c]} == want
