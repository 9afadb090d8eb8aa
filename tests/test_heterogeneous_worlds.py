"""Heterogeneous worlds (newton/_src/sim/model.py:881-900: worlds may differ in topology).  A model whose worlds differ is served
through its world groups (newton_amd/hetero.py): maximal runs of same-topology worlds, one launch per group.

CPU: the grouping itself (runs, slices, concatenation == the global arrays, shape-id translation).
GPU (-m gpu): a mixed model (quadrupeds | box stacks | quadrupeds | pendulums) through the unchanged Newton-shaped calls
(model.state(), CollisionPipeline, SolverXPBD / SolverSemiImplicit / SolverFeatherstone .step, SolverXPBD.rollout) against the
CPU checker stepping the SAME global flat model: contact rows identical in ids / order, poses within the single-step tolerances
of the homogeneous tests (1e-5 rel on body_q), and bitwise equal to a replicated model of the same worlds.
"""
import numpy as np
import pytest

import newton_amd as nt

DT = 1e-3


def _pendulum_builder():
    b = nt.ModelBuilder()
    l0 = b.add_link(xform=[0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0], mass=1.0, inertia=np.eye(3) * 0.05)
    b.add_shape_sphere(l0, radius=0.1)
    j0 = b.add_joint_revolute(-1, l0, axis=(0.0, 1.0, 0.0), parent_xform=[0.0, 0.0, 1.5, 0.0, 0.0, 0.0, 1.0],
                              child_xform=[0.0, 0.0, 0.5, 0.0, 0.0, 0.0, 1.0])
    l1 = b.add_link(xform=[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0], mass=1.0, inertia=np.eye(3) * 0.05)
    b.add_shape_sphere(l1, radius=0.1)
    j1 = b.add_joint_revolute(l0, l1, axis=(0.0, 1.0, 0.0), parent_xform=[0.0, 0.0, -0.5, 0.0, 0.0, 0.0, 1.0],
                              child_xform=[0.0, 0.0, 0.5, 0.0, 0.0, 0.0, 1.0])
    b.add_articulation([j0, j1])
    b.joint_q[-2:] = [0.3, -0.2]
    return b


def _box_builder(n):
    env = nt.ModelBuilder()
    for k in range(n):
        body = env.add_body(xform=[0.0, 0.0, 0.5 + 1.001 * k, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_box(body, hx=0.5, hy=0.5, hz=0.5)
    return env


def mixed_model(layout, device=None, seed=5):
    """layout: sequence of ("quadruped" | "boxes3" | "boxes2" | "pendulum", count) runs, plus one global ground plane."""
    from scenes import quadruped_builder

    kinds = {"quadruped": quadruped_builder(), "boxes3": _box_builder(3), "boxes2": _box_builder(2), "pendulum": _pendulum_builder()}
    scene = nt.ModelBuilder()
    for kind, count in layout:
        for _ in range(count):
            scene.add_world(kinds[kind])
    scene.add_ground_plane(cfg=kinds["quadruped"].default_shape_cfg)
    model = scene.finalize(device=device)
    # lower the quadrupeds onto the ground (feet in contact on the first step) and de-correlate the worlds
    rng = np.random.default_rng(seed)
    jw = np.asarray(model.joint_world)
    for w in range(model.world_count):
        j = np.flatnonzero(jw == w)
        if model.joint_type[j[0]] == nt.JointType.FREE and len(j) == 13:
            q0 = int(model.joint_q_start[j[0]])
            model.joint_q[q0 + 2] -= 0.26 - rng.uniform(0.0, 0.003)
    bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    model.body_q = bq
    model.body_qd = (bqd + rng.normal(0.0, 0.05, size=bqd.shape)).astype(np.float32)
    if np.any(np.asarray(model.body_flags) & 2):
        raise AssertionError("unexpected kinematic bodies")
    return model


LAYOUT = (("quadruped", 3), ("boxes3", 2), ("quadruped", 2), ("pendulum", 4), ("boxes2", 1))


def test_world_runs_and_slices_reproduce_the_global_arrays():
    from newton_amd.hetero import world_runs

    model = mixed_model(LAYOUT)
    assert model.is_heterogeneous and model.env is None
    assert world_runs(model) == [(0, 3), (3, 5), (5, 7), (7, 11), (11, 12)]
    g = model.world_groups
    assert g.ranges == [(0, 3), (3, 5), (5, 7), (7, 11), (11, 12)]
    g.sync_host()  # mixed_model edited body_q / body_qd after the groups were cut (eval_fk_numpy cuts them)
    for k in ("body_q", "body_qd", "body_mass", "joint_q", "joint_qd", "joint_target_q", "joint_type", "joint_axis"):
        assert np.array_equal(np.concatenate([getattr(p, k) for p in g.parts]), getattr(model, k)), k
    # every group is a valid replicated model of its own and knows the global id of each of its shapes
    sw = np.asarray(model.shape_world)
    for (b, e), p in zip(g.ranges, g.parts):
        assert p.env is not None and p.world_count == e - b and not p.is_heterogeneous
        ids = p._global_shape_ids
        assert np.array_equal(np.asarray(model.shape_type)[ids], p.shape_type)
        assert set(sw[ids].tolist()) <= set(range(b, e)) | {-1}
        glob_pairs = ids[np.asarray(p.shape_contact_pairs)]
        want = np.asarray(model.shape_contact_pairs)
        want = want[np.isin(sw[want[:, 0]], range(b, e)) | np.isin(sw[want[:, 1]], range(b, e))]
        assert np.array_equal(glob_pairs, want)
    # a replicated model stays on the direct path
    from scenes import quadruped_scene

    assert not quadruped_scene(3).is_heterogeneous
    with pytest.raises(ValueError):
        _ = quadruped_scene(3).world_groups


def test_host_composites_concatenate_and_split():
    model = mixed_model(LAYOUT)
    s, c = model.state(), model.control()
    assert type(s).__name__ == "GroupedState" and len(s.parts) == 5
    assert np.array_equal(s.body_q, model.body_q) and np.array_equal(s.joint_q, model.joint_q)
    new_q = (model.body_q + 1.0).astype(np.float32)
    s.body_q = new_q
    assert np.array_equal(s.body_q, new_q)
    assert np.array_equal(s.parts[1].body_q, new_q[39:45])
    jf = np.arange(model.joint_dof_count, dtype=np.float32)
    c.joint_f = jf
    assert np.array_equal(c.joint_f, jf) and np.array_equal(c.parts[3].joint_f, jf[18 * 5 + 6 * 6:18 * 5 + 6 * 6 + 8])
    with pytest.raises(ValueError):
        s.body_q = new_q[:-1]
    # no CPU fallback either way
    with pytest.raises(nt._lib.NewtonHipError):
        nt.solvers.SolverXPBD(model)


def _oracle_single_step(model, solver_name, jf):
    from oracle_bridge import Oracle, OracleState

    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    o.collide(os0.body_q, oc)
    ctrl = o.control(joint_f=jf)
    if solver_name == "xpbd":
        o.xpbd_step(os0, os1, ctrl, oc, DT)
    elif solver_name == "semi_implicit":
        o.semi_implicit_step(os0, os1, ctrl, oc, DT)
    else:
        o.featherstone_step(os0, os1, ctrl, oc, DT)
    return os1, oc


@pytest.mark.gpu
@pytest.mark.parametrize("solver_name", ["xpbd", "semi_implicit", "featherstone"])
def test_mixed_worlds_single_step_against_the_checker(oracle_lib, solver_name):
    from tolerances import check

    model = mixed_model(LAYOUT, device="cuda:0")
    rng = np.random.default_rng(3)
    jf = rng.normal(0.0, 1.0, size=model.joint_dof_count).astype(np.float32)
    cls = {"xpbd": nt.solvers.SolverXPBD, "semi_implicit": nt.solvers.SolverSemiImplicit,
           "featherstone": nt.solvers.SolverFeatherstone}[solver_name]
    solver = cls(model)
    assert type(solver).__name__ == "GroupedSolver" and len(solver.parts) == 5
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s0, s1, ctrl = model.state(), model.state(), model.control()
    ctrl.joint_f = jf
    s0.clear_forces()
    pipe.collide(s0, contacts)
    solver.step(s0, s1, ctrl, contacts, DT)

    os1, oc = _oracle_single_step(model, solver_name, jf)
    n = int(oc.count[0])
    assert n > 0 and int(contacts.rigid_contact_count.cpu().numpy()[0]) == n
    assert np.array_equal(contacts.rigid_contact_shape0.cpu().numpy(), oc.shape0[:n])  # global shape ids, append order
    assert np.array_equal(contacts.rigid_contact_shape1.cpu().numpy(), oc.shape1[:n])
    for name in ("point0", "point1", "normal"):
        assert np.max(np.abs(getattr(contacts, "rigid_contact_" + name).cpu().numpy() - getattr(oc, name)[:n])) <= 1e-5, name
    q, qd = s1.body_q.cpu().numpy(), s1.body_qd.cpu().numpy()
    assert q.shape == (model.body_count, 7)
    check(f"hetero_single_step_{solver_name}", q, qd, os1.body_q, os1.body_qd, pos=1e-5, rot=1e-5, lin_vel_abs=2e-4, ang_vel_abs=2e-3)


@pytest.mark.gpu
def test_mixed_worlds_rollout_equals_the_replicated_models_bitwise(oracle_lib):
    """Each group must step exactly like a replicated model of its worlds: rollout of the mixed model == per-kind replicated
    models rolled out on their own, bit for bit; and rollout == the step loop."""
    from newton_amd.worlds import slice_worlds

    model = mixed_model(LAYOUT, device="cuda:0")
    solver = nt.solvers.SolverXPBD(model)
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    s0, s1, ctrl = model.state(), model.state(), model.control()
    out = solver.rollout(s0, s1, ctrl, contacts, DT, 7)
    q = out.body_q.cpu().numpy()
    # the step loop on fresh states
    t0, t1 = model.state(), model.state()
    c2 = pipe.contacts()
    for _ in range(7):
        t0.clear_forces()
        pipe.collide(t0, c2)
        solver.step(t0, t1, ctrl, c2, DT)
        t0, t1 = t1, t0
    assert np.array_equal(t0.body_q.cpu().numpy(), q)
    # independent sub-models
    parts = []
    for b, e in model.world_groups.ranges:
        sub = slice_worlds(model, b, e, device="cuda:0")
        ss = nt.solvers.SolverXPBD(sub)
        a0, a1 = sub.state(), sub.state()
        r = ss.rollout(a0, a1, sub.control(), nt.CollisionPipeline(sub).contacts(), DT, 7)
        parts.append(r.body_q.cpu().numpy())
    assert np.array_equal(np.concatenate(parts), q)
    assert np.all(np.isfinite(q))
