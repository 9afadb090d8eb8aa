"""The gfx950 kernel SOURCES, executed on the CPU, against the oracle -- without a GPU.

tests/emu compiles newton_amd/csrc/*.hpp + nt_kernels.hip for the host against a small HIP shim (one OS thread per GPU thread,
std::barrier for __syncthreads, workgroups one after the other) and these tests drive the resulting library through the same C
ABI and the same env-major SoA layout as the product.  Both sides then run on the same x86 arithmetic with FMA contraction
off, so wherever the kernels restate the oracle operation for operation the results are bit-identical (collision geometry,
Featherstone); where a kernel splits a reduction differently from the serial oracle (XPBD joint rows on two lanes) the
tolerances of the GPU parity tests apply: 1e-5 on positions, 2e-4 on the dt-amplified XPBD velocities.  What this does NOT cover: the real wave scheduling, LDS bank behaviour, register
allocation and the device's libm (ocml) -- that is what the `-m gpu` tests are for.  The emulated library is test
infrastructure; nothing under newton_amd/ can load it."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

import newton_amd as nt  # noqa: E402

TOL = 1e-6


@pytest.fixture(scope="module")
def H(oracle_lib):
    import harness

    harness.lib()  # builds tests/emu/_build/libnewton_emu.so on first use (g++, ~15 s)
    return harness


def _lower(model, dz):
    E = model.world_count
    model.joint_q.reshape(E, -1)[:, 2] -= dz
    bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    model.body_q, model.body_qd = bq, bqd


def _close(a, b, tol=TOL):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b) / np.maximum(np.abs(b), 1.0))) <= tol


def _same_contacts(ct, oc, tol=TOL):
    e = ct.export()
    n = int(oc.count[0])
    assert int(e["count"][0]) == n
    assert np.array_equal(e["shape0"][:n], oc.shape0[:n]) and np.array_equal(e["shape1"][:n], oc.shape1[:n])
    for k in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
        assert n == 0 or np.max(np.abs(e[k][:n] - getattr(oc, k)[:n])) <= tol, k
    return n


@pytest.mark.parametrize("n_env,epb", [(5, 16), (9, 8), (3, 1), (70, 0)])
def test_xpbd_quadruped_step(H, n_env, epb):
    from oracle_bridge import Oracle, OracleState
    from scenes import quadruped_scene

    model = quadruped_scene(n_env)
    _lower(model, 0.26)
    rng = np.random.default_rng(7)
    model.body_qd = (model.body_qd + rng.normal(0, 0.2, size=model.body_qd.shape)).astype(np.float32)
    jf = rng.normal(0, 2.0, size=model.joint_dof_count).astype(np.float32)
    em = H.EmuModel(model)
    s0, s1, ct = H.EmuState(em), H.EmuState(em), H.EmuContacts(em)
    H.collide(em, s0, ct, epb=epb)
    H.xpbd_step(em, s0, s1, H.EmuControl(em, joint_f=jf), ct, 1e-3, epb=epb)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    o.collide(os0.body_q, oc)
    o.xpbd_step(os0, os1, o.control(joint_f=jf), oc, 1e-3)
    assert _same_contacts(ct, oc) > 0
    assert _close(s1.aos("body_q"), os1.body_q, 1e-5) and _close(s1.aos("body_qd"), os1.body_qd, 2e-4)


def test_xpbd_rollout_equals_loop_and_oracle(H):
    from oracle_bridge import Oracle, OracleState
    from scenes import quadruped_scene

    model = quadruped_scene(20)
    _lower(model, 0.24)
    em = H.EmuModel(model)
    ctrl, ct = H.EmuControl(em), H.EmuContacts(em)
    out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, ct, 1e-3, 7)
    a, b = H.EmuState(em), H.EmuState(em)
    for _ in range(7):
        a.body_f[:] = 0
        H.collide(em, a, ct)
        H.xpbd_step(em, a, b, ctrl, ct, 1e-3)
        a, b = b, a
    assert np.array_equal(out.body_q, a.body_q) and np.array_equal(out.body_qd, a.body_qd)  # fused == launch by launch
    o = Oracle(model)
    oout = o.xpbd_rollout(OracleState(model), OracleState(model), o.control(), o.contacts(), 1e-3, 7)
    assert _close(out.aos("body_q"), oout.body_q, 1e-5) and _close(out.aos("body_qd"), oout.body_qd, 1e-3)


@pytest.mark.parametrize("name", sorted(__import__("pair_scenes").CONVEX_CASES))
def test_convex_pair_contacts(H, name):
    from oracle_bridge import Oracle
    from pair_scenes import CONVEX_CASES, pair_model

    model = pair_model(CONVEX_CASES[name])
    em = H.EmuModel(model)
    ct = H.EmuContacts(em)
    H.collide(em, H.EmuState(em), ct)
    o = Oracle(model)
    oc = o.contacts()
    o.collide(model.body_q, oc)
    _same_contacts(ct, oc)


def test_box_stack_and_mixed_primitives(H):
    from oracle_bridge import Oracle, OracleState
    from scenes import box_stack_scene, mixed_primitive_scene

    for model, kw in ((box_stack_scene(5), dict(iterations=4)), (mixed_primitive_scene(4), dict(iterations=2))):
        em = H.EmuModel(model)
        s0, s1, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
        o = Oracle(model)
        os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
        for _ in range(3):
            s0.body_f[:] = 0
            H.collide(em, s0, ct)
            H.xpbd_step(em, s0, s1, ctrl, ct, 1.0 / 240.0, **kw)
            os0.body_f[:] = 0
            o.collide(os0.body_q, oc)
            o.xpbd_step(os0, os1, o.control(), oc, 1.0 / 240.0, **kw)
            assert _same_contacts(ct, oc, 2e-5) > 0
            s0, s1, os0, os1 = s1, s0, os1, os0
        assert _close(s0.aos("body_q"), os0.body_q, 1e-5) and _close(s0.aos("body_qd"), os0.body_qd, 2e-4)


def test_large_scene_runs_one_environment_per_workgroup(H):
    """The reference's ramp line-up (161 pairs, 59 KB of LDS): envs_per_block = 1 is picked automatically."""
    from oracle_bridge import Oracle, OracleState
    from test_ramp_scene import _build

    model, _ = _build()
    em = H.EmuModel(model)
    s0, s1, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    for _ in range(2):
        H.collide(em, s0, ct)
        H.xpbd_step(em, s0, s1, ctrl, ct, 1.0 / 600.0)
        # teacher-forced: the checker collides and steps from the kernels' own state (identical inputs => identical contacts; an
        # open-loop second step starts from states one rounding apart, and the single MPR contacts of the cone / hull cubes on the ramp
        # turn 2e-7 of pose into 3e-2 of normal)
        os0.body_q[:], os0.body_qd[:] = s0.aos("body_q"), s0.aos("body_qd")
        o.collide(os0.body_q, oc)
        o.xpbd_step(os0, os1, o.control(), oc, 1.0 / 600.0)
        assert _same_contacts(ct, oc, TOL) > 20
        assert _close(s1.aos("body_q"), os1.body_q, 1e-5)
        s0, s1, os0, os1 = s1, s0, os1, os0


@pytest.mark.parametrize("tiles,per_env,n_env", [(600, 1, 3), (600, 2, 2), (150, 1, 5)])
def test_more_global_shapes_than_workgroup_lanes(H, tiles, per_env, n_env):
    """stage_global_world stages the static (world -1) shapes' transforms / AABBs once per launch: with more of them than the
    workgroup has lanes (64 ... 512) the staging has to loop, or the pairs against the shapes beyond the workgroup size read
    uninitialised LDS.  Spheres rest on tiles spread over the whole row, the first and the last included."""
    from oracle_bridge import Oracle, OracleState
    from scenes import tiled_floor_scene

    model = tiled_floor_scene(n_env, tiles=tiles, per_env=per_env)
    em = H.EmuModel(model)
    s0, s1, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    H.collide(em, s0, ct)
    o.collide(os0.body_q, oc)
    assert _same_contacts(ct, oc) >= n_env * per_env  # every sphere touches its own tile (+ the neighbours the default gap admits)
    sh1 = ct.export()["shape1"][: n_env * per_env]
    assert sh1.max() >= model.shape_count - tiles + min(tiles - 1, 512) or tiles < 512  # a tile beyond the widest workgroup is hit
    # the fused rollout stages them once per launch too
    out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, H.EmuContacts(em), 1e-3, 4)
    oout = o.xpbd_rollout(OracleState(model), OracleState(model), o.control(), o.contacts(), 1e-3, 4)
    assert _close(out.aos("body_q"), oout.body_q, 1e-5) and _close(out.aos("body_qd"), oout.body_qd, 1e-3)
    # ... and the spheres stay on their tiles (a missed tile lets them fall 4 mm in 4 ms -- far beyond the tolerance above)
    assert float(np.min(out.aos("body_q")[:, 2])) > 0.14


def test_restitution_and_reporting(H):
    from oracle_bridge import Oracle, OracleState
    from scenes import quadruped_scene

    from newton_amd import _lib as L

    model = quadruped_scene(6)
    model.request_state_attributes("body_parent_f")
    _lower(model, 0.26)
    rng = np.random.default_rng(3)
    model.body_qd = (model.body_qd + rng.normal(0, 0.3, size=model.body_qd.shape)).astype(np.float32)
    jf = rng.normal(0, 5.0, size=model.joint_dof_count).astype(np.float32)
    em = H.EmuModel(model)
    t = em.t
    s0, s1, ct = H.EmuState(em), H.EmuState(em, parent_f=True), H.EmuContacts(em)
    impulse = np.zeros((6, t.np * t.cpp, t.env_stride), dtype=np.float32)
    joint_impulse = np.zeros((6, t.nj, t.env_stride), dtype=np.float32)
    rep = L.nt_xpbd_report(impulse.ctypes.data, joint_impulse.ctypes.data)
    H.collide(em, s0, ct)
    H.xpbd_step(em, s0, s1, H.EmuControl(em, joint_f=jf), ct, 1e-3, report=rep, enable_restitution=True)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    o.collide(os0.body_q, oc)
    force = np.zeros((oc.max, 6), dtype=np.float32)
    o.xpbd_step(os0, os1, o.control(joint_f=jf), oc, 1e-3, enable_restitution=True, contact_force_out=force)
    assert _close(s1.aos("body_q"), os1.body_q, 1e-5) and _close(s1.aos("body_qd"), os1.body_qd, 2e-4)
    assert np.abs(os1.body_parent_f).max() > 1.0
    assert np.max(np.abs(s1.aos("body_parent_f") - os1.body_parent_f)) <= 1e-4 * np.abs(os1.body_parent_f).max()
    # contacts.force through nt_contacts_export_force
    n = int(oc.count[0])
    out = np.zeros((ct.rigid_contact_max, 6), dtype=np.float32)
    d = ct.desc()
    import ctypes as C

    H.check(H.lib().nt_contacts_export_force(C.byref(em.desc), C.byref(d), impulse.ctypes.data, 1e-3, ct.rigid_contact_max,
                                             out.ctypes.data, ct.scan.ctypes.data, None), "nt_contacts_export_force")
    assert np.abs(force[:n]).max() > 1.0
    assert np.max(np.abs(out[:n] - force[:n])) <= 1e-4 * np.abs(force[:n]).max() and np.all(out[n:] == 0.0)


def test_semi_implicit_pendulum_and_joint_zoo(H):
    from oracle_bridge import Oracle, OracleState
    from scenes import joint_zoo_scene, pendulum_scene

    for model, dt, steps in ((pendulum_scene(3), 1e-3, 30), (joint_zoo_scene(2), 1e-4, 30)):
        em = H.EmuModel(model)
        s0, s1, ctrl = H.EmuState(em), H.EmuState(em), H.EmuControl(em)
        o = Oracle(model)
        os0, os1 = OracleState(model), OracleState(model)
        for _ in range(steps):
            s0.body_f[:] = 0
            H.semi_implicit_step(em, s0, s1, ctrl, None, dt)
            os0.body_f[:] = 0
            o.semi_implicit_step(os0, os1, o.control(), None, dt)
            s0, s1, os0, os1 = s1, s0, os1, os0
        assert _close(s0.aos("body_q"), os0.body_q, 1e-5) and _close(s0.aos("body_qd"), os0.body_qd, 1e-4)


@pytest.mark.parametrize("n_env,epb", [(6, 0), (5, 8), (16, 4)])
def test_featherstone_step_and_rollout(H, n_env, epb):
    """Includes the wave-cooperative Cholesky: its wavefront fence becomes a rendezvous of the participating lanes."""
    from oracle_bridge import Oracle, OracleState
    from scenes import quadruped_scene

    model = quadruped_scene(n_env)
    model.request_state_attributes("body_parent_f")
    _lower(model, 0.26)
    rng = np.random.default_rng(11)
    model.joint_qd = (model.joint_qd + rng.normal(0, 0.3, size=model.joint_qd.shape)).astype(np.float32)
    jf = rng.normal(0, 2.0, size=model.joint_dof_count).astype(np.float32)
    em = H.EmuModel(model)
    ctrl, ct = H.EmuControl(em, joint_f=jf), H.EmuContacts(em)
    s0, s1 = H.EmuState(em), H.EmuState(em, parent_f=True)
    H.collide(em, s0, ct, epb=0)
    H.featherstone_step(em, s0, s1, ctrl, ct, 1e-3, epb=epb)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    o.collide(os0.body_q, oc)
    o.featherstone_step(os0, os1, o.control(joint_f=jf), oc, 1e-3)
    assert oc.count[0] > 0
    for name, tol in (("joint_q", TOL), ("body_q", TOL), ("joint_qd", 1e-5), ("body_qd", 1e-5)):
        assert _close(s1.aos(name), getattr(os1, name), tol), name
    assert np.max(np.abs(s1.aos("body_parent_f") - os1.body_parent_f)) <= 1e-5 * np.abs(os1.body_parent_f).max()
    # fused rollout == launch-by-launch loop (bitwise) and tracks the oracle
    a, b = H.EmuState(em), H.EmuState(em)
    out = H.featherstone_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, ct, 1e-3, 3, epb=epb)
    for _ in range(3):
        a.body_f[:] = 0
        H.collide(em, a, ct)
        H.featherstone_step(em, a, b, ctrl, ct, 1e-3, epb=epb)
        a, b = b, a
    assert np.array_equal(out.joint_q, a.joint_q) and np.array_equal(out.body_q, a.body_q)
    # (substeps >= 1 of the rollout start from the previous substep's FK instead of repeating eval_rigid_fk: same bits)
    assert np.array_equal(out.joint_qd, a.joint_qd) and np.array_equal(out.body_qd, a.body_qd)


def test_featherstone_uniform_tile_of_16_is_bitwise_the_tiles_of_4(H):
    """Round 6: SolverFeatherstone's rollout keeps 16 environments per workgroup when the parameters are uniform (one block-shared
    parameter copy, the tree mode's own solve region with a packed H, 32 lanes per environment: nt_featherstone.hip).  Same arithmetic per
    quantity, other layout and lane count: the result must be BITWISE the one of the per-environment tiles of 4 -- on a second, partly
    filled workgroup too (17 environments), with live contacts."""
    from scenes import quadruped_scene

    model = quadruped_scene(17)
    _lower(model, 0.26)
    rng = np.random.default_rng(3)
    model.joint_qd = (model.joint_qd + rng.normal(0, 0.3, size=model.joint_qd.shape)).astype(np.float32)
    em = H.EmuModel(model)
    assert em.desc.params_uniform == 1
    jf = np.tile(rng.normal(0, 2.0, size=model.joint_dof_count // 17).astype(np.float32), 17)  # (uniform controls are not required; same per env here)
    ctrl = H.EmuControl(em, joint_f=jf)
    outs = []
    for epb in (4, 16):
        ct = H.EmuContacts(em)
        out = H.featherstone_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, ct, 1e-3, 3, epb=epb)
        outs.append((out.joint_q.copy(), out.joint_qd.copy(), out.body_q.copy(), out.body_qd.copy(), ct.export()))
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert np.array_equal(a, b)
    assert int(outs[0][4]["count"][0]) == int(outs[1][4]["count"][0]) > 0
    for k in ("shape0", "shape1", "point0", "normal"):
        assert np.array_equal(outs[0][4][k], outs[1][4][k]), k


@pytest.mark.parametrize("lowered,substeps,tol", [(False, 60, 2e-6), (True, 10, 2e-4)])
def test_featherstone_tree_and_dense_orders_stay_together_over_a_rollout(H, lowered, substeps, tol):
    """SolverFeatherstone defaults to the tree-structured mass matrix (composite inertias + leaf-first L^T D L), which
    is within the single-step contract of the reference's dense order but not bit-identical to it.  Bound the drift where it could
    accumulate -- open-loop rollouts of both orders from the same state: 60 substeps of quadrupeds swinging their legs in the air
    (smooth dynamics: the two factorisations must stay within a few ulps of accumulated rounding), and 10 substeps with the feet in
    the ground (the explicit penalty contacts amplify ANY rounding difference 3-10x per substep -- DESIGN.md section 4 -- so the
    bound there is the one the dense order itself keeps against the checker)."""
    from oracle_bridge import Oracle, OracleState
    from scenes import quadruped_scene

    model = quadruped_scene(4)
    if lowered:
        _lower(model, 0.25)
    rng = np.random.default_rng(5)
    model.joint_qd = (model.joint_qd + rng.normal(0, 0.2, size=model.joint_qd.shape)).astype(np.float32)
    em = H.EmuModel(model)
    res = {}
    for dense in (False, True):
        ctrl, ct = H.EmuControl(em), H.EmuContacts(em)
        out = H.featherstone_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, ct, 1e-3, substeps, dense=dense)
        res[dense] = (out.aos("joint_q").copy(), int(ct.env_count.sum()))
    assert (res[False][1] > 0) == lowered and (res[True][1] > 0) == lowered
    d_orders = float(np.max(np.abs(res[False][0] - res[True][0])))
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    for _ in range(substeps):
        o.collide(os0.body_q, oc)
        o.featherstone_step(os0, os1, o.control(), oc, 1e-3)
        os0, os1 = os1, os0
    d_tree = float(np.max(np.abs(res[False][0] - os0.joint_q)))
    d_dense = float(np.max(np.abs(res[True][0] - os0.joint_q)))
    print(f"[featherstone orders] lowered={lowered} substeps={substeps}: tree-dense {d_orders:.2e}, tree-checker {d_tree:.2e}, dense-checker {d_dense:.2e}")
    assert d_orders <= tol and d_tree <= tol and d_dense <= tol, (d_orders, d_tree, d_dense)


def test_featherstone_kinematic_root_and_zoo(H):
    from oracle_bridge import Oracle, OracleState
    from scenes import joint_zoo_scene
    from test_kinematic_links import _pendulum_on_kinematic_root

    for model in (_pendulum_on_kinematic_root(None, "revolute")[0], joint_zoo_scene(3, free_root=True)):
        em = H.EmuModel(model)
        s0, s1, ctrl = H.EmuState(em), H.EmuState(em), H.EmuControl(em)
        o = Oracle(model)
        os0, os1 = OracleState(model), OracleState(model)
        for k in range(25):
            if model.joint_count == 2:  # prescribed motion of the kinematic root joint
                q, qd = 0.4 * np.sin(0.025 * k), 0.4 * 6.0 * np.cos(0.025 * k)
                s0.joint_q[0, 0, :em.t.env_count], s0.joint_qd[0, 0, :em.t.env_count] = q, qd
                os0.joint_q[0], os0.joint_qd[0] = q, qd
            s0.body_f[:] = 0
            H.featherstone_step(em, s0, s1, ctrl, None, 1e-3)
            os0.body_f[:] = 0
            o.featherstone_step(os0, os1, o.control(), None, 1e-3)
            s0, s1, os0, os1 = s1, s0, os1, os0
        assert _close(s0.aos("joint_q"), os0.joint_q, 1e-5) and _close(s0.aos("body_q"), os0.body_q, 1e-5)


def test_emulated_library_is_test_only():
    """The product loader resolves only newton_amd/libnewton_hip.so: no environment override, no reference to the emulated
    library anywhere in the package."""
    from newton_amd import _lib

    assert "emu" not in os.path.basename(_lib.LIB_PATH)
    assert "environ" not in open(_lib.__file__).read()
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "newton_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "libnewton_emu" not in open(os.path.join(dirpath, f)).read(), f


@pytest.mark.parametrize("n_hulls,n_env", [(40, 3), (64, 1)])
def test_pair_heavy_scene_keeps_contact_records_in_hbm(H, n_hulls, n_env):
    """Config C5 without SDF / hydroelastic: hulls in a five-wall bin, n(n-1)/2 + 5n candidate pairs per environment (2 336
    for 64 hulls).  The per-contact solver records no longer fit LDS, so nt_model.contact_scratch_in_hbm is chosen and the
    one-environment-per-workgroup kernels keep them in nt_contacts.cw."""
    from oracle_bridge import Oracle, OracleState
    from scenes import hull_bin_scene

    model = hull_bin_scene(n_env, n_hulls)
    em = H.EmuModel(model)
    assert em.desc.contact_scratch_in_hbm == 1 and model.env.np == n_hulls * (n_hulls - 1) // 2 + 5 * n_hulls
    s0, s1, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    for step in range(2):
        s0.body_f[:] = 0
        H.collide(em, s0, ct)
        H.xpbd_step(em, s0, s1, ctrl, ct, 1.0 / 600.0, enable_restitution=(step == 1))
        os0.body_f[:] = 0
        o.collide(os0.body_q, oc)
        o.xpbd_step(os0, os1, o.control(), oc, 1.0 / 600.0, enable_restitution=(step == 1))
        assert _same_contacts(ct, oc, TOL if step == 0 else 2e-5) > 5 * n_hulls
        s0, s1, os0, os1 = s1, s0, os1, os0
    assert _close(s0.aos("body_q"), os0.body_q, 1e-5) and _close(s0.aos("body_qd"), os0.body_qd, 2e-4)
    if n_env > 1:  # fused rollout == launch-by-launch loop, bitwise
        out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, ct, 1.0 / 600.0, 3)
        # (the rollout keeps its substeps' records in nt_contacts.cr and writes the Contacts buffers at the last substep only)
        ids0, ids1, data, hit = ct.shape0.copy(), ct.shape1.copy(), ct.data.copy(), ct.pair_hit.copy()
        a, b = H.EmuState(em), H.EmuState(em)
        for k in range(3):
            a.body_f[:] = 0
            H.collide(em, a, ct)
            if k == 2:  # the Contacts the rollout left are those of its last collide
                E = model.env.env_count
                live = ct.shape0[:, :E] != ct.shape1[:, :E]
                assert np.array_equal(ids0[:, :E], ct.shape0[:, :E]) and np.array_equal(ids1[:, :E], ct.shape1[:, :E])
                assert np.array_equal(hit[:, :E], ct.pair_hit[:, :E]) and np.array_equal(data[:, :, :E][:, live], ct.data[:, :, :E][:, live])
            H.xpbd_step(em, a, b, ctrl, ct, 1.0 / 600.0)
            a, b = b, a
        assert np.array_equal(out.body_q, a.body_q) and np.array_equal(out.body_qd, a.body_qd)
    # the other solvers refuse this mode instead of overflowing LDS
    with pytest.raises(RuntimeError):
        H.semi_implicit_step(em, s0, s1, ctrl, ct, 1e-4)
    # the tile takes its wide workgroup (384 lanes, 20 rows of polygon scratch each) while that still fits the CU -- it does for
    # config C5's 64 hulls (158 KB of 160) -- and says so
    import ctypes as C

    from newton_amd import _lib

    shape = (C.c_int32 * 5)()
    p = _lib.nt_xpbd_params(2, 0.4, 0.4, 0.0, 0.0, 0.8, 1, 0.0, 0, 0)
    assert H.lib().nt_xpbd_rollout_shape(C.byref(em.desc), C.byref(p), None, shape) == 0
    assert list(shape)[:2] == [1, 384] and shape[4] == 3  # one environment per workgroup, 384 lanes, convex + pair-heavy


def test_boundary_helpers(H):
    """nt_state_reset (masked env columns), nt_contacts_export with a capacity below the count (the counter keeps counting,
    writes are clamped), nt_pack_aos / nt_unpack_aos round trip."""
    import ctypes as C

    from scenes import mixed_primitive_scene

    rng = np.random.default_rng(0)
    for E in (1, 7, 33):
        model = mixed_primitive_scene(E)
        em = H.EmuModel(model)
        t = em.t
        src, dst = H.EmuState(em), H.EmuState(em)
        names = ("body_q", "body_qd", "joint_q", "joint_qd", "body_f")
        for n in names:
            getattr(dst, n)[...] = rng.normal(size=getattr(dst, n).shape).astype(np.float32)
        before = {n: getattr(dst, n).copy() for n in names}
        mask = (rng.random(E) < 0.5).astype(np.uint8)
        dd, ds = dst.desc(), src.desc()
        H.check(H.lib().nt_state_reset(C.byref(em.desc), C.byref(dd), C.byref(ds), mask.ctypes.data, None), "nt_state_reset")
        for n in names:
            want = before[n].copy()
            want[..., :E][..., mask.astype(bool)] = getattr(src, n)[..., :E][..., mask.astype(bool)]
            assert np.array_equal(getattr(dst, n)[..., :E], want[..., :E]), n
        ct = H.EmuContacts(em)
        H.collide(em, H.EmuState(em), ct)
        full = ct.export()
        n = int(full["count"][0])
        assert n > 3
        cap = n // 2
        out = {"count": np.zeros(1, np.int32), "shape0": np.full(cap, -7, np.int32), "shape1": np.full(cap, -7, np.int32)}
        for k in ("point0", "point1", "offset0", "offset1", "normal"):
            out[k] = np.zeros((cap, 3), np.float32)
        out["margin0"], out["margin1"] = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        d = ct.desc()
        H.check(H.lib().nt_contacts_export(C.byref(em.desc), C.byref(d), cap, *(out[k].ctypes.data for k in (
            "count", "shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")),
            ct.scan.ctypes.data, None), "nt_contacts_export")
        assert int(out["count"][0]) == n  # keeps counting past the capacity (collide.py:176-177)
        assert np.array_equal(out["shape0"], full["shape0"][:cap]) and np.array_equal(out["point0"], full["point0"][:cap])
        aos = rng.normal(size=(E * t.nb, 7)).astype(np.float32)
        soa, back = np.zeros((7, t.nb, t.env_stride), np.float32), np.zeros((E * t.nb, 7), np.float32)
        H.check(H.lib().nt_pack_aos(aos.ctypes.data, soa.ctypes.data, 7, t.nb, E, t.env_stride, None), "nt_pack_aos")
        H.check(H.lib().nt_unpack_aos(soa.ctypes.data, back.ctypes.data, 7, t.nb, E, t.env_stride, None), "nt_unpack_aos")
        assert np.array_equal(back, aos) and np.array_equal(soa, em.to_soa(aos, 7, t.nb))


def test_finite_plane(H):
    """A finite plane is a rectangle: tight AABB through its support map (collide.py:448-465, support_function.py:334-345), so
    shapes beyond its edges are no candidates; the analytic plane-primitive contacts ignore the extents like the reference's
    collide_plane_* functions; cones and hulls meet it through MPR/GJK without the infinite-plane box proxy."""
    from oracle_bridge import Oracle, OracleState

    from newton_amd import _np_math as nm

    rng = np.random.default_rng(1)
    env = nt.ModelBuilder()
    kinds = ["sphere", "box", "capsule", "cylinder", "ellipsoid", "cone", "hull", "box", "sphere", "cone", "hull", "cylinder"]
    for k, kind in enumerate(kinds):
        b = env.add_body(xform=[-1.6 + 0.3 * k, rng.uniform(-0.9, 0.9), 0.12, *nm.quat_rpy(*rng.uniform(-0.5, 0.5, size=3))])
        if kind == "sphere":
            env.add_shape_sphere(b, radius=0.1)
        elif kind == "box":
            env.add_shape_box(b, hx=0.1, hy=0.08, hz=0.06)
        elif kind == "capsule":
            env.add_shape_capsule(b, radius=0.06, half_height=0.1)
        elif kind == "cylinder":
            env.add_shape_cylinder(b, radius=0.08, half_height=0.1)
        elif kind == "ellipsoid":
            env.add_shape_ellipsoid(b, rx=0.12, ry=0.08, rz=0.06)
        elif kind == "cone":
            env.add_shape_cone(b, radius=0.08, half_height=0.1)
        else:
            env.add_shape_convex_hull(b, mesh=nt.Mesh.convex_hull_of(rng.normal(size=(14, 3)) * 0.06))
    scene = nt.ModelBuilder()
    scene.replicate(env, 3)
    plane = scene.add_shape_plane(xform=[0.0, 0.0, 0.0, *nm.quat_rpy(0.05, -0.03, 0.4)], width=2.0, length=1.0)
    model = scene.finalize()
    em = H.EmuModel(model)
    s0, s1, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    H.collide(em, s0, ct)
    pairs, lo, hi = o.collide(os0.body_q, oc)
    assert _same_contacts(ct, oc) > 30
    assert np.all(hi[plane] - lo[plane] < [2.6, 2.1, 0.5])  # the rectangle, not a half space
    with_plane = {int(p[0]) for p in pairs if p[1] == plane}
    assert 0 not in with_plane and 11 not in with_plane and {3, 5, 6} <= with_plane  # bodies past the short edges are culled
    H.xpbd_step(em, s0, s1, ctrl, ct, 1e-3)
    o.xpbd_step(os0, os1, o.control(), oc, 1e-3)
    assert _close(s1.aos("body_q"), os1.body_q, 1e-5)


@pytest.mark.parametrize("dense", [True, False])
def test_featherstone_large_articulation_one_environment_per_workgroup(H, dense):
    """A 40-link serial chain (40 dofs): P and H alone are (6 nj + nd) x 40 floats, 73 KB of LDS per environment, so the solver
    runs one environment per workgroup with all 64 lanes of the Cholesky wave on it; bit-identical to the oracle in the reference's
    dense operation order.  The tree-structured mass matrix (a 40-level dof tree here: the chain is its worst case) solves the same
    ill-conditioned system in another order: 1e-5 on positions, 2e-4 relative on velocities after 3 steps."""
    import ctypes as C

    from oracle_bridge import Oracle, OracleState

    b = nt.ModelBuilder()
    prev, joints = -1, []
    rng = np.random.default_rng(0)
    cfg = nt.ModelBuilder.ShapeConfig(has_shape_collision=False)
    for k in range(40):
        link = b.add_link()
        b.add_shape_capsule(link, radius=0.02, half_height=0.05, cfg=cfg)
        joints.append(b.add_joint_revolute(prev, link, axis=[(1, 0, 0), (0, 1, 0), (0, 0, 1)][k % 3],
                                           parent_xform=[0.0, 0.0, 0.12 if k else 2.0, 0.0, 0.0, 0.0, 1.0],
                                           target_ke=5.0, target_kd=0.5, armature=0.01))
        prev = link
    b.add_articulation(joints)
    scene = nt.ModelBuilder()
    scene.replicate(b, 3)
    model = scene.finalize()
    model.joint_q = rng.uniform(-0.3, 0.3, size=model.joint_coord_count).astype(np.float32)
    model.joint_qd = rng.normal(0.0, 0.3, size=model.joint_dof_count).astype(np.float32)
    em = H.EmuModel(model)
    assert 4 * H.lib().nt_featherstone_lds_bytes_per_env(C.byref(em.desc)) > 160 * 1024  # four environments do not fit
    s0, s1, ctrl = H.EmuState(em), H.EmuState(em), H.EmuControl(em)
    o = Oracle(model)
    os0, os1 = OracleState(model), OracleState(model)
    for _ in range(3):
        H.featherstone_step(em, s0, s1, ctrl, None, 1e-3, dense=dense)
        o.featherstone_step(os0, os1, o.control(), None, 1e-3)
        s0, s1, os0, os1 = s1, s0, os1, os0
    for name in ("joint_q", "joint_qd", "body_q", "body_qd"):
        tol = TOL if dense else (1e-5 if name.endswith("_q") else 2e-4)
        assert _close(s0.aos(name), getattr(os0, name), tol), name


def test_velocity_from_position_delta(H):
    """SolverXPBD.compute_body_velocity_from_position_delta (solver_xpbd.py:767-783, update_body_velocities
    xpbd/kernels.py:2547-2579): runs after the iterations and before restitution, on every tile kind; velocities are pose
    differences over dt, so pose differences of 1e-6 between the two libms come back as 1e-3 at dt = 1e-3 -- the bitwise checks
    (rollout == loop, poses unchanged by the flag) carry the weight."""
    from oracle_bridge import Oracle, OracleState
    from scenes import hull_bin_scene, pendulum_scene, quadruped_scene

    flag = dict(compute_body_velocity_from_position_delta=True)
    for model, dt, rest, lower in ((quadruped_scene(5, seed=21), 1e-3, False, True), (quadruped_scene(3, seed=22), 1e-3, True, True),
                                   (pendulum_scene(3, seed=4), 2e-3, False, False), (hull_bin_scene(2, 40), 1.0 / 600.0, True, False)):
        if lower:
            _lower(model, 0.25)
        em = H.EmuModel(model)
        s0, s1, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
        plain = H.EmuState(em)
        o = Oracle(model)
        os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
        H.collide(em, s0, ct)
        o.collide(os0.body_q, oc)
        has = int(oc.count[0]) > 0
        H.xpbd_step(em, s0, s1, ctrl, ct if has else None, dt, enable_restitution=rest, **flag)
        H.xpbd_step(em, s0, plain, ctrl, ct if has else None, dt, enable_restitution=False)
        o.xpbd_step(os0, os1, o.control(), oc if has else None, dt, enable_restitution=rest, **flag)
        assert np.array_equal(s1.body_q, plain.body_q)  # the flag only rewrites velocities
        assert not np.array_equal(s1.body_qd, plain.body_qd)
        assert _close(s1.aos("body_q"), os1.body_q, 1e-5) and _close(s1.aos("body_qd"), os1.body_qd, 2e-5 / dt)
        if model.env.np:  # fused rollout == launch-by-launch loop, bitwise
            out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, ct, dt, 3, enable_restitution=rest, **flag)
            a, b = H.EmuState(em), H.EmuState(em)
            for _ in range(3):
                a.body_f[:] = 0
                H.collide(em, a, ct)
                H.xpbd_step(em, a, b, ctrl, ct, dt, enable_restitution=rest, **flag)
                a, b = b, a
            assert np.array_equal(out.body_q, a.body_q) and np.array_equal(out.body_qd, a.body_qd)


@pytest.mark.parametrize("dense", [True, False])
@pytest.mark.parametrize("free_root", [False, True])
def test_featherstone_free_and_distance_joints_below_the_root(H, free_root, dense):
    """solver_featherstone.py:229-265,1006-1046: descendant FREE / DISTANCE joints are integrated in internal parent-origin
    coordinates, then the child pose is re-integrated from its world COM twist, joint_q rebuilt from the poses and the rest of
    the articulation refreshed.  Emulated kernels vs the checker over 25 steps, and fused rollout == per-step loop bitwise."""
    from oracle_bridge import Oracle, OracleState
    from scenes import free_child_scene

    model = free_child_scene(3, seed=31, free_root=free_root)
    rng = np.random.default_rng(5)
    jf = rng.normal(0, 0.5, size=model.joint_dof_count).astype(np.float32)
    em = H.EmuModel(model)
    s0, s1, ctrl = H.EmuState(em), H.EmuState(em), H.EmuControl(em, joint_f=jf)
    o = Oracle(model)
    os0, os1 = OracleState(model), OracleState(model)
    for _ in range(25):
        s0.body_f[:] = 0
        H.featherstone_step(em, s0, s1, ctrl, None, 1e-3, dense=dense)
        os0.body_f[:] = 0
        o.featherstone_step(os0, os1, o.control(joint_f=jf), None, 1e-3)
        s0, s1, os0, os1 = s1, s0, os1, os0
    vtol = 1e-4 if dense else 1e-3  # 25 open-loop steps; the tree-structured mass matrix solves the same system in another order
    assert _close(s0.aos("joint_q"), os0.joint_q, 1e-5) and _close(s0.aos("joint_qd"), os0.joint_qd, vtol)
    assert _close(s0.aos("body_q"), os0.body_q, 1e-5) and _close(s0.aos("body_qd"), os0.body_qd, vtol)
    assert np.abs(os0.body_q - model.body_q).max() > 1e-3  # it moved
    ct = H.EmuContacts(em)
    out = H.featherstone_rollout(em, H.EmuState(em), H.EmuState(em), ctrl, ct, 1e-3, 3, dense=dense)
    a, b = H.EmuState(em), H.EmuState(em)
    for _ in range(3):
        a.body_f[:] = 0
        H.collide(em, a, ct)
        H.featherstone_step(em, a, b, ctrl, ct, 1e-3, dense=dense)
        a, b = b, a
    for k in ("body_q", "body_qd", "joint_q", "joint_qd"):
        assert np.array_equal(getattr(out, k), getattr(a, k)), k
