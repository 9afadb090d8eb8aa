"""Parity at BASELINE.json's full sizes (configs C4 / C3 / C2), through the C ABI, on exactly the bench workloads.

The environments are independent, so the serial C++ oracle can still walk all of them in seconds: every environment of the
4096-env quadruped batch and of the 256-env box-stack batch is compared with the oracle (state within the stated fp32
tolerance, per-env contact counts and contact shape ids bit-exact).  On top of that, size-independent properties: an
environment's result does not depend on the batch it is simulated in (bitwise, different workgroup / tile position), nor on
envs_per_block; quaternions stay normalised; two identical launches are bit-identical."""
import os

import numpy as np
import pytest

import tolerances as tol
from test_gpu_parity_xpbd import _compare_contacts, _lower_quadrupeds, _rel, _setup

pytestmark = pytest.mark.gpu
DT = 1e-3
# BASELINE.json's sizes; the emulated dry run of this file (tests/emu/run_gpu_tests_emulated.py) shrinks them through the
# environment because one OS thread per GPU thread is ~10^4 times slower than the device
N_C4 = int(os.environ.get("NT_FULL_SIZE_C4_ENVS", "4096"))
N_C5 = int(os.environ.get("NT_FULL_SIZE_C5_ENVS", "2048"))


def _per_env_counts(model, oc):
    t = model.env
    n = int(oc.count[0])
    s0, s1 = oc.shape0[:n], oc.shape1[:n]
    L0, nloc = t.shape_local0, max(t.ns, 1)
    loc0 = (s0 >= L0) & (s0 < L0 + t.env_count * t.ns)
    env_of = np.where(loc0, (s0 - L0) // nloc, (s1 - L0) // nloc)
    return np.bincount(env_of, minlength=t.env_count)


@pytest.mark.parametrize("lowered", [False, True])
def test_c4_4096_quadrupeds_one_frame_vs_oracle(lowered):
    """bench.py's N=1 workload (4096 envs, XPBD iterations=2, one frame = 10 fused substeps); `lowered` puts the feet into
    the ground so the contact solve is active in every environment."""
    from oracle_bridge import OracleState
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, 4096, seed=1)
    if lowered:
        _lower_quadrupeds(nt, model, 0.26)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()

    # identical inputs: the whole contact set of all 4096 environments is bit-exact (pairs, counts, ids), geometry <= 1e-5
    pipe.collide(s0, contacts)
    pairs, _, _ = o.collide(os0.body_q, oc)
    _compare_contacts(model, contacts, oc, pairs)
    if lowered:
        assert int(oc.count[0]) >= 4096 * 4

    out = nt.solvers.SolverXPBD(model, iterations=2).rollout(s0, s1, None, contacts, DT, 10)
    oout = o.xpbd_rollout(os0, os1, o.control(), oc, DT, 10, iterations=2)
    q, qd = out.body_q.cpu().numpy(), out.body_qd.cpu().numpy()
    # true relative errors (tests/tolerances.py): positions / rotations <= 1e-4 (north_star); XPBD velocities are position
    # differences / dt, so they are gated on the absolute error that a few fp32 position ulps produce after the division
    gates = dict(pos=1e-5, rot=1e-5, lin_vel_abs=tol.velocity_ulp_bound(1.0, DT, 8), ang_vel_abs=4e-3)
    if not lowered:  # no threshold event in a frame of PD hold above the ground: plain maxima (measured 5.3e-7 / 1.8e-7 / 2.0e-4 / 1.4e-4)
        tol.check(f"c4_4096_quadrupeds_frame lowered={lowered}", q, qd, oout.body_q, oout.body_qd, **gates)
    else:  # all feet in the ground for ten substeps: per-environment gates (tests/tolerances.py:check_rollout; measured on the
        # MI355X: median 7e-8, p99 6.4e-7, one environment of 4096 a threshold event apart at 3.0e-4)
        tol.check_rollout(f"c4_4096_quadrupeds_frame lowered={lowered}", q, qd, oout.body_q, oout.body_qd, model.env.nb, **gates)
    assert np.all(np.abs(np.linalg.norm(q[:, 3:], axis=1) - 1.0) < 1e-5)
    if lowered:
        _explain_c4_outliers(nt, model, o, q, qd)
    # contacts of the 10th substep come from states that already differ by rounding: a contact sitting within ~1e-6 of
    # the gap threshold may flip in a handful of the 4096 x 13 pairs, everything else must agree exactly
    got, want = contacts.rigid_contact_count_per_env.cpu().numpy(), _per_env_counts(model, oc)
    assert np.mean(got == want) >= 0.999
    assert abs(int(got.sum()) - int(want.sum())) <= 8


def _explain_c4_outliers(nt, model, o, q_frame, qd_frame):
    """The outlier allowance of check_rollout, tested (tolerances.explain_rollout_outliers): the frame again substep by substep on
    the device (ten one-substep rollouts: bitwise the ten-substep launch) and on the oracle, then every environment beyond 1e-5 must
    re-join the oracle when the oracle restarts from the device's own state next to the environment's first divergent substep."""
    from oracle_bridge import OracleState

    s0, s1 = model.state(), model.state()
    contacts = nt.CollisionPipeline(model).contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    gpu, ora = [(s0.body_q.cpu().numpy().copy(), s0.body_qd.cpu().numpy().copy())], [(os0.body_q.copy(), os0.body_qd.copy())]
    a, b, oa, ob = s0, s1, os0, os1
    TAIL = 2  # substeps past the frame: an environment that first parts from the oracle in the frame's LAST substep re-joins in these
    for _ in range(10 + TAIL):
        out = solver.rollout(a, b, None, contacts, DT, 1)
        a, b = (b, a) if out is b else (a, b)
        gpu.append((a.body_q.cpu().numpy().copy(), a.body_qd.cpu().numpy().copy()))
        oout = o.xpbd_rollout(oa, ob, o.control(), oc, DT, 1, iterations=2)
        oa, ob = (ob, oa) if oout is ob else (oa, ob)
        ora.append((oa.body_q.copy(), oa.body_qd.copy()))
    assert np.array_equal(gpu[10][0], q_frame) and np.array_equal(gpu[10][1], qd_frame)  # ten launches of one substep == one of ten

    def restart(k, q, qd):
        ra, rb = OracleState(model), OracleState(model)
        ra.body_q[:], ra.body_qd[:] = q, qd
        states = []
        for _ in range(10 + TAIL - k):
            rout = o.xpbd_rollout(ra, rb, o.control(), oc, DT, 1, iterations=2)
            ra, rb = (rb, ra) if rout is rb else (ra, rb)
            states.append((ra.body_q.copy(), ra.body_qd.copy()))
        return states

    return tol.explain_rollout_outliers("c4_4096_quadrupeds_frame lowered=True", gpu, ora, restart, model.env.nb, frame=10)


def test_c4_env_result_is_independent_of_batch_and_tile():
    """Environment k of the 4096-env batch == the same environment simulated in a 96-env batch (same seed => same
    per-env jitter), bitwise, and for every envs_per_block; a relaunch is bit-identical."""
    from scenes import quadruped_scene

    import newton_amd as nt

    def run(n, epb):
        model = quadruped_scene(n, device="cuda:0", seed=1)
        _lower_quadrupeds(nt, model, 0.24)
        s0, s1 = model.state(), model.state()
        contacts = nt.CollisionPipeline(model, envs_per_block=epb).contacts()
        out = nt.solvers.SolverXPBD(model, iterations=2, envs_per_block=epb).rollout(s0, s1, None, contacts, DT, 10)
        nb = model.env.nb
        return out.body_q.cpu().numpy().reshape(n, nb, 7), out.body_qd.cpu().numpy().reshape(n, nb, 6)

    q_full, qd_full = run(4096, 0)
    q_again, qd_again = run(4096, 0)
    assert np.array_equal(q_full, q_again) and np.array_equal(qd_full, qd_again)
    for n, epb in ((96, 0), (96, 8), (40, 16), (24, 1)):  # wider tiles do not fit this model's LDS footprint
        q, qd = run(n, epb)
        assert np.array_equal(q, q_full[:n]), (n, epb)
        assert np.array_equal(qd, qd_full[:n]), (n, epb)


def test_c3_4096_quadrupeds_featherstone_frame_vs_oracle():
    """Config C3 at full size: SolverFeatherstone, 4096 envs, 10 fused substeps of PD hold in free flight (open loop; the frame
    with live contacts is the next test)."""
    from oracle_bridge import OracleState
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, N_C4, seed=1)
    s0, s1 = model.state(), model.state()
    contacts = nt.CollisionPipeline(model).contacts()
    out = nt.solvers.SolverFeatherstone(model).rollout(s0, s1, None, contacts, DT, 10)

    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    c = o.control()
    for _ in range(10):
        os0.body_f[:] = 0
        o.collide(os0.body_q, oc)
        o.featherstone_step(os0, os1, c, oc, DT)
        os0, os1 = os1, os0
    assert _rel(out.joint_q.cpu().numpy(), os0.joint_q) <= 1e-4
    tol.check("c3_4096_quadrupeds_featherstone_frame", out.body_q.cpu().numpy(), out.body_qd.cpu().numpy(), os0.body_q,
              os0.body_qd, pos=1e-6, rot=1e-6, lin_vel=1e-5, ang_vel=1e-5)  # measured: 3e-12 / 6e-8 / 1.7e-7 / 5.9e-8
    jqd, jqd_ref = out.joint_qd.cpu().numpy(), os0.joint_qd
    assert float(np.max(np.abs(jqd - jqd_ref) / np.maximum(np.abs(jqd_ref), 0.05))) <= 1e-3
    assert np.array_equal(contacts.rigid_contact_count_per_env.cpu().numpy(), _per_env_counts(model, oc))


@pytest.mark.parametrize("mass_matrix", ["tree", "dense"])
def test_c3_4096_quadrupeds_featherstone_frame_with_live_contacts(mass_matrix):
    """(Both mass-matrix modes of SolverFeatherstone: the default tree-structured one and the reference's dense operation order,
    at the same gates.)  Config C3 at full size WITH contacts: the feet of all 4096 robots pressed into the ground, the 10 substeps of one frame.
    SolverFeatherstone's contacts are explicit penalty forces (ke ~ 1e4 on light feet): they amplify a rounding difference 3 - 10x
    per substep (tests/test_gpu_parity_featherstone.py::test_quadruped_impact_phase_stepwise; measured here: an open-loop frame
    ends 1.4e-3 apart in joint_q), so the frame is compared substep by substep from the oracle's state -- every substep on
    identical inputs: contact counts per environment exact, state <= 1e-5 / velocities <= 5e-4 in all but 0.2 % of the
    environments (those within 20x) -- and the
    fused 10-substep rollout from the same start is held against the call-by-call loop bit for bit."""
    from oracle_bridge import OracleState
    from scenes import quadruped_scene

    nt, model, o = _setup(quadruped_scene, N_C4, seed=1)
    _lower_quadrupeds(nt, model, 0.26)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverFeatherstone(model, mass_matrix=mass_matrix)
    os0, os1 = OracleState(model), OracleState(model)
    oc, c = o.contacts(), o.control()
    for k in range(10):
        s0.joint_q, s0.joint_qd, s0.body_q, s0.body_qd = os0.joint_q, os0.joint_qd, os0.body_q, os0.body_qd
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, DT)
        os0.body_f[:] = 0
        o.collide(os0.body_q, oc)
        assert int(oc.count[0]) >= N_C4 * 4, k  # live contacts in every environment, every substep
        assert np.array_equal(contacts.rigid_contact_count_per_env.cpu().numpy(), _per_env_counts(model, oc)), k
        o.featherstone_step(os0, os1, c, oc, DT)
        # per environment, relative to the environment's own velocity scale; (6 environments x 120 steps pass 1e-5 / 5e-4 in
        # test_quadruped_impact_phase_stepwise; over 4096 environments x 10 substeps single environments sit on a stiff explicit
        # contact: gated like an open-loop XPBD frame -- all but 0.2 % of the environments inside, the rest within 20x)
        E, nc, nd = N_C4, model.env.nc, model.env.nd
        gq, wq = s1.joint_q.cpu().numpy().reshape(E, nc), os1.joint_q.reshape(E, nc)
        gv, wv = s1.joint_qd.cpu().numpy().reshape(E, nd), os1.joint_qd.reshape(E, nd)
        eq = np.abs(gq - wq).max(axis=1) / np.maximum(np.abs(wq).max(axis=1), 1.0)
        ev = np.abs(gv - wv).max(axis=1) / np.maximum(np.abs(wv).max(axis=1), 1.0)
        worst = int(np.argmax(ev))
        info = dict(step=k, q_max=float(eq.max()), v_max=float(ev.max()), q_out=int((eq > 1e-5).sum()), v_out=int((ev > 5e-4).sum()),
                    worst_env=worst, worst_env_speed=float(np.abs(wv[worst]).max()))
        print(f"[c3 live contacts, mass_matrix={mass_matrix}]", info)
        assert (eq > 1e-5).sum() <= 0.002 * E and (ev > 5e-4).sum() <= 0.002 * E, info
        assert eq.max() <= 2e-4 and ev.max() <= 0.05, info
        assert _rel(s1.body_q.cpu().numpy(), os1.body_q) <= 2e-4, k
        os0, os1 = os1, os0
    # the fused frame == the call-by-call frame, bit for bit, with the contacts live
    a0, a1, b0, b1 = model.state(), model.state(), model.state(), model.state()
    ca, cb = pipe.contacts(), pipe.contacts()
    out = solver.rollout(a0, a1, None, ca, DT, 10)
    for _ in range(10):
        b0.clear_forces()
        pipe.collide(b0, cb)
        solver.step(b0, b1, None, cb, DT)
        b0, b1 = b1, b0
    for name in ("joint_q", "joint_qd", "body_q", "body_qd"):
        assert np.array_equal(getattr(out, name).cpu().numpy(), getattr(b0, name).cpu().numpy()), name
    assert np.isfinite(out.body_q.cpu().numpy()).all()


@pytest.mark.parametrize("broad_phase", ["explicit", "nxn"])
def test_c2_256_box_stacks_vs_oracle(broad_phase):
    """Config C2 at full size: 256 envs x 8 stacked boxes, XPBD iterations=4, dt=1/240, box-box through MPR/GJK + manifold."""
    from oracle_bridge import OracleState
    from scenes import box_stack_scene

    nt, model, o = _setup(box_stack_scene, 256, seed=0)
    dt = 1.0 / 240.0
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model, broad_phase=broad_phase)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    c = o.control()
    for _ in range(3):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, dt)
        s0, s1 = s1, s0
        os0.body_f[:] = 0
        pairs, _, _ = o.collide(os0.body_q, oc, broad_phase=broad_phase)
        o.xpbd_step(os0, os1, c, oc, dt, iterations=4)
        os0, os1 = os1, os0
    n = int(oc.count[0])
    assert n >= 256 * 8 * 4
    assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == n
    assert np.array_equal(contacts.rigid_contact_count_per_env.cpu().numpy(), _per_env_counts(model, oc))
    t = model.env
    mask = contacts.candidate_pair_mask.cpu().numpy()
    got = {tuple(p) for p in np.asarray(model.shape_contact_pairs).reshape(t.env_count, t.np, 2)[mask]}
    assert got == {tuple(p) for p in pairs}
    tol.check(f"c2_256_box_stacks broad_phase={broad_phase}", s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy(), os0.body_q,
              os0.body_qd, pos=1e-6, rot=1e-6, lin_vel=1e-5, ang_vel=1e-5)  # measured: bit-identical positions, 1e-11 velocities


def test_c4_convex_variant_4096_box_quadrupeds_vs_oracle():
    """Config C4's convex-convex variant: box links on a box slab, every one of the 13 candidate pairs
    of an environment is box-box through MPR/GJK + manifold; 4096 envs, XPBD iterations=2, one frame of 10 fused substeps
    with the feet lowered into the slab so that the convex contacts are active in every environment."""
    from oracle_bridge import OracleState
    from scenes import quadruped_convex_scene

    nt, model, o = _setup(quadruped_convex_scene, N_C4, seed=1)
    assert model.env.np_analytic == 0 and model.env.np == 13 and model.env.cpp == 5
    _lower_quadrupeds(nt, model, 0.26)
    s0, s1 = model.state(), model.state()
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    os0, os1 = OracleState(model), OracleState(model)
    oc = o.contacts()
    pipe.collide(s0, contacts)
    pairs, _, _ = o.collide(os0.body_q, oc)
    _compare_contacts(model, contacts, oc, pairs)
    assert int(oc.count[0]) >= N_C4 * 4  # every foot (a tilted box corner or edge) touches the slab
    out = nt.solvers.SolverXPBD(model, iterations=2).rollout(s0, s1, None, contacts, DT, 10)
    oout = o.xpbd_rollout(os0, os1, o.control(), oc, DT, 10, iterations=2)
    tol.check("c4_convex_variant_4096_frame", out.body_q.cpu().numpy(), out.body_qd.cpu().numpy(), oout.body_q, oout.body_qd,
              pos=1e-5, rot=1e-5, lin_vel_abs=tol.velocity_ulp_bound(1.0, DT, 8), ang_vel_abs=4e-3)  # measured: 1.2e-6 / 1.9e-6 /
    # 3.1e-4 m/s / 1.4e-3 rad/s
    got, want = contacts.rigid_contact_count_per_env.cpu().numpy(), _per_env_counts(model, oc)
    assert np.mean(got == want) >= 0.995


def test_c5_geometry_2048_envs_64_hulls_vs_oracle():
    """Config C5's geometry at BASELINE.json's size: 2048 envs x 64 convex hulls in a five-wall bin (2 336 candidate pairs per
    env, 4.8 M in total) -- one collide + one XPBD step on the device; the serial oracle walks a 256-env sample (worlds
    [0,128) and [1920,2048), sliced out of the same model) and everything is compared on it: candidate-pair set, per-env
    contact counts, contact ids (bit-exact), contact geometry (<= 1e-5) and the stepped state."""
    import torch
    from oracle_bridge import Oracle, OracleState
    from scenes import hull_bin_scene

    from newton_amd.worlds import slice_worlds

    import newton_amd as nt

    E = N_C5
    S = min(128, E // 2)  # oracle sample: the first and the last S environments
    model = hull_bin_scene(E, 64, device="cuda:0")
    t = model.env
    assert t.np == 2336 and model.device_model().desc.contact_scratch_in_hbm == 1
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    s0, s1 = model.state(), model.state()
    dt = 1.0 / 600.0
    pipe.collide(s0, contacts)
    solver.step(s0, s1, None, contacts, dt)
    torch.cuda.synchronize()
    counts = contacts.rigid_contact_count_per_env.cpu().numpy()
    mask = contacts.candidate_pair_mask.cpu().numpy()  # [E, np]
    q1 = s1.body_q.cpu().numpy().reshape(E, t.nb, 7)
    qd1 = s1.body_qd.cpu().numpy().reshape(E, t.nb, 6)
    n_total = int(contacts.rigid_contact_count.cpu().numpy()[0])
    assert n_total == int(counts.sum()) and n_total > 5 * 64 * E // 4
    assert len(np.unique(counts)) > min(8, E // 2)  # the per-env jitter makes the environments differ
    assert t.np_analytic == 0 and t.shape_local0 == 0  # all pairs convex => the flat export is env-major; globals at the tail
    flat = {k: getattr(contacts, "rigid_contact_" + k).cpu().numpy()[:n_total]
            for k in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")}
    first = np.concatenate([[0], np.cumsum(counts)])
    for b, e in ((0, S), (E - S, E)):
        sub = slice_worlds(model, b, e, device="cpu")
        o = Oracle(sub)
        os0, os1, oc = OracleState(sub), OracleState(sub), o.contacts()
        pairs, _, _ = o.collide(os0.body_q, oc)
        n = int(oc.count[0])
        assert np.array_equal(counts[b:e], _per_env_counts(sub, oc))
        sub_pairs = np.asarray(sub.shape_contact_pairs).reshape(e - b, t.np, 2)
        assert {tuple(p) for p in sub_pairs[mask[b:e]]} == {tuple(p) for p in pairs}
        lo, hi = int(first[b]), int(first[e])
        assert hi - lo == n

        def to_sub(ids):  # global Newton shape id -> id inside the slice
            return np.where(ids < E * t.ns, ids - b * t.ns, ids - E * t.ns + (e - b) * t.ns)

        assert np.array_equal(to_sub(flat["shape0"][lo:hi]), oc.shape0[:n])
        assert np.array_equal(to_sub(flat["shape1"][lo:hi]), oc.shape1[:n])
        for k in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
            assert np.max(np.abs(flat[k][lo:hi] - getattr(oc, k)[:n])) <= 1e-5, k
        o.xpbd_step(os0, os1, o.control(), oc, dt)
        tol.check(f"c5_geometry_2048 envs[{b},{e})", q1[b:e].reshape(-1, 7), qd1[b:e].reshape(-1, 6), os1.body_q, os1.body_qd,
                  pos=1e-5, rot=1e-5, lin_vel=1e-4, ang_vel_abs=4e-3)  # (the drop state: contacts are admitted by the gap but none
        # penetrates yet, so this step checks the integrator and the slot bookkeeping of the pair-heavy tile, not its contact solve)

    # ---- the same comparison on the SETTLED pile: the hulls are dropped and settled on the device
    # (fused rollouts of the pair-heavy tile), then ONE collide + step from that state on the device and -- from the identical state --
    # on the checker.  Hundreds of contacts per world penetrate, every body of the pile is corrected through nt_contacts.cw.
    SETTLE = int(os.environ.get("NT_FULL_SIZE_C5_SETTLE_FRAMES", "30"))
    dts = 1.0 / 1200.0
    cur, other = s0, s1
    for _ in range(SETTLE):
        out = solver.rollout(cur, other, None, contacts, dts, 10)
        if out is other:
            cur, other = other, cur
    torch.cuda.synchronize()
    qs, qds = cur.body_q.cpu().numpy().reshape(E, t.nb, 7), cur.body_qd.cpu().numpy().reshape(E, t.nb, 6)
    assert np.all(np.isfinite(qs)) and float(qs[:, :, 2].min()) > -0.01  # a pile inside the bin
    cur.clear_forces()
    pipe.collide(cur, contacts)
    solver.step(cur, other, None, contacts, dts)
    torch.cuda.synchronize()
    counts = contacts.rigid_contact_count_per_env.cpu().numpy()
    q2 = other.body_q.cpu().numpy().reshape(E, t.nb, 7)
    qd2 = other.body_qd.cpu().numpy().reshape(E, t.nb, 6)
    for b, e in ((0, S), (E - S, E)):
        sub = slice_worlds(model, b, e, device="cpu")
        o = Oracle(sub)
        os0 = OracleState(sub, qs[b:e].reshape(-1, 7), qds[b:e].reshape(-1, 6))
        os1, os_free, oc = OracleState(sub), OracleState(sub), o.contacts()
        o.collide(os0.body_q, oc)
        assert np.array_equal(counts[b:e], _per_env_counts(sub, oc))  # identical inputs: per-world contact counts exact
        o.xpbd_step(os0, os1, o.control(), oc, dts)
        o.xpbd_step(os0, os_free, o.control(), None, dts)  # the same step without contacts: what the contact solve contributes
        touched = np.linalg.norm(os1.body_qd[:, :3] - os_free.body_qd[:, :3], axis=1) > 1e-3
        assert touched.mean() > 0.5, touched.mean()  # most bodies of the pile are corrected by penetrating contacts
        errs = tol.check(f"c5_geometry_2048 settled envs[{b},{e})", q2[b:e].reshape(-1, 7), qd2[b:e].reshape(-1, 6), os1.body_q, os1.body_qd,
                         pos=1e-5, rot=2e-5, lin_vel_abs=tol.velocity_ulp_bound(1.0, dts, 16), ang_vel_abs=2e-2)
        # (hulls of 3-6 cm radius: one position ulp at the contact point is ~1e-3 rad/s after the division by dt and the lever arm)
        assert errs["lin_vel_abs"]["max"] > 0.0  # not a vacuous comparison: device and checker round differently somewhere
