"""The mesh-SDF leg INSIDE CollisionPipeline.collide on the MI355X (collide.py:1999 ->
narrow_phase.py:2838-3167): rows against the float32 checker chain, run-to-run bit identity, and the three solvers consuming
the rows (XPBD inside its step kernel, SemiImplicit / Featherstone through the ordered force gather)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

pytestmark = pytest.mark.gpu
N_C5 = int(os.environ.get("NT_FULL_SIZE_C5_ENVS", "2048"))  # BASELINE.json config 5's world count
C5_SETTLE_STEPS = int(os.environ.get("NT_C5_SETTLE_STEPS", "600"))  # (both shrunk only by the emulated dry run of this file)

FIELDS = ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")


def _rows(contacts):
    f = contacts._flat
    n = int(f.row_start[-1].item())
    d = {k: getattr(f, k)[:n].cpu().numpy() for k in (*FIELDS, "key")}
    d["row_start"] = f.row_start.cpu().numpy()
    return d


@pytest.mark.parametrize("walls", [False, True])
def test_collide_rows_match_the_checker_chain(walls):
    import newton_amd as nt
    from sdf_pipeline_checker import checker_rows, sdf_scene

    E = 3
    model = sdf_scene(E, 5, device="cuda:0", walls=walls)
    assert len(model.env.sdf_pair) == (10 + (10 if walls else 0)) and model.env.np == 5  # hull-hull (+ hull-wall) vs hull-plane
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    contacts = pipe.contacts()
    state = model.state()
    pipe.collide(state, contacts)
    got = _rows(contacts)
    ov = pipe._sdf_leg.overflow(contacts._flat)
    assert not ov["overflow"], ov
    leg = pipe._sdf_leg
    X, lo, hi = leg.world_xform.cpu().numpy(), leg.aabb_lower.cpu().numpy(), leg.aabb_upper.cpu().numpy()
    want, cand, (Xc, loc, hic) = checker_rows(model, np.asarray(model.body_q), world_xform=X, aabbs=(lo, hi))
    # the exported world transforms / AABBs themselves (compute_shape_aabbs): against the C++ checker
    finite = np.abs(loc) < 1e5
    assert np.abs(X - Xc).max() <= 1e-6 and np.abs(lo - loc)[finite].max() <= 1e-6 and np.abs(hi - hic)[np.abs(hic) < 1e5].max() <= 1e-6
    # candidate pairs: exact, in order, per world
    pc = pipe._sdf_leg.pair_count.cpu().numpy()
    wp = pipe._sdf_leg.world_pairs.cpu().numpy().reshape(E, -1, 2)
    for w in range(E):
        assert pc[w] == len(cand[w]) and [tuple(p) for p in wp[w, : pc[w]]] == cand[w], w
    assert sum(len(c) for c in cand) > 0
    # rows: ids, fingerprints, per-world ranges exact; geometry to 1e-5
    assert len(want["key"]) == len(got["key"]) > 0
    assert np.array_equal(got["row_start"], np.concatenate([[0], np.cumsum(np.bincount(want["world"], minlength=E))]))
    assert np.array_equal(got["key"], want["key"])
    for k in ("shape0", "shape1"):
        assert np.array_equal(got[k], want[k]), k
    for k in FIELDS[2:]:  # same shape transforms in, same rows out
        assert np.abs(got[k] - want[k]).max() <= 2e-6, (k, np.abs(got[k] - want[k]).max())
    # Newton-shaped view: the slot contacts first, then the live rows
    n_slots = int(contacts.rigid_contact_count_per_env.sum().item())
    live = got["shape0"] != got["shape1"]
    assert int(contacts.rigid_contact_count.item()) == n_slots + int(live.sum())
    assert np.array_equal(contacts.rigid_contact_shape0.cpu().numpy()[n_slots:n_slots + int(live.sum())], got["shape0"][live])


def test_collide_twice_is_bitwise_identical_and_blocks_cover_the_rows():
    import newton_amd as nt
    from sdf_pipeline_checker import sdf_scene

    E = 4
    model = sdf_scene(E, 6, device="cuda:0", walls=True, seed=9)
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    c1, c2 = pipe.contacts(), pipe.contacts()
    state = model.state()
    pipe.collide(state, c1)
    pipe.collide(state, c2)
    a, b = _rows(c1), _rows(c2)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    # every live row sits in exactly the blocks of its two bodies (dynamic ones), ascending
    t = model.env
    f = c1._flat
    bs, bl = f.body_blk_start.cpu().numpy().reshape(E, t.nb + 1), f.body_blk_list.cpu().numpy()
    seen = {}
    for w in range(E):
        for body in range(t.nb):
            last = -1
            for i in range(bs[w, body], bs[w, body + 1]):
                r0, side, cnt = bl[i, 0] >> 1, bl[i, 0] & 1, bl[i, 1]
                assert r0 > last and a["row_start"][w] <= r0 and r0 + cnt <= a["row_start"][w + 1]
                last = r0
                for r in range(r0, r0 + cnt):
                    seen.setdefault(r, []).append((body, side))
    sb = np.asarray(model.shape_body)
    for r in range(len(a["key"])):
        want = []
        for side, s in enumerate((a["shape0"][r], a["shape1"][r])):
            pass
        # the blocks list a pair's rows whether or not a row was gap-rejected; check the live ones
        if a["shape0"][r] != a["shape1"][r]:
            w = int(np.searchsorted(a["row_start"], r, side="right") - 1)
            for side, s in enumerate((a["shape0"][r], a["shape1"][r])):
                if sb[s] >= 0:
                    want.append((int(sb[s]) - w * t.nb, side))
            assert sorted(seen.get(r, [])) == sorted(want), r


def test_staged_narrow_phase_gives_the_rows_of_the_single_kernel(monkeypatch):
    """The dense three-stage narrow phase (cull over all pairs -> one lane per survivor -> reduction per pair; the default) and
    the one-workgroup-per-pair kernel produce the same rows bit for bit, and the survivor list reports its fill."""
    import newton_amd as nt
    from sdf_pipeline_checker import sdf_scene

    model = sdf_scene(3, 7, device="cuda:0", walls=True, seed=4)
    _pile(model)
    state = model.state()
    rows = {}
    for staged in ("1", "0"):
        monkeypatch.setenv("NT_SDF_STAGED", staged)
        pipe = nt.CollisionPipeline(model, broad_phase="sap")
        assert pipe._sdf_leg.staged == (staged == "1")
        c = pipe.contacts()
        pipe.collide(state, c)
        rows[staged] = _rows(c)
        info = pipe._sdf_leg.overflow(c._flat)
        assert not info["overflow"]
        if staged == "1":
            assert info["cull_survivors"] >= info["rows"] > 0 and info["dropped_survivors"] == 0
    for k in rows["1"]:
        assert np.array_equal(rows["1"][k], rows["0"][k]), k


def test_unreduced_rows_are_every_contact_of_the_edge_search():
    """CollisionPipeline(reduce_contacts=False) with mesh-SDF pairs (narrow_phase.py:3044,3097-3130: mesh_sdf_collision_kernel
    writes every contact): the rows are the checker's unreduced contacts -- ids and fingerprints exact, in ascending fingerprint
    per pair, the search's own normal -- the reduced rows are a subset of them with the same geometry up to the reducer's
    octahedral normal, a second call is bitwise identical, and the matcher takes the rows."""
    import newton_amd as nt
    from sdf_pipeline_checker import checker_rows, sdf_scene

    E = 3
    model = sdf_scene(E, 6, device="cuda:0", walls=True, seed=9)
    _pile(model)
    state = model.state()
    pipe = nt.CollisionPipeline(model, broad_phase="sap", reduce_contacts=False, sdf_contacts_per_shape=240,
                                contact_matching="latest")
    contacts = pipe.contacts()
    pipe.collide(state, contacts)
    got = _rows(contacts)
    ov = pipe._sdf_leg.overflow(contacts._flat)
    assert not ov["overflow"], ov
    leg = pipe._sdf_leg
    X, lo, hi = leg.world_xform.cpu().numpy(), leg.aabb_lower.cpu().numpy(), leg.aabb_upper.cpu().numpy()
    want, _, _ = checker_rows(model, state.body_q.cpu().numpy(), world_xform=X, aabbs=(lo, hi), reduce=False)
    assert len(want["key"]) == len(got["key"]) > 0
    assert np.array_equal(got["row_start"], np.concatenate([[0], np.cumsum(np.bincount(want["world"], minlength=E))]))
    assert np.array_equal(got["key"], want["key"])
    for k in ("shape0", "shape1"):
        assert np.array_equal(got[k], want[k]), k
    for k in FIELDS[2:]:
        assert np.abs(got[k] - want[k]).max() <= 2e-6, (k, np.abs(got[k] - want[k]).max())
    # both modes are present and interleave in the fingerprint order of at least one pair
    assert ((got["key"] >> 1) & 1).min() == 0 and ((got["key"] >> 1) & 1).max() == 1
    # the reduced pipeline keeps a subset: same (pair, fingerprint), same points; fewer rows where a pair has many contacts
    red_pipe = nt.CollisionPipeline(model, broad_phase="sap")
    rc = red_pipe.contacts()
    red_pipe.collide(state, rc)
    red = _rows(rc)
    assert 0 < len(red["key"]) <= len(got["key"])
    at = {(int(a), int(b), int(k)): i for i, (a, b, k) in enumerate(zip(got["shape0"], got["shape1"], got["key"]))}
    assert len(at) == len(got["key"])  # a fingerprint once per pair
    idx = np.asarray([at[(int(a), int(b), int(k))] for a, b, k in zip(red["shape0"], red["shape1"], red["key"])])
    for k in ("point0", "point1"):
        assert np.abs(got[k][idx] - red[k]).max() <= 1e-3, k  # (the points shift with the normal's octahedral round trip)
    assert np.einsum("ij,ij->i", got["normal"][idx], red["normal"]).min() > 0.9999
    # run to run
    pipe2 = nt.CollisionPipeline(model, broad_phase="sap", reduce_contacts=False, sdf_contacts_per_shape=240)
    again = pipe2.contacts()
    pipe2.collide(state, again)
    second = _rows(again)
    for k in got:
        assert np.array_equal(got[k], second[k]), k
    # the matcher over unreduced rows: an unchanged second frame matches every live row to itself
    pipe.collide(state, contacts)
    f = contacts._flat
    n = int(f.row_start[-1].item())
    live = (f.shape0[:n] != f.shape1[:n]).cpu().numpy()
    mi = contacts.rigid_contact_match_index.cpu().numpy()
    n_slots = int(contacts.rigid_contact_count_per_env.sum().item())
    row_mi = mi[n_slots:n_slots + int(live.sum())]
    assert (row_mi != -1).all() and (row_mi >= 0).mean() > 0.95  # (coincident twins of two edges may lose their claim: -2)


def test_survivor_list_overflow_is_reported_and_harmless():
    """A survivor list that is too small drops survivors without touching anything outside its stripes: the call reports it
    (`dropped_survivors`, `overflow`), the rows that remain are well-formed, and a second pipeline with the default
    capacity on the same state is unaffected."""
    import newton_amd as nt
    from sdf_pipeline_checker import sdf_scene

    model = sdf_scene(3, 7, device="cuda:0", walls=True, seed=4)
    _pile(model)
    state = model.state()
    full_pipe = nt.CollisionPipeline(model, broad_phase="sap")
    full = full_pipe.contacts()
    full_pipe.collide(state, full)
    want = _rows(full)
    assert not full_pipe._sdf_leg.overflow(full._flat)["overflow"]
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    leg = pipe._sdf_leg
    # shrink the survivor list of this leg to a handful of entries (same buffers, smaller declared capacity)
    leg.hit_capacity, leg.hit_stripe_count = 16, 2
    leg.raw_capacity = leg.hit_capacity + leg.row_capacity
    c = pipe.contacts()
    pipe.collide(state, c)
    info = leg.overflow(c._flat)
    assert info["dropped_survivors"] > 0 and info["overflow"], info
    got = _rows(c)
    live = got["shape0"] != got["shape1"]
    # (with survivors missing other contacts may win a pair's slots: the rows are valid contacts, not a subset of the full result)
    assert live.sum() < (want["shape0"] != want["shape1"]).sum()
    assert np.isfinite(got["point0"][live]).all() and np.isfinite(got["normal"][live]).all()
    assert got["shape0"][live].min() >= 0 and got["shape1"][live].max() < model.shape_count
    again = full_pipe.contacts()
    full_pipe.collide(state, again)
    b = _rows(again)
    for k in want:
        assert np.array_equal(want[k], b[k]), k


def test_row_capacity_overflow_is_reported_and_memory_safe():
    """More rows than the FlatRows arrays hold: the per-world ranges and the per-body block lists stop at the capacity (nothing
    is read or written behind the arrays), the call reports it, rows beyond the declared capacity keep their sentinel, and the
    solvers step on what was kept.  A frame with fewer rows afterwards leaves no stale row alive."""
    import torch

    import newton_amd as nt
    from sdf_pipeline_checker import sdf_scene

    model = sdf_scene(3, 7, device="cuda:0", walls=True, seed=4)
    _pile(model)
    state, out = model.state(), model.state()
    full_pipe = nt.CollisionPipeline(model, broad_phase="sap")
    full = full_pipe.contacts()
    full_pipe.collide(state, full)
    n_full = int(full._flat.row_start[-1].item())
    assert n_full > 40 and not full_pipe._sdf_leg.overflow(full._flat)["overflow"]

    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    c = pipe.contacts()
    f = c._flat
    cap = n_full // 3  # the declared capacity ends inside the second world's rows; the buffers behind it are the real ones
    f.capacity = cap
    for k in FIELDS[2:]:
        getattr(f, k)[cap:] = 777.0  # canaries behind the declared capacity
    f.shape0[cap:] = -7
    f.shape1[cap:] = -7
    f.cw[cap:] = 777.0
    pipe.collide(state, c)
    info = pipe._sdf_leg.overflow(f)
    assert info["overflow"] and info["rows"] == n_full and info["row_capacity"] == cap, info
    rs = f.row_start.cpu().numpy()
    assert rs[-1] == cap and np.all(np.diff(rs) >= 0) and rs.max() <= cap
    for k in FIELDS[2:]:
        assert bool((getattr(f, k)[cap:] == 777.0).all()), k
    assert bool((f.shape0[cap:] == -7).all()) and bool((f.shape1[cap:] == -7).all())
    # the kept rows are the first `cap` rows of the full result, bit for bit
    want, got = _rows(full), _rows(c)
    for k in FIELDS:
        assert np.array_equal(got[k], want[k][:cap]), k
    # every block of every body ends inside the arrays
    t = model.env
    bs = f.body_blk_start.cpu().numpy().reshape(t.env_count, t.nb + 1)
    bl = f.body_blk_list.cpu().numpy()
    for w in range(t.env_count):
        for i in range(bs[w, 0], bs[w, -1]):
            r0, count = bl[i, 0] >> 1, bl[i, 1]
            assert count > 0 and r0 + count <= cap, (w, i, r0, count)
    nt.solvers.SolverXPBD(model, iterations=2).step(state, out, None, c, 1e-3)
    torch.cuda.synchronize()
    assert np.isfinite(out.body_q.cpu().numpy()).all()
    assert bool((f.cw[cap:] == 777.0).all())  # the step's correction records stay inside as well
    # next frame: hulls far apart -> no rows; nothing of the crowded frame survives
    q = np.asarray(model.body_q).copy()
    q[:, :2] *= 50.0
    q[:, 2] += 3.0 + 5.0 * np.tile(np.arange(t.nb), t.env_count)  # also off the walls and off each other
    far = model.state()
    far.body_q = q
    pipe.collide(far, c)
    assert int(f.row_start[-1].item()) == 0 and bool((f.shape0[:cap] == -1).all()) and bool((f.shape1[:cap] == -1).all())


def test_worlds_without_candidates_give_no_rows():
    """Hulls far apart: no candidate pair survives the AABB test, every stage runs on an empty population."""
    import newton_amd as nt
    from sdf_pipeline_checker import sdf_scene

    model = sdf_scene(2, 5, device="cuda:0", walls=False, seed=11)
    q = np.asarray(model.body_q).copy()
    t = model.env
    c = q[:, :3].reshape(t.env_count, t.nb, 3)
    c[:, :, 0] = np.arange(t.nb)[None, :] * 5.0  # 5 m apart
    c[:, :, 2] = 3.0
    q[:, :3] = c.reshape(-1, 3)
    model.body_q = q
    model.joint_q.reshape(-1, 7)[:, :3] = q[:, :3]
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    contacts = pipe.contacts()
    pipe.collide(model.state(), contacts)
    info = pipe._sdf_leg.overflow(contacts._flat)
    assert info["rows"] == 0 and info["pairs_per_world_max"] == 0 and not info["overflow"], info
    assert info["cull_survivors"] == 0


def _oracle_contacts_with_rows(model, o, body_q, rows):
    """Checker contacts = its own slot contacts (tile pairs) + the product's SDF rows appended (live ones)."""
    oc = o.contacts(cmax=max(1000, model.shape_contact_pair_count * 5) + len(rows["key"]))
    o.collide(body_q, oc)
    n = int(oc.count[0])
    live = np.flatnonzero(rows["shape0"] != rows["shape1"])
    for k in FIELDS:
        getattr(oc, k)[n:n + len(live)] = rows[k][live]
    oc.count[0] = n + len(live)
    return oc


def _pile(model, seed=3):
    """Push the hulls of every world together so that the SDF rows carry penetrating contacts."""
    q = np.asarray(model.body_q).copy()
    t = model.env
    c = q[:, :3].reshape(t.env_count, t.nb, 3)
    c[:, :, :2] *= 0.4
    q[:, :3] = c.reshape(-1, 3)
    model.body_q = q
    model.joint_q.reshape(-1, 7)[:, :3] = q[:, :3]


def test_xpbd_step_consumes_the_rows_like_the_checker():
    import torch

    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState
    from sdf_pipeline_checker import sdf_scene

    E = 3
    model = sdf_scene(E, 5, device="cuda:0", seed=11)
    _pile(model)
    # the SDF pairs are not tile pairs: the checker must not run them through MPR / GJK either
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    contacts = pipe.contacts()
    s0, s1 = model.state(), model.state()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    pipe.collide(s0, contacts)
    rows = _rows(contacts)
    assert (rows["shape0"] != rows["shape1"]).sum() > 5
    solver.step(s0, s1, model.control(), contacts, 1.0e-3)
    torch.cuda.synchronize()
    host = _host_twin_without_sdf_pairs(model)
    o = Oracle(host)
    oc = _oracle_contacts_with_rows(host, o, np.asarray(model.body_q), rows)
    os0, os1 = OracleState(host), OracleState(host)
    o.xpbd_step(os0, os1, o.control(), oc, 1.0e-3, iterations=2)
    dq = np.abs(s1.body_q.cpu().numpy() - os1.body_q).max()
    dv = np.abs(s1.body_qd.cpu().numpy() - os1.body_qd).max()
    moved = np.abs(os1.body_q - np.asarray(model.body_q)).max()
    assert dq <= 1e-5 and dv <= 5e-3 and moved > 1e-4, (dq, dv, moved)
    # without the rows the checker ends elsewhere: the rows did take part
    oc2 = o.contacts()
    o.collide(np.asarray(model.body_q), oc2)
    ot0, ot1 = OracleState(host), OracleState(host)
    o.xpbd_step(ot0, ot1, o.control(), oc2, 1.0e-3, iterations=2)
    assert np.abs(ot1.body_q - os1.body_q).max() > 1e-5


@pytest.mark.parametrize("what", ["contact_force", "velocity_from_delta", "restitution"])
def test_xpbd_reporting_and_velocity_update_include_the_rows(what):
    """SolverXPBD on a model whose pairs go through the mesh-SDF leg: Contacts.force covers the rows (their weighted impulses,
    accumulate_weighted_contact_impulse xpbd/kernels.py:2403-2461; constraint_inv_weight counts slots and rows alike) and
    compute_body_velocity_from_position_delta works on the rows' corrections, and the restitution pass (apply_rigid_restitution,
    xpbd/kernels.py:2583-2728) walks the rows behind the slots -- against the checker on the same contact list."""
    import torch

    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState
    from sdf_pipeline_checker import sdf_scene

    model = sdf_scene(3, 5, device="cuda:0", seed=11)
    if what == "contact_force":
        model.request_contact_attributes("force")
    _pile(model)
    if what == "restitution":  # approaching hulls with bouncy materials: the pass has something to do
        rng = np.random.default_rng(5)
        qd = np.zeros((model.body_count, 6), np.float32)
        qd[:, :3] = -2.0 * np.asarray(model.body_q)[:, :3] * np.array([1.0, 1.0, 0.0], np.float32) + rng.normal(0.0, 0.05, (model.body_count, 3))
        model.body_qd = qd
        model.joint_qd = qd.reshape(-1).copy()
        model.shape_material_restitution = np.full(model.shape_count, 0.6, np.float32)
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    contacts = pipe.contacts()
    s0, s1 = model.state(), model.state()
    solver = nt.solvers.SolverXPBD(model, iterations=3, enable_restitution=what == "restitution")
    solver.compute_body_velocity_from_position_delta = what == "velocity_from_delta"
    pipe.collide(s0, contacts)
    rows = _rows(contacts)
    n_rows = int((rows["shape0"] != rows["shape1"]).sum())
    assert n_rows > 5
    dt = 1.0e-3
    solver.step(s0, s1, model.control(), contacts, dt)
    host = _host_twin_without_sdf_pairs(model)
    o = Oracle(host)
    oc = _oracle_contacts_with_rows(host, o, np.asarray(model.body_q), rows)
    n = int(oc.count[0])
    os0, os1 = OracleState(host), OracleState(host)
    want = np.zeros((oc.max, 6), np.float32)
    o.xpbd_step(os0, os1, o.control(), oc, dt, iterations=3, contact_force_out=want if what == "contact_force" else None,
                compute_body_velocity_from_position_delta=what == "velocity_from_delta", enable_restitution=what == "restitution")
    torch.cuda.synchronize()
    dq = np.abs(s1.body_q.cpu().numpy() - os1.body_q).max()
    dv = np.abs(s1.body_qd.cpu().numpy() - os1.body_qd).max()
    assert dq <= 1e-5 and dv <= 5e-3, (dq, dv)
    if what == "contact_force":
        solver.update_contacts(contacts)
        assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == n
        assert np.array_equal(contacts.rigid_contact_shape0.cpu().numpy()[:n], oc.shape0[:n])
        got = contacts.force.cpu().numpy()[:n]
        scale = np.abs(want[:n]).max()
        assert scale > 0 and np.abs(want[n - n_rows:n]).max() > 1e-3 * scale  # the rows carry force
        assert np.abs(got - want[:n]).max() <= 2e-3 * scale, (np.abs(got - want[:n]).max(), scale)
    elif what == "velocity_from_delta":
        assert np.abs(os1.body_qd).max() > 1e-3  # velocities rebuilt from the position change the rows caused
    else:  # the checker without restitution ends with other velocities: the pass acted on the rows
        ot0, ot1 = OracleState(host), OracleState(host)
        o.xpbd_step(ot0, ot1, o.control(), oc, dt, iterations=3)
        assert np.abs(ot1.body_qd - os1.body_qd).max() > 1e-2


def _host_twin_without_sdf_pairs(model):
    """A shallow host copy of the model whose shape_contact_pairs hold only the tile pairs (what the checker's collide walks)."""
    import copy

    t = model.env
    host = copy.copy(model)
    pairs = np.asarray(model.shape_contact_pairs).reshape(t.env_count, -1, 2)
    host.shape_contact_pairs = np.ascontiguousarray(pairs[:, t.tile_pair_index].reshape(-1, 2))
    host.shape_contact_pair_count = len(host.shape_contact_pairs)
    return host


@pytest.mark.parametrize("solver_name", ["semi_implicit", "featherstone"])
def test_penalty_solvers_consume_the_rows_like_the_checker(solver_name):
    import torch

    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState
    from sdf_pipeline_checker import sdf_scene

    E = 3
    model = sdf_scene(E, 5, device="cuda:0", seed=12)
    _pile(model)
    if solver_name == "semi_implicit":  # (SolverFeatherstone rebuilds body_qd from joint_qd: its rows are evaluated at rest)
        model.body_qd = np.random.default_rng(0).uniform(-0.2, 0.2, size=(model.body_count, 6)).astype(np.float32)
    pipe = nt.CollisionPipeline(model, broad_phase="sap")
    contacts = pipe.contacts()
    s0, s1 = model.state(), model.state()
    cls = nt.solvers.SolverSemiImplicit if solver_name == "semi_implicit" else nt.solvers.SolverFeatherstone
    solver = cls(model)
    pipe.collide(s0, contacts)
    rows = _rows(contacts)
    f_before = s0.body_f.cpu().numpy().copy()
    solver.step(s0, s1, model.control(), contacts, 2.5e-4)
    torch.cuda.synchronize()
    assert np.array_equal(s0.body_f.cpu().numpy(), f_before)  # state_in.body_f is not touched
    host = _host_twin_without_sdf_pairs(model)
    o = Oracle(host)
    oc = _oracle_contacts_with_rows(host, o, np.asarray(model.body_q), rows)
    os0, os1 = OracleState(host), OracleState(host)
    if solver_name == "semi_implicit":
        o.semi_implicit_step(os0, os1, o.control(), oc, 2.5e-4)
    else:
        o.featherstone_step(os0, os1, o.control(), oc, 2.5e-4)
    dq = np.abs(s1.body_q.cpu().numpy() - os1.body_q).max()
    dv = np.abs(s1.body_qd.cpu().numpy() - os1.body_qd).max()
    assert dq <= 1e-5 and dv <= 2e-3, (dq, dv)
    oc2 = o.contacts()
    o.collide(np.asarray(model.body_q), oc2)
    ot0, ot1 = OracleState(host), OracleState(host)
    (o.semi_implicit_step if solver_name == "semi_implicit" else o.featherstone_step)(ot0, ot1, o.control(), oc2, 2.5e-4)
    assert np.abs(ot1.body_qd - os1.body_qd).max() > 1e-3  # the rows' forces are in the result


def test_featherstone_rollout_with_sdf_pairs_is_the_launch_by_launch_loop():
    """SolverFeatherstone.rollout on a model whose pairs go through the mesh-SDF leg: the frame is the reference's loop
    (clear_forces, collide incl. the SDF leg, step, swap), bit for bit what the caller's own loop produces."""
    import torch

    import newton_amd as nt
    from sdf_pipeline_checker import sdf_scene

    def run(fused):
        model = sdf_scene(3, 5, device="cuda:0", seed=12)
        _pile(model)
        pipe = nt.CollisionPipeline(model, broad_phase="sap")
        contacts = pipe.contacts()
        s0, s1 = model.state(), model.state()
        solver = nt.solvers.SolverFeatherstone(model)
        if fused:
            out = solver.rollout(s0, s1, None, contacts, 2.5e-4, 5)
        else:
            for _ in range(5):
                s0.clear_forces()
                pipe.collide(s0, contacts)
                solver.step(s0, s1, None, contacts, 2.5e-4)
                s0, s1 = s1, s0
            out = s0
        torch.cuda.synchronize()
        assert int(_rows(contacts)["row_start"][-1]) > 0  # the SDF leg produced rows: the frame went through it
        return out.joint_q.cpu().numpy().copy(), out.body_q.cpu().numpy().copy(), out.body_qd.cpu().numpy().copy()

    a, b = run(True), run(False)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.isfinite(a[1]).all()


_C5_PILE = {}


def _c5_settled_pile(E):
    """Config C5's pile at rest: E worlds x 64 hulls dropped into the bin and settled for 0.5 s by SolverXPBD on the rows of the
    mesh-SDF leg (600 collide + step substeps at dt = 1/1200; both C5 tests start from these poses) -> (model, pipe, state)."""
    import newton_amd as nt
    import scenes

    if E not in _C5_PILE:
        model = scenes.hull_bin_scene(E, 64, device="cuda:0", seed=2, sdf=True, mu=0.5, shape_cfg=dict(gap=0.005))
        if C5_SETTLE_STEPS < 100:  # emulated dry run of the test logic only: no time to fall, squeeze the start lattice instead
            q = np.asarray(model.body_q).copy()
            q[:, :2] *= 0.63
            q[:, 2] = 0.045 + (q[:, 2] - 0.08) * 0.63
            model.body_q = q
            model.joint_q.reshape(-1, 7)[:, :3] = q[:, :3]
        pipe = nt.CollisionPipeline(model, broad_phase="sap")
        c1 = pipe.contacts()
        s0, s1, ctrl = model.state(), model.state(), model.control()
        solver = nt.solvers.SolverXPBD(model, iterations=2)
        for _ in range(C5_SETTLE_STEPS):  # 0.5 s: the hulls drop 2 cm onto the floor and each other
            s0.clear_forces()
            pipe.collide(s0, c1)
            solver.step(s0, s1, ctrl, c1, 1.0 / 1200.0)
            s0, s1 = s1, s0
        _C5_PILE[E] = (model, pipe, s0)
    return _C5_PILE[E]


def test_config_c5_rows_at_full_size():
    """Config C5 at its stated size -- 2 048 worlds x 64 hulls with uint16 texture SDFs in a bin of five SDF boxes, every pair
    through the SDF leg -- settled with SolverXPBD consuming the rows: two collide() calls give bit-identical rows, no capacity is
    exceeded, the pile is at rest inside the bin, and the rows of sampled worlds equal the float32 checker chain (ids, fingerprints
    and counts exact; geometry 2e-6 on the device's own shape transforms, which are held against the checker's to 1e-6)."""
    import torch

    if getattr(torch.cuda, "_newton_emulated", False) and N_C5 > 4:
        pytest.skip("2 048 worlds x 64 hulls: device only (hours in emulation)")
    from sdf_pipeline_checker import checker_rows

    E = N_C5
    model, pipe, s0 = _c5_settled_pile(E)
    t = model.env
    assert t.np == 0 and len(t.sdf_pair) == 64 * 63 // 2 + 64 * 5
    c1, c2 = pipe.contacts(), pipe.contacts()
    pipe.collide(s0, c1)
    pipe.collide(s0, c2)
    torch.cuda.synchronize()
    a, b = _rows(c1), _rows(c2)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    ov = pipe._sdf_leg.overflow(c1._flat)
    assert not ov["overflow"] and ov["rows"] > 20 * E, ov
    q, qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
    assert np.isfinite(q).all() and q[:, 2].min() > 0.0 and q[:, 2].max() < 0.6 and np.abs(q[:, :2]).max() < 0.45
    assert np.median(np.linalg.norm(qd[:, :3], axis=1)) < 0.05
    # sampled worlds against the checker: a slice of the model with the settled poses
    from newton_amd.worlds import slice_worlds

    leg = pipe._sdf_leg
    X, lo, hi = leg.world_xform.cpu().numpy(), leg.aabb_lower.cpu().numpy(), leg.aabb_upper.cpu().numpy()
    pc = leg.pair_count.cpu().numpy()
    for w in range(0, E, max(E // 16, 1)):  # 16 worlds across the batch
        sub = slice_worlds(model, w, w + 1)
        ids = np.asarray(sub._global_shape_ids)
        bq = q[w * t.nb:(w + 1) * t.nb]
        want, cand, (Xc, loc, hic) = checker_rows(sub, bq, world_xform=X[ids], aabbs=(lo[ids], hi[ids]))
        assert np.abs(X[ids] - Xc).max() <= 1e-6 and np.abs(lo[ids] - loc).max() <= 1e-6 and np.abs(hi[ids] - hic).max() <= 1e-6
        r0, r1 = a["row_start"][w], a["row_start"][w + 1]
        assert pc[w] == len(cand[0]) and r1 - r0 == len(want["key"]) > 0, (w, pc[w], len(cand[0]), r1 - r0, len(want["key"]))
        assert np.array_equal(a["key"][r0:r1], want["key"])
        back = {int(g): k for k, g in enumerate(ids)}
        for k in ("shape0", "shape1"):
            got = np.array([back[int(s)] if s >= 0 else -1 for s in a[k][r0:r1]])
            assert np.array_equal(got, want[k]), (w, k)
        for k in FIELDS[2:]:
            assert np.abs(a[k][r0:r1] - want[k]).max() <= 2e-6, (w, k, np.abs(a[k][r0:r1] - want[k]).max())


def test_config_c5_hydroelastic_rows_at_full_size():
    """The hydroelastic half of config C5 at its stated size -- 2 048 worlds x 64 hulls + five walls, every shape HYDROELASTIC
    (kh = 1e10), HydroelasticSDF.Config() as it comes (reduce_contacts, pre_prune_contacts, normal_matching) -- on the settled pile
    of the test above: two collide() calls give bit-identical rows, stiffnesses and friction scales, no capacity is exceeded
    (SdfLeg.overflow()), and the rows of sampled worlds equal the checker chain oracle_hydro.hydro_pipeline (pinned by the executed
    reference, tests/test_hydro_reference_vectors.py): ids, keys, order and counts exact, geometry 2e-6, stiffness 1e-5 relative.
    Reference: sdf_hydroelastic.py:905-1296, contact_reduction_hydroelastic.py:1683-1913."""
    import torch

    if getattr(torch.cuda, "_newton_emulated", False) and N_C5 > 4:
        pytest.skip("2 048 worlds x 64 hulls: device only (hours in emulation)")
    import newton_amd as nt
    import scenes
    from newton_amd.worlds import slice_worlds
    from sdf_pipeline_checker import hydro_checker_rows

    E = N_C5
    _, _, settled = _c5_settled_pile(E)
    q = settled.body_q.cpu().numpy()
    model = scenes.hull_bin_scene(E, 64, device="cuda:0", seed=2, sdf=True, mu=0.5, shape_cfg=dict(gap=0.005), hydroelastic=True)
    t = model.env
    assert t.np == 0 and len(t.sdf_pair) == 64 * 63 // 2 + 64 * 5 and bool(np.all(t.sdf_pair_hydro))
    pipe = nt.CollisionPipeline(model, broad_phase="sap", sdf_hydroelastic_config=nt.geometry.HydroelasticSDF.Config(),
                                sdf_contacts_per_shape=400, sdf_hydro_faces_per_shape=1000)
    c1, c2 = pipe.contacts(), pipe.contacts()
    s0 = model.state()
    s0.body_q = q
    pipe.collide(s0, c1)
    ov = pipe._sdf_leg.overflow(c1._flat)
    assert not ov["overflow"], ov  # (a full face buffer drops faces in arrival order: nothing below would be deterministic)
    pipe.collide(s0, c2)
    torch.cuda.synchronize()
    a, b = _rows(c1), _rows(c2)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    f = c1._flat
    n = len(a["key"])
    stiff, fric = f.stiffness[:n].cpu().numpy(), f.friction_scale[:n].cpu().numpy()
    assert np.array_equal(stiff, c2._flat.stiffness[:n].cpu().numpy()) and np.array_equal(fric, c2._flat.friction_scale[:n].cpu().numpy())
    ov = pipe._sdf_leg.overflow(f)
    assert not ov["overflow"] and ov["rows"] == n > 20 * E and ov["hydro_faces"] > 100 * E, ov
    live = a["shape0"] != a["shape1"]
    assert live.mean() > 0.99 and (stiff >= 0).all() and (stiff[live] > 0).mean() > 0.5, (live.mean(), (stiff > 0).mean(), stiff.min())
    assert np.isfinite(a["point0"]).all() and np.isfinite(a["normal"]).all() and np.isfinite(stiff).all()
    assert np.abs(np.linalg.norm(a["normal"][live], axis=1) - 1.0).max() < 1e-5
    # sampled worlds against the checker chain, on the device's own shape transforms (held against the checker's to 1e-6)
    leg = pipe._sdf_leg
    X, lo, hi = leg.world_xform.cpu().numpy(), leg.aabb_lower.cpu().numpy(), leg.aabb_upper.cpu().numpy()
    pc = leg.pair_count.cpu().numpy()
    for w in range(E // 8, E, max(E // 4, 1)):  # 4 worlds across the batch (the numpy checker needs ~20 s per world)
        sub = slice_worlds(model, w, w + 1)
        ids = np.asarray(sub._global_shape_ids)
        bq = q[w * t.nb:(w + 1) * t.nb]
        want, cand = hydro_checker_rows(sub, bq, X[ids], (lo[ids], hi[ids]), True)
        r0, r1 = a["row_start"][w], a["row_start"][w + 1]
        assert pc[w] == len(cand[0]) and r1 - r0 == len(want["key"]) > 20, (w, pc[w], len(cand[0]), r1 - r0, len(want["key"]))
        assert np.array_equal(a["key"][r0:r1], np.asarray(want["key"]))
        back = {int(g): k for k, g in enumerate(ids)}
        for k in ("shape0", "shape1"):
            got = np.array([back[int(s)] for s in a[k][r0:r1]])
            assert np.array_equal(got, np.asarray(want[k])), (w, k)
        assert np.abs(a["point0"][r0:r1] - np.asarray(want["point0"])).max() <= 2e-6, w
        assert np.abs(a["normal"][r0:r1] - np.asarray(want["normal"])).max() <= 2e-6, w
        ws = np.asarray(want["stiffness"], np.float32)
        assert np.abs(stiff[r0:r1] - ws).max() <= 1e-5 * np.abs(ws).max(), w
        assert np.array_equal(fric[r0:r1], np.ones(r1 - r0, np.float32))


def hydro_scene(world_count, device=None, seed=21):
    """Per world: a hydroelastic box pad, a hydroelastic sphere pressed into it and a hydroelastic hull (mesh SDF) resting on the pad;
    a second, non-hydroelastic hull with an SDF touches the first hull (mesh-SDF edge contacts, the pipeline's other leg)."""
    import newton_amd as nt

    rng = np.random.default_rng(seed)
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = 0.01
    hyd = env.default_shape_cfg.copy()
    hyd.configure_sdf(max_resolution=16, is_hydroelastic=True, kh=2.0e7)
    pad = env.add_body(xform=[0, 0, 0.05, 0, 0, 0, 1])
    env.add_shape_box(pad, hx=0.2, hy=0.2, hz=0.05, cfg=hyd)
    ball = env.add_body(xform=[0.04, -0.03, 0.155, *nt._np_math.quat_rpy(0.2, -0.1, 0.3)])
    env.add_shape_sphere(ball, radius=0.06, cfg=hyd)
    pts = rng.normal(size=(12, 3))
    pts *= 0.06 / np.linalg.norm(pts, axis=1).max()
    mesh = nt.Mesh.convex_hull_of(pts)
    mesh.build_sdf(max_resolution=16, margin=0.02, narrow_band_range=(-0.05, 0.05))
    hcfg = env.default_shape_cfg.copy()
    hcfg.is_hydroelastic, hcfg.kh = True, 5.0e7
    hull = env.add_body(xform=[-0.1, 0.08, 0.14, *nt._np_math.quat_rpy(0.4, 0.2, -0.3)])
    env.add_shape_convex_hull(hull, mesh=mesh, cfg=hcfg)
    mesh2 = nt.Mesh.convex_hull_of(pts * 0.9)
    mesh2.build_sdf(max_resolution=16, margin=0.02, narrow_band_range=(-0.05, 0.05))
    other = env.add_body(xform=[-0.1, 0.17, 0.16, *nt._np_math.quat_rpy(-0.3, 0.1, 0.5)])
    env.add_shape_convex_hull(other, mesh=mesh2)
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = 0.01
    scene.replicate(env, world_count)
    scene.add_ground_plane()
    model = scene.finalize(device=device)
    off = np.random.default_rng(seed + 1).uniform(-0.003, 0.003, size=(model.body_count, 3)).astype(np.float32)
    model.body_q[:, :3] += off
    model.joint_q.reshape(-1, 7)[:, :3] += off
    return model


@pytest.mark.parametrize("reduce", [False, True, "moment"])
def test_hydroelastic_rows_inside_collide_and_their_stiffness_in_the_penalty_solver(reduce):
    """CollisionPipeline(sdf_hydroelastic_config=HydroelasticSDF.Config(reduce_contacts=False)): pairs of two HYDROELASTIC shapes take
    the SDF-SDF leg (SAT, octree, marching cubes: rows with Contacts.rigid_contact_stiffness), other SDF pairs the edge leg, both in
    one row set; against the checker chain (oracle_hydro.hydro_pipeline is pinned by the executed reference), two runs bitwise,
    and SolverSemiImplicit consuming the per-contact stiffness like the checker's eval_body_contact.
    reduce: HydroelasticSDF.Config() as it comes (reduce_contacts, pre_prune_contacts, normal_matching): per-bin aggregates, voxel-
    local pruning, the pair's table, winners with the aggregate stiffness and matched normals (contact_reduction_hydroelastic.py)."""
    import torch

    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState
    from sdf_pipeline_checker import hydro_checker_rows  # (puts oracle/ on the path)

    E = 2
    model = hydro_scene(E, device="cuda:0")
    t = model.env
    # hydroelastic: pad-ball, pad-hull, ball-hull | edge contacts: pad-other, hull-other | tiles: ball-other (MPR / GJK) + four plane pairs
    assert t.sdf_pair_hydro.sum() == 3 and (~t.sdf_pair_hydro).sum() == 2 and t.np == 5
    cfg = nt.geometry.HydroelasticSDF.Config(moment_matching=reduce == "moment") if reduce else nt.geometry.HydroelasticSDF.Config(reduce_contacts=False)
    pipe = nt.CollisionPipeline(model, broad_phase="sap", sdf_hydroelastic_config=cfg, sdf_contacts_per_shape=400)
    c1, c2 = pipe.contacts(), pipe.contacts()
    s0, s1 = model.state(), model.state()
    pipe.collide(s0, c1)
    pipe.collide(s0, c2)
    torch.cuda.synchronize()
    a, b = _rows(c1), _rows(c2)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    f = c1._flat
    n = len(a["key"])
    stiff = f.stiffness[:n].cpu().numpy()
    assert np.array_equal(stiff, c2._flat.stiffness[:n].cpu().numpy())
    assert not pipe._sdf_leg.overflow(f)["overflow"]
    # ---- the checker chain on the device's shape transforms
    leg = pipe._sdf_leg
    X, lo, hi = leg.world_xform.cpu().numpy(), leg.aabb_lower.cpu().numpy(), leg.aabb_upper.cpu().numpy()
    want, _ = hydro_checker_rows(model, np.asarray(model.body_q), X, (lo, hi), reduce)
    lim = (20, 10) if reduce else (100, 50)
    assert len(want["key"]) == n > lim[0] and (np.asarray(want["stiffness"]) > 0).sum() > lim[1] and (np.asarray(want["stiffness"]) == 0).sum() > 0
    assert np.array_equal(a["key"], np.asarray(want["key"])) and np.array_equal(a["shape0"], np.asarray(want["shape0"]))
    assert np.array_equal(a["shape1"], np.asarray(want["shape1"]))
    assert np.abs(a["point0"] - np.asarray(want["point0"])).max() <= 2e-6 and np.abs(a["normal"] - np.asarray(want["normal"])).max() <= 2e-6
    ws = np.asarray(want["stiffness"], np.float32)
    assert np.abs(stiff - ws).max() <= 1e-5 * np.abs(ws).max()
    fric = f.friction_scale[:n].cpu().numpy()
    if reduce == "moment":  # anchors (key 0x400000 | bin) and friction scales that preserve the bins' friction moments
        wf = np.asarray(want["friction"], np.float32)
        assert (np.asarray(want["key"]) >= 0x400000).sum() >= 2 and (np.abs(wf[wf > 0] - 1.0) > 1e-3).sum() > 5
        assert np.abs(fric - wf).max() <= 2e-5
    else:
        assert np.array_equal(fric, np.where((ws > 0) & bool(reduce), 1.0, 0.0).astype(np.float32))  # reduced rows carry scale 1
    # ---- SolverSemiImplicit consumes the per-contact stiffness (kernels_contact.py:452-459) like the checker
    solver = nt.solvers.SolverSemiImplicit(model)
    solver.step(s0, s1, model.control(), c1, 1.0e-4)
    torch.cuda.synchronize()
    host = _host_twin_without_sdf_pairs(model)
    o = Oracle(host)
    oc = o.contacts(cmax=max(1000, host.shape_contact_pair_count * 5) + n)
    o.collide(np.asarray(model.body_q), oc)
    n0 = int(oc.count[0])
    for k in FIELDS:
        getattr(oc, k)[n0:n0 + n] = a[k]
    oc.count[0] = n0 + n
    st = np.zeros(oc.max, np.float32)
    st[n0:n0 + n] = stiff
    fr = np.zeros(oc.max, np.float32)
    fr[n0:n0 + n] = fric
    oc.set_properties(st, np.zeros(oc.max, np.float32), fr)
    os0, os1 = OracleState(host), OracleState(host)
    o.semi_implicit_step(os0, os1, o.control(), oc, 1.0e-4)
    dq, dv = np.abs(s1.body_q.cpu().numpy() - os1.body_q).max(), np.abs(s1.body_qd.cpu().numpy() - os1.body_qd).max()
    assert dq <= 1e-5 and dv <= 1e-3 * max(1.0, np.abs(os1.body_qd).max()), (dq, dv)
    oc2 = o.contacts()
    o.collide(np.asarray(model.body_q), oc2)
    ot0, ot1 = OracleState(host), OracleState(host)
    o.semi_implicit_step(ot0, ot1, o.control(), oc2, 1.0e-4)
    assert np.abs(ot1.body_qd - os1.body_qd).max() > 1e-3  # the hydroelastic patches push


# ---- frame-to-frame matching over the rows (contact_match.py:266-391,442-562; collide.py:2033-2135) -----------------------------
def _exported_with_keys(contacts, state, shape_body):
    """The exported (deterministically ordered) contact list with the reference's sort keys: sub key = sub-contact index for slot
    contacts (consecutive within a pair), the row fingerprint for the SDF leg's rows."""
    import oracle_match as O

    n = int(contacts.rigid_contact_count.cpu().numpy()[0])
    f = {k: getattr(contacts, "rigid_contact_" + k).cpu().numpy()[:n].copy() for k in FIELDS}
    order = contacts.export_order()
    raw = order.cpu().numpy()[:n] if order is not None else np.arange(n)
    n0, live = contacts._flat_n0, contacts._flat_live.cpu().numpy()
    rowkey = contacts._flat.key.cpu().numpy()
    pair = f["shape0"].astype(np.int64) * (1 << 32) + f["shape1"]
    sub = np.zeros(n, dtype=np.int64)
    for i in range(n):
        if raw[i] >= n0:
            sub[i] = rowkey[live[raw[i] - n0]]
        else:
            sub[i] = sub[i - 1] + 1 if i > 0 and pair[i] == pair[i - 1] else 0
    keys = np.array([O.sort_key(a, b, k) for a, b, k in zip(f["shape0"], f["shape1"], sub)], dtype=np.int64)
    assert np.all(np.diff(keys >> 23) >= 0)  # ascending (shape0, shape1): the deterministic order
    mid = O.midpoints(state.body_q.cpu().numpy(), shape_body, f["shape0"], f["shape1"], f["point0"], f["point1"])
    return n, keys, mid, f, raw >= n0


def test_contact_matching_latest_covers_the_sdf_rows_with_report():
    """CollisionPipeline(contact_matching="latest", contact_report=True) on a model whose hull-hull pairs take the mesh-SDF leg:
    rigid_contact_match_index over slot contacts AND rows equals oracle_match on the exported arrays, frame after frame; the
    new / broken reports follow from it; reset_contact_matching(world_mask) forgets the selected worlds only."""
    import torch

    import newton_amd as nt
    import oracle_match as O
    from sdf_pipeline_checker import sdf_scene

    E = 3
    model = sdf_scene(E, 5, device="cuda:0", seed=11)
    _pile(model)
    pipe = nt.CollisionPipeline(model, broad_phase="sap", contact_matching="latest", contact_report=True,
                                contact_matching_pos_threshold=0.004, contact_matching_normal_dot_threshold=0.9)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    s0, s1 = model.state(), model.state()
    shape_body = np.asarray(model.shape_body)
    prev = None
    matched_rows = broken_rows = 0
    for frame in range(5):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        torch.cuda.synchronize()
        n, keys, mid, f, is_row = _exported_with_keys(contacts, s0, shape_body)
        assert is_row.sum() > 5 and (~is_row).sum() > 0
        got = contacts.rigid_contact_match_index.cpu().numpy()[:n]
        new_n = int(contacts.rigid_contact_new_count.cpu().numpy()[0])
        assert np.array_equal(np.sort(contacts.rigid_contact_new_indices.cpu().numpy()[:new_n]), np.flatnonzero(got < 0))
        if prev is None:
            assert np.all(got == -1) and int(contacts.rigid_contact_broken_count.cpu().numpy()[0]) == 0
        else:
            want = O.match(keys, mid, f["normal"], *prev, pos_threshold=0.004, normal_dot_threshold=0.9)
            assert np.array_equal(got, want), (frame, np.flatnonzero(got != want)[:10], got[got != want][:10], want[got != want][:10])
            matched_rows += int((want[is_row] >= 0).sum())
            broken_rows += int((want[is_row] == -2).sum())
            bn = int(contacts.rigid_contact_broken_count.cpu().numpy()[0])
            broken = np.sort(contacts.rigid_contact_broken_indices.cpu().numpy()[:bn])
            assert np.array_equal(broken, np.setdiff1d(np.arange(len(prev[0])), want[want >= 0]))
        prev = (keys, mid, f["normal"])
        solver.step(s0, s1, None, contacts, 1.0e-3)
        s0, s1 = s1, s0
    assert matched_rows > 10, (matched_rows, broken_rows)
    # forget world 1 only: its contacts (slots and rows) report MATCH_NOT_FOUND, the other worlds keep matching
    mask = np.array([False, True, False])
    pipe.reset_contact_matching(mask)
    s0.clear_forces()
    pipe.collide(s0, contacts)
    n, keys, mid, f, is_row = _exported_with_keys(contacts, s0, shape_body)
    got = contacts.rigid_contact_match_index.cpu().numpy()[:n]
    world = np.asarray(model.shape_world)[np.maximum(f["shape0"], f["shape1"])]
    assert np.all(got[world == 1] == -1) and (got[(world != 1) & is_row] >= 0).sum() > 0
    keep = np.asarray(model.shape_world)[(prev[0] >> 23) & 0xFFFFF] != 1
    want = O.match(keys, mid, f["normal"], prev[0][keep], prev[1][keep], prev[2][keep], pos_threshold=0.004, normal_dot_threshold=0.9)
    remap = np.flatnonzero(keep)
    assert np.array_equal(got, np.where(want >= 0, remap[np.maximum(want, 0)], want))
    bn = int(contacts.rigid_contact_broken_count.cpu().numpy()[0])
    assert np.all(keep[contacts.rigid_contact_broken_indices.cpu().numpy()[:bn]])  # rows of the reset world are not "broken"
    pipe.reset_contact_matching()
    pipe.collide(s0, contacts)
    assert np.all(contacts.rigid_contact_match_index.cpu().numpy()[: int(contacts.rigid_contact_count.item())] == -1)


def test_contact_matching_sticky_replays_matched_sdf_rows():
    """contact_matching="sticky" with SDF rows: a matched contact that still touches keeps last frame's body-frame points, offsets
    and normal (contact_match.py:530-562); everything else is the fresh record (a second, non-sticky pipeline on the same state)."""
    import torch

    import newton_amd as nt
    import oracle_match as O
    from sdf_pipeline_checker import sdf_scene

    E = 2
    model = sdf_scene(E, 5, device="cuda:0", seed=11)
    _pile(model)
    kw = dict(broad_phase="sap", contact_matching_pos_threshold=0.004, contact_matching_normal_dot_threshold=0.9)
    pipe = nt.CollisionPipeline(model, contact_matching="sticky", **kw)
    fresh_pipe = nt.CollisionPipeline(model, deterministic=True, broad_phase="sap")
    contacts, fresh = pipe.contacts(), fresh_pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    s0, s1 = model.state(), model.state()
    shape_body = np.asarray(model.shape_body)
    prev = prev_f = None
    replayed_rows = 0
    for frame in range(5):
        s0.clear_forces()
        fresh_pipe.collide(s0, fresh)
        pipe.collide(s0, contacts)
        torch.cuda.synchronize()
        n, keys, _, f, is_row = _exported_with_keys(contacts, s0, shape_body)
        nf, fkeys, fmid, ff, _ = _exported_with_keys(fresh, s0, shape_body)
        assert n == nf and np.array_equal(keys, fkeys)
        got = contacts.rigid_contact_match_index.cpu().numpy()[:n]
        if prev is None:
            assert np.all(got == -1)
        else:  # matching runs on the fresh records against the records saved (after their own replay) last frame
            want = O.match(fkeys, fmid, ff["normal"], *prev, pos_threshold=0.004, normal_dot_threshold=0.9)
            assert np.array_equal(got, want)
        q = s0.body_q.cpu().numpy()
        for i in range(n):
            same = all(np.array_equal(f[k][i], ff[k][i]) for k in FIELDS)
            if got[i] < 0:
                assert same, (frame, i)
                continue
            p = O.midpoints(q, shape_body, [ff["shape0"][i]] * 2, [ff["shape0"][i], ff["shape1"][i]], [ff["point0"][i]] * 2,
                            [ff["point0"][i], ff["point1"][i]])  # row 0: p0 world (both points equal), row 1: the midpoint
            p0w = p[0]
            p1w = 2.0 * p[1] - p0w
            gap = float(np.dot(p1w - p0w, ff["normal"][i]) - (ff["margin0"][i] + ff["margin1"][i]))
            if same and gap > -1e-6:
                continue
            assert gap <= 1e-6, (frame, i, gap)
            for k in ("point0", "point1", "offset0", "offset1", "normal"):
                assert np.array_equal(f[k][i], prev_f[k][got[i]]), (frame, i, k)
            replayed_rows += int(is_row[i])
        mid = O.midpoints(q, shape_body, f["shape0"], f["shape1"], f["point0"], f["point1"])
        prev, prev_f = (keys, mid, f["normal"]), f
        solver.step(s0, s1, None, contacts, 1.0e-3)
        s0, s1 = s1, s0
    assert replayed_rows > 5, replayed_rows


def test_contact_matching_is_refused_for_hydroelastic_pairs():
    import newton_amd as nt

    model = hydro_scene(2, device="cuda:0")
    with pytest.raises(NotImplementedError, match="hydroelastic"):
        nt.CollisionPipeline(model, contact_matching="latest", sdf_hydroelastic_config=nt.geometry.HydroelasticSDF.Config())
