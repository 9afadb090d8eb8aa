"""Frame-to-frame contact matching: the slot-space gfx950 kernels (emulated here; device test in
tests/test_zx_round2_gpu.py) against oracle/oracle_match.py, the restatement of the reference's key-sorted matcher
(newton/_src/geometry/contact_match.py:266-391).  Scenes: a box stack (5-slot manifolds whose points move a little) advanced by
a few XPBD steps, plus hand-made cases for the thresholds and the one-to-one race."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def H(oracle_lib):
    import harness

    harness.lib()
    return harness


def _history(em):
    from newton_amd import _lib as L

    t = em.t
    ns = max(t.np * t.cpp, 1)
    keep = [np.zeros((3, ns, t.env_stride), np.float32), np.zeros((3, ns, t.env_stride), np.float32), np.zeros((ns, t.env_stride), np.uint8)]
    h = L.nt_contact_history()
    h.prev_pos_world, h.prev_normal, h.prev_live = (k.ctypes.data for k in keep)
    return h, keep


def _flat(em, ct, state_q):
    """Key-sorted flat view of the fixed slots: (keys, midpoints, normals, slot ids [n,2] = (env, slot))."""
    import oracle_match as O

    t, model = em.t, em.model
    ns = t.np * t.cpp
    rows = []
    for env in range(t.env_count):
        for slot in range(ns):
            s0, s1 = int(ct.shape0[slot, env]), int(ct.shape1[slot, env])
            if s0 < 0 or s0 == s1:
                continue
            rows.append((O.sort_key(s0, s1, slot % t.cpp), env, slot, s0, s1))
    rows.sort()
    keys = np.array([r[0] for r in rows], dtype=np.int64)
    p0 = np.array([[ct.data[c, r[2], r[1]] for c in range(0, 3)] for r in rows], dtype=np.float32).reshape(-1, 3)
    p1 = np.array([[ct.data[c, r[2], r[1]] for c in range(3, 6)] for r in rows], dtype=np.float32).reshape(-1, 3)
    nrm = np.array([[ct.data[c, r[2], r[1]] for c in range(12, 15)] for r in rows], dtype=np.float32).reshape(-1, 3)
    mid = O.midpoints(state_q, np.asarray(model.shape_body), [r[3] for r in rows], [r[4] for r in rows], p0, p1)
    return keys, mid, nrm, [(r[1], r[2]) for r in rows]


def _emu_match(H, em, state, ct, hist, pos_thr=0.0005, ndot=0.995, mask=None):
    t = em.t
    out = np.full((max(t.np * t.cpp, 1), t.env_stride), -7, dtype=np.int32)
    ds, dc = state.desc(), ct.desc()
    H.check(H.lib().nt_contacts_match(C.byref(em.desc), C.byref(ds), C.byref(dc), C.byref(hist), pos_thr, ndot,
                                      mask.ctypes.data if mask is not None else None, out.ctypes.data, None), "nt_contacts_match")
    return out


def test_box_stack_manifolds_match_across_frames(H):
    import oracle_match as O
    from scenes import box_stack_scene

    model = box_stack_scene(4, n_boxes=4, seed=2, jitter=2e-3)
    em = H.EmuModel(model)
    a, b, ct, ctrl = H.EmuState(em), H.EmuState(em), H.EmuContacts(em), H.EmuControl(em)
    hist, keep = _history(em)
    prev = None
    for frame in range(3):
        a.body_f[:] = 0
        H.collide(em, a, ct)
        q = a.aos("body_q")
        slots = _emu_match(H, em, a, ct, hist)
        keys, mid, nrm, ids = _flat(em, ct, q)
        if prev is None:
            assert all(slots[s, e] == -1 for e, s in ids)  # no history yet: MATCH_NOT_FOUND everywhere
        else:
            want = O.match(keys, mid, nrm, *prev[:3])
            prev_flat = {es: i for i, es in enumerate(prev[3])}
            got = np.array([slots[s, e] if slots[s, e] < 0 else prev_flat[(e, int(slots[s, e]))] for e, s in ids], dtype=np.int32)
            assert np.array_equal(got, want)
            assert (want >= 0).sum() > 0.5 * len(want)  # resting stack: most manifold points persist
        ds, dc = a.desc(), ct.desc()
        H.check(H.lib().nt_contacts_save_history(C.byref(em.desc), C.byref(ds), C.byref(dc), C.byref(hist), None), "save")
        prev = (keys, mid, nrm, ids)
        H.xpbd_step(em, a, b, ctrl, ct, 1.0 / 240.0, iterations=4)
        a, b = b, a


def test_thresholds_race_and_world_reset(H):
    """Hand-made history: (1) a candidate beyond the position threshold or with a turned normal is MATCH_BROKEN, (2) two new
    contacts closest to the same previous one: the nearer wins, the other is MATCH_BROKEN (no second choice), (3) a pair without
    previous contacts is MATCH_NOT_FOUND, (4) reset worlds report MATCH_NOT_FOUND."""
    import oracle_match as O
    from scenes import box_stack_scene

    model = box_stack_scene(3, n_boxes=2, seed=None, jitter=0.0)
    em = H.EmuModel(model)
    a, ct = H.EmuState(em), H.EmuContacts(em)
    H.collide(em, a, ct)
    t = em.t
    hist, (ppos, pnrm, plive) = _history(em)
    ds, dc = a.desc(), ct.desc()
    H.check(H.lib().nt_contacts_save_history(C.byref(em.desc), C.byref(ds), C.byref(dc), C.byref(hist), None), "save")
    base = _emu_match(H, em, a, ct, hist)
    ids = [(e, s) for e in range(t.env_count) for s in range(t.np * t.cpp) if ct.shape0[s, e] >= 0]
    assert all(base[s, e] == s for e, s in ids)  # identical frame: every contact matches itself
    # (1) move env 0's history of its first live slot by 1 mm; turn the normal of the second
    e0 = [s for e, s in ids if e == 0]
    ppos[0, e0[0], 0] += 1e-3
    pnrm[:, e0[1], 0] = [1.0, 0.0, 0.0]
    m = _emu_match(H, em, a, ct, hist)
    assert m[e0[0], 0] == -2 and m[e0[1], 0] == -2 and all(m[s, 0] == s for s in e0[2:])
    # (2) env 1: make two history entries of one pair coincide with new contact k: both new contacts k and k+1 then see the same
    # closest previous slot only if their own is gone -> drop new k+1's own history entry
    e1 = [s for e, s in ids if e == 1]
    k, k1 = e1[0], e1[1]
    assert k // t.cpp == k1 // t.cpp
    plive[k1, 1] = 0
    m = _emu_match(H, em, a, ct, hist, pos_thr=10.0)  # huge threshold: k+1 now reaches for k's entry, k is nearer (distance 0)
    assert m[k, 1] == k and m[k1, 1] == -2
    # (3) env 2: no history at all for one pair
    p = e1[0] // t.cpp
    plive[p * t.cpp:(p + 1) * t.cpp, 2] = 0
    m = _emu_match(H, em, a, ct, hist)
    assert all(m[s, 2] == -1 for s in range(p * t.cpp, (p + 1) * t.cpp) if ct.shape0[s, 2] >= 0)
    # (4) reset mask
    mask = np.array([0, 1, 0], dtype=np.uint8)
    m = _emu_match(H, em, a, ct, hist, mask=mask)
    assert all(m[s, 1] == -1 for e, s in ids if e == 1)
    # the oracle agrees on the flat view of case (2)
    keys, mid, nrm, fid = _flat(em, ct, a.aos("body_q"))
    live_prev = [(e, s) for e, s in fid if plive[s, e]]
    sel = [i for i, es in enumerate(fid) if plive[es[1], es[0]]]
    pk, pm, pn = keys[sel], np.array([[ppos[c, s, e] for c in range(3)] for e, s in live_prev], np.float32), \
        np.array([[pnrm[c, s, e] for c in range(3)] for e, s in live_prev], np.float32)
    want = O.match(keys, mid, nrm, pk, pm, pn, pos_threshold=10.0)
    m = _emu_match(H, em, a, ct, hist, pos_thr=10.0)
    prev_flat = {es: i for i, es in enumerate(live_prev)}
    got = np.array([m[s, e] if m[s, e] < 0 else prev_flat[(e, int(m[s, e]))] for e, s in fid], dtype=np.int32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,frames", [("box_stack", 6), ("mixed_primitives", 8)])
def test_sticky_replay_reproduces_the_reference_matcher(H, name, frames):
    """contact_matching="sticky" (contact_match.py:530-562,933-996): match -> replay_matched -> save_sorted_state on the fixed slots
    (nt_contacts_match / nt_contacts_replay_matched / nt_contacts_save_history, emulated) against the reference ContactMatcher
    executed frame by frame on the same contacts (tests/golden/make_match_reference_vectors.py): same match indices, same
    replayed body-frame points / offsets / normals."""
    import oracle_match as O
    from scenes import box_stack_scene, mixed_primitive_scene

    from newton_amd import _lib as L

    ref = np.load(os.path.join(ROOT, "tests", "golden", "match_reference_vectors.npz"))
    model = box_stack_scene(1, n_boxes=4, seed=2, jitter=5e-3) if name == "box_stack" else mixed_primitive_scene(1, seed=4)
    em = H.EmuModel(model)
    t = em.t
    ns = t.np * t.cpp
    hist, keep = _history(em)
    body_frame = np.zeros((12, ns, t.env_stride), np.float32)
    hist.prev_body_frame = body_frame.ctypes.data
    ct = H.EmuContacts(em)
    prev_ids = None
    replayed = 0
    for k in range(frames):
        st = H.EmuState(em, body_q=ref[f"{name}/{k}/body_q"])
        H.collide(em, st, ct)
        slots = _emu_match(H, em, st, ct, hist)
        ds, dc = st.desc(), ct.desc()
        H.check(H.lib().nt_contacts_replay_matched(C.byref(em.desc), C.byref(ds), C.byref(dc), C.byref(hist), slots.ctypes.data, None),
                "nt_contacts_replay_matched")
        keys, mid, nrm, ids = _flat(em, ct, st.aos("body_q"))
        assert np.array_equal(keys, ref[f"{name}/{k}/keys"])
        want = ref[f"{name}/{k}/sticky_match"]
        if prev_ids is None:
            assert np.all(want == -1) and all(slots[s, e] == -1 for e, s in ids)
        else:
            prev_flat = {es: i for i, es in enumerate(prev_ids)}
            got = np.array([slots[s, e] if slots[s, e] < 0 else prev_flat[(e, int(slots[s, e]))] for e, s in ids], dtype=np.int32)
            assert np.array_equal(got, want)
        rows = lambda c0: np.array([[ct.data[c0 + c, s, e] for c in range(3)] for e, s in ids], np.float32).reshape(-1, 3)  # noqa: E731
        for field, c0 in (("point0", 0), ("point1", 3), ("offset0", 6), ("offset1", 9), ("normal", 12)):
            assert np.array_equal(rows(c0), ref[f"{name}/{k}/sticky_{field}"]), (k, field)
        replayed += int((np.abs(ref[f"{name}/{k}/sticky_point0"] - ref[f"{name}/{k}/point0"]).max(axis=1) > 0).sum())
        H.check(H.lib().nt_contacts_save_history(C.byref(em.desc), C.byref(ds), C.byref(dc), C.byref(hist), None), "save")
        prev_ids = ids
    assert replayed > 0 or name != "box_stack"


def test_row_matcher_thresholds_race_ties_and_inert_rows(H):
    """nt_flat_rows_match / _save_history / _replay_matched (the SDF legs' rows: contact_match.py:266-391,442-562) on hand-made row
    blocks, against oracle_match: a static world with three candidate pairs; inert rows inside a block; two new rows racing for one
    previous row (nearer wins, equal distance -> smaller key bits); a pair that was not a candidate last frame; sticky replay of
    touching matches only."""
    import oracle_match as O

    from newton_amd import _lib as L

    lib = H.lib()
    E, PPW, cap = 1, 4, 32
    i32, f32 = np.int32, np.float32
    sc = L.nt_sdf_scene()
    shape_body = np.full(8, -1, i32)  # all shapes static: body-frame points are world points
    gap = np.zeros(16, f32)
    tp = np.zeros((1, 2), i32)
    gshape = np.zeros(1, i32)
    sc.env_count, sc.env_stride, sc.nb, sc.ns, sc.shape_local0 = E, 64, 1, 8, 0
    sc.template_pairs, sc.template_pair, sc.gshape_id = 1, tp.ctypes.data, gshape.ctypes.data
    sc.shape_body, sc.shape_gap, sc.pairs_per_world = shape_body.ctypes.data, gap.ctypes.data, PPW
    body_q = np.zeros((7, 1, 64), f32)
    body_q[6] = 1.0

    def frame(pairs):
        """pairs: [(s0, s1, [(key, point, normal, live)])] -> the io struct + its arrays"""
        a = dict(pair_count=np.array([len(pairs)], i32), world_pairs=np.zeros((E * PPW, 2), i32), blk=np.zeros((E * PPW, 2), i32),
                 pair_row=np.zeros(E * PPW, i32), row_start=np.zeros(E + 1, i32), shape0=np.full(cap, -1, i32),
                 shape1=np.full(cap, -1, i32), point0=np.zeros((cap, 3), f32), point1=np.zeros((cap, 3), f32),
                 offset0=np.zeros((cap, 3), f32), offset1=np.zeros((cap, 3), f32), normal=np.zeros((cap, 3), f32),
                 margin0=np.zeros(cap, f32), margin1=np.zeros(cap, f32), key=np.zeros(cap, i32))
        r = 0
        for k, (s0, s1, rows) in enumerate(pairs):
            a["world_pairs"][k] = (s0, s1)
            a["blk"][k] = (0, len(rows))
            a["pair_row"][k] = r
            for key, pt, nrm, live in rows:
                if live:
                    a["shape0"][r], a["shape1"][r] = s0, s1
                a["point0"][r] = a["point1"][r] = pt
                a["offset0"][r] = (r, 0, 0)
                a["normal"][r] = nrm
                a["key"][r] = key
                r += 1
        a["row_start"][1] = r
        io = L.nt_sdf_rows_io()
        for k, v in a.items():
            setattr(io, k, v.ctypes.data)
        io.raw_capacity, io.row_capacity = 1, cap
        return io, a

    hist = dict(prev_row_start=np.zeros(E + 1, i32), prev_pair_count=np.zeros(E, i32), prev_world_pairs=np.zeros((E * PPW, 2), i32),
                prev_pair_row=np.zeros(E * PPW, i32), prev_pair_rows=np.zeros(E * PPW, i32), prev_live=np.zeros(cap, np.uint8),
                prev_pos_world=np.zeros((cap, 3), f32), prev_normal=np.zeros((cap, 3), f32), prev_body_frame=np.zeros((cap, 12), f32),
                prev_claim=np.full(cap, -1, np.int64))
    h = L.nt_flat_history()
    for k, v in hist.items():
        setattr(h, k, v.ctypes.data)
    up, side = (0.0, 0.0, 1.0), (1.0, 0.0, 0.0)
    prev_pairs = [(1, 2, [(8, (0.0, 0.0, 0.0), up, True), (12, (0.1, 0.0, 0.0), up, False), (16, (0.2, 0.0, 0.0), up, True)]),
                  (1, 5, [(4, (1.0, 0.0, 0.0), up, True), (8, (1.0, 0.01, 0.0), up, True)]),
                  (3, 4, [(4, (2.0, 0.0, 0.0), up, True)])]
    io0, a0 = frame(prev_pairs)
    H.check(lib.nt_flat_rows_save_history(C.byref(sc), C.byref(io0), body_q.ctypes.data, C.byref(h), None), "save")
    assert list(hist["prev_live"][:6]) == [1, 0, 1, 1, 1, 1] and np.all(hist["prev_claim"][:6] == -1)
    new_pairs = [
        # pair (1,2): row near prev row 0 (0.2 mm), a second row nearer to it (0.1 mm) -> the first loses the race; a row next to
        # the INERT previous row (no candidate within 0.5 mm -> broken); a row on prev row 2 with a turned normal (broken)
        (1, 2, [(40, (0.0002, 0.0, 0.0), up, True), (44, (0.0, 0.0001, 0.0), up, True), (48, (0.1, 0.0, 0.0), up, True),
                (52, (0.2, 0.0, 0.0), side, True)]),
        # pair (1,5): two rows at equal distance from prev row 3 -> the smaller key wins the tie, the other is broken; an inert row
        (1, 5, [(24, (1.0, 0.0, 0.0003), up, True), (20, (1.0, 0.0, -0.0003), up, True), (28, (1.0, 0.0, 0.0), up, False)]),
        # pair (2,6): was not a candidate last frame -> not found;  pair (3,4): matches
        (2, 6, [(4, (3.0, 0.0, 0.0), up, True)]),
        (3, 4, [(4, (2.0, 0.0001, 0.0), up, True)])]
    io1, a1 = frame(new_pairs)
    m = np.full(cap, -7, i32)
    H.check(lib.nt_flat_rows_match(C.byref(sc), C.byref(io1), body_q.ctypes.data, C.byref(h), 0.0005, 0.995, m.ctypes.data, None), "match")
    assert list(m[:9]) == [-2, 0, -2, -2, -2, 3, -1, -1, 5], m[:9]
    # the oracle on the flat view of the same rows
    def flat(a):
        live = np.flatnonzero(a["shape0"] >= 0)
        keys = np.array([O.sort_key(a["shape0"][r], a["shape1"][r], a["key"][r]) for r in live], np.int64)
        return live, keys, a["point0"][live], a["normal"][live]
    pl, pk, pp, pn = flat(a0)
    nl, nk, npos, nn = flat(a1)
    want = O.match(nk, npos, nn, pk, pp, pn)
    assert np.array_equal(np.where(m[nl] >= 0, np.searchsorted(pl, np.maximum(m[nl], 0)), m[nl]), want)
    # sticky replay: matched rows whose fresh gap <= 0 take the saved record (all margins 0, points coincide: gap = 0 -> replayed)
    a1["margin0"][8] = -1.0  # row 8's fresh gap becomes +1: it keeps its own record
    H.check(lib.nt_flat_rows_replay_matched(C.byref(sc), C.byref(io1), body_q.ctypes.data, C.byref(h), m.ctypes.data, None), "replay")
    assert np.array_equal(a1["point0"][1], a0["point0"][0]) and np.array_equal(a1["offset0"][1], a0["offset0"][0])
    assert np.array_equal(a1["point0"][5], a0["point0"][3]) and a1["offset0"][5][0] == 3.0
    assert a1["point0"][8][1] == np.float32(0.0001) and a1["offset0"][8][0] == 8.0  # not replayed
    assert a1["point0"][0][0] == np.float32(0.0002)  # broken rows keep the fresh record
    # a world whose history was forgotten (prev_pair_count = 0) matches nothing
    hist["prev_pair_count"][0] = 0
    H.check(lib.nt_flat_rows_match(C.byref(sc), C.byref(io1), body_q.ctypes.data, C.byref(h), 0.0005, 0.995, m.ctypes.data, None), "match")
    assert np.all(m[:9] == -1)
    # argument checks
    assert lib.nt_flat_rows_match(C.byref(sc), C.byref(io1), None, C.byref(h), 0.0005, 0.995, m.ctypes.data, None) != 0
    h.prev_body_frame = None
    assert lib.nt_flat_rows_replay_matched(C.byref(sc), C.byref(io1), body_q.ctypes.data, C.byref(h), m.ctypes.data, None) != 0
