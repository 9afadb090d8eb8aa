"""Standalone broad phases on raw AABB arrays (newton.geometry.BroadPhaseAllPairs / BroadPhaseSAP / BroadPhaseExplicit),
restated from newton/tests/test_broad_phase.py:78-145,399-1465,2272-2330: random boxes with exclusive / shared collision
groups, several worlds plus shared (-1) shapes, visual-only shapes filtered by flags, per-shape gaps, excluded pairs, the
immovable-pair filter and capacity overflow -- the candidate SET must equal a numpy brute force exactly.

CPU: the oracle (N x N and sort-and-sweep), the host-side world map, and the kernels' own pair logic compiled for the host
(tools/broadphase_host_check.cpp shares csrc/nt_broadphase_core.hpp with the HIP kernels).  GPU: the three classes."""
import os
import subprocess
import sys
from math import sqrt

import ctypes as C
import numpy as np
import pytest

from newton_amd.enums import ShapeFlags
from newton_amd.geometry import precompute_world_map

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))  # broadphase_cases: the case table shared with the fixture generator
_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)


def _group_pair(a, b):
    if a == 0 or b == 0:
        return False
    if a > 0:
        return a == b or b < 0
    return a != b


def brute_force(lower, upper, gap, group, world, flags=None, filter_pairs=()):
    """find_overlapping_pairs_np (test_broad_phase.py:93-145) with the kernel's gap rule (sum of the two gaps)."""
    n = lower.shape[0]
    filt = {tuple(p) for p in filter_pairs}
    out = set()
    for i in range(n):
        if flags is not None and not flags[i] & int(ShapeFlags.COLLIDE_SHAPES):
            continue
        for j in range(i + 1, n):
            if flags is not None and not flags[j] & int(ShapeFlags.COLLIDE_SHAPES):
                continue
            if world[i] != -1 and world[j] != -1 and world[i] != world[j]:
                continue
            if not _group_pair(int(group[i]), int(group[j])) or (i, j) in filt:
                continue
            c = np.float32(gap[i]) + np.float32(gap[j])
            if np.all(lower[i] <= upper[j] + c) and np.all(upper[i] >= lower[j] - c):
                out.add((i, j))
    return out


def make_case(name):
    """The reference's random configurations (same generators, seeds and distributions)."""
    if name == "single_world":  # test_nxn_broadphase: 30 boxes, one world
        rng = np.random.Generator(np.random.PCG64(42))
        n = 30
        centers, sizes = rng.random((n, 3)) * 3.0, rng.random((n, 3)) * 2.0
        group = rng.integers(1, 6, size=n, dtype=np.int32)
        group[rng.choice(n, size=int(sqrt(n)), replace=False)] = -1
        world = np.zeros(n, dtype=np.int32)
        flags = None
    elif name == "multiple_worlds":  # test_nxn_broadphase_multiple_worlds: 50 boxes, 4 worlds + shared shapes
        rng = np.random.Generator(np.random.PCG64(123))
        n = 50
        centers, sizes = rng.random((n, 3)) * 5.0, rng.random((n, 3)) * 1.5
        group = rng.integers(1, 6, size=n, dtype=np.int32)
        group[rng.choice(n, size=int(sqrt(n)), replace=False)] = -1
        world = rng.integers(0, 4, size=n, dtype=np.int32)
        world[rng.choice(n, size=max(3, n // 10), replace=False)] = -1
        flags = None
    elif name == "shape_flags":  # test_nxn_broadphase_with_shape_flags: a third of the shapes are visual-only
        rng = np.random.Generator(np.random.PCG64(456))
        n = 40
        centers, sizes = rng.random((n, 3)) * 4.0, rng.random((n, 3)) * 1.5
        group = rng.integers(1, 4, size=n, dtype=np.int32)
        group[rng.choice(n, size=5, replace=False)] = -1
        world = rng.integers(0, 3, size=n, dtype=np.int32)
        world[rng.choice(n, size=4, replace=False)] = -1
        flags = np.full(n, int(ShapeFlags.COLLIDE_SHAPES) | int(ShapeFlags.VISIBLE), dtype=np.int32)
        flags[rng.choice(n, size=n // 3, replace=False)] = int(ShapeFlags.VISIBLE)
    elif name == "per_shape_gap":  # test_per_shape_gap_broad_phase: boxes that only touch through their gaps
        rng = np.random.Generator(np.random.PCG64(7))
        n = 48
        centers, sizes = rng.random((n, 3)) * 3.5, np.full((n, 3), 0.3)
        group = np.where(np.arange(n) % 2 == 0, -1, -2).astype(np.int32)  # negative groups collide with OTHER groups
        world = rng.integers(-1, 2, size=n, dtype=np.int32)
        flags = None
    else:
        raise KeyError(name)
    lower = (centers - sizes).astype(np.float32)
    upper = (centers + sizes).astype(np.float32)
    gap = np.zeros(n, dtype=np.float32)
    if name == "per_shape_gap":
        gap = rng.uniform(0.0, 0.6, size=n).astype(np.float32)
    return lower, upper, gap, group, world, flags


CASES = ["single_world", "multiple_worlds", "shape_flags", "per_shape_gap"]


def _oracle(lib, mode, lower, upper, gap, group, world, flags, filter_pairs=None, cap=None, shape_body=None, body_flags=None,
            include=True, displacement=None, limit=None):
    """displacement: [n, 3] float32 -> the swept entry points (check_aabb_overlap_moving; SAP with the reference's projected
    intervals, `limit` = sort_axis_displacement_limit or None)."""
    index_map, ends = precompute_world_map(world, flags)
    fp = np.ascontiguousarray(filter_pairs if filter_pairs is not None else np.zeros((0, 2)), dtype=np.int32)
    n = lower.shape[0]
    cap = n * (n - 1) // 2 + 1 if cap is None else cap
    out = np.zeros((max(cap, 1), 2), dtype=np.int32)
    ptr = lambda a, t: a.ctypes.data_as(t) if a is not None else None  # noqa: E731
    head = (ptr(lower, _f), ptr(upper, _f), ptr(gap, _f), ptr(group, _i), ptr(world, _i), ptr(index_map, _i), ptr(ends, _i),
            len(ends), max(0, len(ends) - 1), ptr(fp, _i), len(fp), ptr(shape_body, _i), ptr(body_flags, _i), int(include))
    if displacement is None:
        fn = {"nxn": lib.o_broadphase_nxn, "sap": lib.o_broadphase_sap}[mode]
        fn.restype = C.c_int
        count = fn(*head, ptr(out, _i), cap)
    else:
        disp = np.ascontiguousarray(displacement, dtype=np.float32)
        fn = {"nxn": lib.o_broadphase_nxn_swept, "sap": lib.o_broadphase_sap_swept}[mode]
        fn.restype = C.c_int
        extra = (C.c_float(-1.0 if limit is None else limit),) if mode == "sap" else ()
        count = fn(*head, ptr(disp, _f), *extra, ptr(out, _i), cap)
    return count, out[: min(count, cap)]


def overlap_moving(lower, upper, disp, c, i, j):
    """check_aabb_overlap_moving (broad_phase_common.py:41-85) in numpy float32, written independently of the checker: the
    boxes i < j, widened by the combined gap c, must overlap at one common time of [0, 1] while translating by disp."""
    enter, exit_time = np.float32(0.0), np.float32(1.0)
    for axis in range(3):
        lo2, up2 = np.float32(lower[j, axis] - c), np.float32(upper[j, axis] + c)
        delta = np.float32(disp[i, axis] - disp[j, axis])
        if delta == 0.0:
            if lower[i, axis] > up2 or upper[i, axis] < lo2:
                return False
            continue
        a, b = np.float32((lo2 - upper[i, axis]) / delta), np.float32((up2 - lower[i, axis]) / delta)
        if a > b:
            a, b = b, a
        enter, exit_time = max(enter, a), min(exit_time, b)
        if enter > exit_time:
            return False
    return True


def brute_force_swept(lower, upper, gap, group, world, disp, flags=None, filter_pairs=()):
    """brute_force with the swept pair test (what BroadPhaseAllPairs returns with shape_displacement)."""
    n = lower.shape[0]
    filt = {tuple(p) for p in filter_pairs}
    out = set()
    for i in range(n):
        if flags is not None and not flags[i] & int(ShapeFlags.COLLIDE_SHAPES):
            continue
        for j in range(i + 1, n):
            if flags is not None and not flags[j] & int(ShapeFlags.COLLIDE_SHAPES):
                continue
            if world[i] != -1 and world[j] != -1 and world[i] != world[j]:
                continue
            if not _group_pair(int(group[i]), int(group[j])) or (i, j) in filt:
                continue
            if overlap_moving(lower, upper, disp, np.float32(gap[i]) + np.float32(gap[j]), i, j):
                out.add((i, j))
    return out


def test_world_map_layout():
    world = np.array([1, -1, 0, 0, 2, -1, 1, 0], dtype=np.int32)
    index_map, ends = precompute_world_map(world)
    assert index_map.tolist() == [2, 3, 7, 1, 5, 0, 6, 1, 5, 4, 1, 5, 1, 5]
    assert ends.tolist() == [5, 9, 12, 14]
    flags = np.full(8, int(ShapeFlags.COLLIDE_SHAPES), dtype=np.int32)
    flags[[3, 5]] = 0  # visual-only shapes drop out of every segment
    index_map, ends = precompute_world_map(world, flags)
    assert index_map.tolist() == [2, 7, 1, 0, 6, 1, 4, 1, 1] and ends.tolist() == [3, 6, 8, 9]
    index_map, ends = precompute_world_map(np.array([-1, -1], dtype=np.int32))
    assert index_map.tolist() == [0, 1] and ends.tolist() == [2]  # only the dedicated shared segment
    index_map, ends = precompute_world_map(np.zeros(0, dtype=np.int32))
    assert index_map.size == 0 and ends.tolist() == [0]
    with pytest.raises(ValueError):
        precompute_world_map(np.array([0, -2], dtype=np.int32))
    with pytest.raises(ValueError):
        precompute_world_map(np.array([0, 1], dtype=np.int32), np.array([2], dtype=np.int32))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["nxn", "sap"])
def test_oracle_matches_brute_force(oracle_lib, case, mode):
    lower, upper, gap, group, world, flags = make_case(case)
    want = brute_force(lower, upper, gap, group, world, flags)
    assert len(want) > 5
    count, pairs = _oracle(oracle_lib, mode, lower, upper, gap, group, world, flags)
    assert count == len(want) and {tuple(p) for p in pairs} == want
    assert len({tuple(p) for p in pairs}) == count  # shared shapes are not reported once per world


def test_oracle_filters_capacity_and_edge_cases(oracle_lib):
    lower, upper, gap, group, world, flags = make_case("multiple_worlds")
    want = sorted(brute_force(lower, upper, gap, group, world, flags))
    # excluded pairs (sorted, canonical)
    filt = np.array(want[::3], dtype=np.int32)
    count, pairs = _oracle(oracle_lib, "nxn", lower, upper, gap, group, world, flags, filter_pairs=filt)
    assert {tuple(p) for p in pairs} == set(want) - {tuple(p) for p in filt}
    # capacity overflow: the counter keeps counting, writes are clamped (broad_phase_common.py:204-218)
    count, pairs = _oracle(oracle_lib, "nxn", lower, upper, gap, group, world, flags, cap=7)
    assert count == len(want) and len(pairs) == 7 and {tuple(p) for p in pairs} <= set(want)
    # immovable filter: shapes 0..9 static, bodies 0..4 kinematic
    n = lower.shape[0]
    shape_body = np.arange(n, dtype=np.int32) - 10
    body_flags = np.ones(n, dtype=np.int32)
    body_flags[:5] = 2
    immovable = lambda s: shape_body[s] < 0 or body_flags[shape_body[s]] & 2  # noqa: E731
    count, pairs = _oracle(oracle_lib, "sap", lower, upper, gap, group, world, flags, shape_body=shape_body,
                           body_flags=body_flags, include=False)
    assert {tuple(p) for p in pairs} == {p for p in want if not (immovable(p[0]) and immovable(p[1]))}
    # empty, single shape, all groups off, identical boxes
    z3 = np.zeros((0, 3), dtype=np.float32)
    zi = np.zeros(0, dtype=np.int32)
    assert _oracle(oracle_lib, "nxn", z3, z3, np.zeros(0, np.float32), zi, zi, None)[0] == 0
    one = np.zeros((1, 3), dtype=np.float32)
    assert _oracle(oracle_lib, "sap", one, one + 1, np.zeros(1, np.float32), np.ones(1, np.int32), np.zeros(1, np.int32), None)[0] == 0
    same_lo, same_hi = np.zeros((6, 3), dtype=np.float32), np.ones((6, 3), dtype=np.float32)
    g0 = np.zeros(6, dtype=np.int32)
    assert _oracle(oracle_lib, "nxn", same_lo, same_hi, np.zeros(6, np.float32), g0, g0, None)[0] == 0
    assert _oracle(oracle_lib, "sap", same_lo, same_hi, np.zeros(6, np.float32), g0 + 1, g0, None)[0] == 15


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_kernel_logic_on_the_host_matches_oracle(oracle_lib, tmp_path, case, mode):
    """The lane program of broadphase_segment_kernel (shared header), run lane by lane on the CPU."""
    exe = tmp_path / "bp_host_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "broadphase_host_check.cpp")], check=True)
    lower, upper, gap, group, world, flags = make_case(case)
    want = brute_force(lower, upper, gap, group, world, flags)
    filt = np.array(sorted(want)[::4], dtype=np.int32).reshape(-1, 2)
    index_map, ends = precompute_world_map(world, flags)
    m = index_map
    if mode == 1:  # the per-segment sort BroadPhaseSAP performs on the device
        key = lower[index_map, 0] - gap[index_map]
        seg = np.searchsorted(ends, np.arange(len(index_map)), side="right")
        order = np.argsort(key, kind="stable")
        order = order[np.argsort(seg[order], kind="stable")]
        m = index_map[order]
    has_gap = int(case == "per_shape_gap")
    fields = [lower.shape[0], len(filt), len(ends), max(0, len(ends) - 1), len(m), mode, has_gap]
    text = " ".join(map(str, fields)) + "\n" + " ".join(repr(float(x)) for x in lower.ravel()) + "\n" + \
        " ".join(repr(float(x)) for x in upper.ravel()) + "\n"
    if has_gap:
        text += " ".join(repr(float(x)) for x in gap) + "\n"
    text += " ".join(map(str, group)) + "\n" + " ".join(map(str, world)) + "\n" + " ".join(map(str, filt.ravel())) + "\n" + \
        " ".join(map(str, m)) + "\n" + " ".join(map(str, ends)) + "\n"
    res = subprocess.run([str(exe)], input=text, capture_output=True, text=True, check=True).stdout.split()
    count = int(res[0])
    got = {(int(res[1 + 2 * k]), int(res[2 + 2 * k])) for k in range(count)}
    assert len(got) == count
    assert got == want - {tuple(p) for p in filt}


# ------------------------------------------------------------------------------------------------ GPU: the three classes
def _gpu_run(cls_name, lower, upper, gap, group, world, flags, cap=None, filter_pairs=None, **kw):
    import torch

    from newton_amd import geometry

    dev = "cuda:0"
    t = lambda a, d: torch.as_tensor(np.ascontiguousarray(a), dtype=d, device=dev)  # noqa: E731
    n = lower.shape[0]
    cap = n * (n - 1) // 2 + 1 if cap is None else cap
    pairs = torch.full((max(cap, 1), 2), -1, dtype=torch.int32, device=dev)
    count = torch.full((1,), 123, dtype=torch.int32, device=dev)  # launch() must zero it
    bp = getattr(geometry, cls_name)(world, flags, device=dev)
    fp = t(filter_pairs, torch.int32) if filter_pairs is not None else None
    extra = {k: t(v, torch.int32) if isinstance(v, np.ndarray) else v for k, v in kw.items()}
    bp.launch(t(lower, torch.float32), t(upper, torch.float32), t(gap, torch.float32), t(group, torch.int32),
              t(world, torch.int32), n, pairs, count, filter_pairs=fp, **extra)
    c = int(count.cpu().numpy()[0])
    return c, pairs.cpu().numpy()[: min(c, cap)], (pairs, count)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("cls_name", ["BroadPhaseAllPairs", "BroadPhaseSAP"])
def test_hip_broad_phase_matches_brute_force(cls_name, case):
    lower, upper, gap, group, world, flags = make_case(case)
    want = brute_force(lower, upper, gap, group, world, flags)
    count, pairs, _ = _gpu_run(cls_name, lower, upper, gap, group, world, flags)
    assert count == len(want)
    assert {tuple(p) for p in pairs} == want and len({tuple(p) for p in pairs}) == count


@pytest.mark.gpu
def test_hip_broad_phase_filters_capacity_explicit_and_scale():
    import torch

    from newton_amd import geometry

    lower, upper, gap, group, world, flags = make_case("multiple_worlds")
    want = sorted(brute_force(lower, upper, gap, group, world, flags))
    filt = np.array(want[::3], dtype=np.int32)
    for cls_name in ("BroadPhaseAllPairs", "BroadPhaseSAP"):
        count, pairs, _ = _gpu_run(cls_name, lower, upper, gap, group, world, flags, filter_pairs=filt)
        assert {tuple(p) for p in pairs} == set(want) - {tuple(p) for p in filt}
        count, pairs, (dev_pairs, dev_count) = _gpu_run(cls_name, lower, upper, gap, group, world, flags, cap=7)
        assert count == len(want) and {tuple(p) for p in pairs} <= set(want) and len(pairs) == 7
        n = lower.shape[0]
        shape_body = np.arange(n, dtype=np.int32) - 10
        body_flags = np.ones(n, dtype=np.int32)
        body_flags[:5] = 2
        immovable = lambda s: shape_body[s] < 0 or body_flags[shape_body[s]] & 2  # noqa: E731
        count, pairs, _ = _gpu_run(cls_name, lower, upper, gap, group, world, flags, shape_body=shape_body,
                                   body_flags=body_flags, include_static_kinematic_pairs=False)
        assert {tuple(p) for p in pairs} == {p for p in want if not (immovable(p[0]) and immovable(p[1]))}
    # canonical view of the atomically appended list
    count, pairs, (dev_pairs, dev_count) = _gpu_run("BroadPhaseAllPairs", lower, upper, gap, group, world, flags)
    assert geometry.sort_candidate_pairs(dev_pairs, count).cpu().numpy().tolist() == [list(p) for p in want]

    # explicit pair list: AABB test only
    dev = "cuda:0"
    t = lambda a, d: torch.as_tensor(np.ascontiguousarray(a), dtype=d, device=dev)  # noqa: E731
    rng = np.random.default_rng(3)
    n = lower.shape[0]
    listed = np.array(sorted({tuple(sorted(p)) for p in rng.integers(0, n, size=(300, 2)) if p[0] != p[1]}), dtype=np.int32)
    pairs = torch.zeros((len(listed), 2), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    geometry.BroadPhaseExplicit(device=dev).launch(t(lower, torch.float32), t(upper, torch.float32), None, t(listed, torch.int32),
                                                   len(listed), pairs, count)
    c = int(count.cpu().numpy()[0])
    want_e = {tuple(p) for p in listed if np.all(lower[p[0]] <= upper[p[1]]) and np.all(upper[p[0]] >= lower[p[1]])}
    assert c == len(want_e) and {tuple(p) for p in pairs.cpu().numpy()[:c]} == want_e

    # scale: 512 worlds x 40 shapes + 8 shared shapes, NxN and SAP agree with each other and with the oracle
    rng = np.random.default_rng(11)
    W, per, shared = 512, 40, 8
    n = W * per + shared
    centers = rng.random((n, 3)).astype(np.float32) * 4.0
    half = (rng.random((n, 3)).astype(np.float32) * 0.35 + 0.05)
    lower, upper = centers - half, centers + half
    world = np.concatenate([np.repeat(np.arange(W, dtype=np.int32), per), np.full(shared, -1, dtype=np.int32)])
    group = rng.integers(-2, 4, size=n).astype(np.int32)
    gap = rng.uniform(0.0, 0.05, size=n).astype(np.float32)
    cap = 400_000
    res = {}
    for cls_name in ("BroadPhaseAllPairs", "BroadPhaseSAP"):
        count, pairs, _ = _gpu_run(cls_name, lower, upper, gap, group, world, None, cap=cap)
        assert count < cap
        res[cls_name] = {tuple(p) for p in pairs}
        assert len(res[cls_name]) == count
    assert res["BroadPhaseAllPairs"] == res["BroadPhaseSAP"]
    from oracle_bridge import lib

    ocount, opairs = _oracle(lib(), "nxn", lower, upper, gap, group, world, None, cap=cap)
    assert res["BroadPhaseAllPairs"] == {tuple(p) for p in opairs} and ocount == len(opairs)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("entry", ["nt_broadphase_nxn", "nt_broadphase_sap"])
def test_emulated_kernels_match_brute_force(oracle_lib, entry, case):
    """nt_broadphase.hip itself (wave-aggregated append included), executed on the CPU by tests/emu with ballots / shuffles as
    lane rendezvous."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import harness as H

    from newton_amd import _lib as L

    lower, upper, gap, group, world, flags = make_case(case)
    want = brute_force(lower, upper, gap, group, world, flags)
    filt = np.array(sorted(want)[::5], dtype=np.int32).reshape(-1, 2)
    index_map, ends = precompute_world_map(world, flags)
    m = index_map
    if entry.endswith("sap"):
        key = lower[index_map, 0] - gap[index_map]
        seg = np.searchsorted(ends, np.arange(len(index_map)), side="right")
        order = np.argsort(key, kind="stable")
        m = np.ascontiguousarray(index_map[order[np.argsort(seg[order], kind="stable")]])
    v = L.nt_broadphase_in()
    v.lower, v.upper, v.gap = lower.ctypes.data, upper.ctypes.data, gap.ctypes.data
    v.group, v.world = group.ctypes.data, world.ctypes.data
    v.filter_pairs, v.num_filter_pairs, v.include_static_kinematic_pairs = filt.ctypes.data, len(filt), 1
    cap = 5  # smaller than the result: the counter keeps counting
    pairs = np.full((len(want) + 8, 2), -1, dtype=np.int32)
    count = np.zeros(1, dtype=np.int32)
    fn = getattr(H.lib(), entry)
    for c in (len(pairs), cap):
        count[:] = 0
        pairs[:] = -1
        H.check(fn(C.byref(v), m.ctypes.data, ends.ctypes.data, len(ends), max(0, len(ends) - 1), len(m), pairs.ctypes.data,
                   count.ctypes.data, c, None), entry)
        expect = want - {tuple(p) for p in filt}
        assert int(count[0]) == len(expect)
        got = {tuple(p) for p in pairs[: min(c, len(expect))]}
        assert got <= expect and len(got) == min(c, len(expect)) and (c == cap or got == expect)


@pytest.mark.parametrize("case", CASES)
def test_emulated_device_sap_sorts_in_lds_and_matches_brute_force(oracle_lib, case):
    """nt_broadphase_sap_device: projection on the reference's axis + per-segment bitonic sort (LDS) + sweep, all inside the
    library -- no host sort.  The sorted map must be a permutation of every segment ordered by the projected interval start
    (broad_phase_sap.py:44-79,787-811) and the pair set the brute-force one."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import harness as H

    from newton_amd import _lib as L

    lower, upper, gap, group, world, flags = make_case(case)
    want = brute_force(lower, upper, gap, group, world, flags)
    index_map, ends = precompute_world_map(world, flags)
    v = L.nt_broadphase_in()
    v.lower, v.upper, v.gap = lower.ctypes.data, upper.ctypes.data, gap.ctypes.data
    v.group, v.world = group.ctypes.data, world.ctypes.data
    v.filter_pairs, v.num_filter_pairs, v.include_static_kinematic_pairs = None, 0, 1
    n = len(index_map)
    sorted_map = np.full(max(n, 1), -1, dtype=np.int32)
    proj = np.zeros((2, max(n, 1)), dtype=np.float32)
    pairs = np.full((len(want) + 8, 2), -1, dtype=np.int32)
    count = np.zeros(1, dtype=np.int32)
    seg_len = np.diff(np.concatenate([[0], ends])) if len(ends) else np.zeros(0, dtype=np.int64)
    H.check(H.lib().nt_broadphase_sap_device(C.byref(v), index_map.ctypes.data, ends.ctypes.data, len(ends), max(0, len(ends) - 1), n,
                                             int(seg_len.max()) if len(seg_len) else 0, sorted_map.ctypes.data, proj.ctypes.data,
                                             pairs.ctypes.data, count.ctypes.data, len(pairs), None), "nt_broadphase_sap_device")
    assert int(count[0]) == len(want) and {tuple(p) for p in pairs[: len(want)]} == want
    d = np.array([0.5935, 0.7790, 0.1235], dtype=np.float32)
    d = d / np.float32(np.sqrt(np.sum(d * d)))
    begin = 0
    for end in ends:
        seg = sorted_map[begin:end]
        assert sorted(seg.tolist()) == sorted(index_map[begin:end].tolist())
        half = 0.5 * (upper[seg] - lower[seg]) + gap[seg, None]
        lo = (0.5 * (lower[seg] + upper[seg])) @ d - np.abs(d) @ half.T
        assert np.all(np.diff(proj[0, begin:end]) >= 0.0) and np.allclose(proj[0, begin:end], lo, atol=1e-5)
        begin = end


# ------------------------------------------------------------------------------------------------ swept AABBs (speculative mode)
# shape_displacement / sort_axis_displacement_limit of the three launch() methods: check_aabb_overlap_moving
# (broad_phase_common.py:41-85) and _sap_project_aabb (broad_phase_sap.py:44-79).  Pinned by the reference classes executed on the
# stand-in (tests/golden/broadphase_reference_vectors.npz, variants "swept*") and by the reference's own known answers
# (newton/tests/test_broad_phase.py:261-339 test_swept_aabb_requires_simultaneous_overlap, :341-390 length validation).
SWEPT_VARIANTS = ["swept", "swept_capped", "swept_capped_zero", "swept_filtered"]
BP_VEC = os.path.join(ROOT, "tests", "golden", "broadphase_reference_vectors.npz")


def swept_known_answers():
    """(lower, upper, displacement or None, sort limit or None, expected pair count, classes) of the reference test."""
    h = np.float32(0.1)
    box = lambda *c: (np.array(c, np.float32) - h, np.array(c, np.float32) + h)  # noqa: E731
    mk = lambda a, b: (np.stack([a[0], b[0]]), np.stack([a[1], b[1]]))  # noqa: E731
    every = ("BroadPhaseAllPairs", "BroadPhaseSAP", "BroadPhaseExplicit")
    lo, up = mk(box(0, 0, 0), box(0.5, -0.5, 0))
    same = np.array([[1, 1, 0], [1, 1, 0]], np.float32)
    union_lo = lo.copy()
    union_up = np.stack([np.array([1, 1, 0], np.float32) + h, np.array([1.5, 0.5, 0], np.float32) + h])
    col_lo, col_up = mk(box(0, 0, 0), box(1, 1, 0))
    col_d = np.array([[1, 1, 0], [0, 0, 0]], np.float32)
    fast_lo, fast_up = mk(box(0, 0, 0), box(0.4, 0, 0))
    fast_d = np.array([[2.0, 0, 0], [1.7, 0, 0]], np.float32)
    return [
        ("swept unions overlap (static test on the unions)", union_lo, union_up, None, None, 1, every),
        ("same motion: never at the same place at the same time", lo, up, same, None, 0, every),
        ("one box runs into the other", col_lo, col_up, col_d, None, 1, every),
        ("fast pair with a capped sort-axis extension", fast_lo, fast_up, fast_d, 0.25, 1, ("BroadPhaseSAP",)),
    ]


def _emu_swept(kind, v, disp, limit, cap=None):
    """One launch of the emulated kernels (tests/emu): kind in nxn / sap / explicit; v = a variant dict of broadphase_cases."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import harness as H

    from newton_amd import _lib as L

    lower, upper, gap = (np.ascontiguousarray(v[k], np.float32) for k in ("lower", "upper", "gap"))
    n = lower.shape[0]
    view = L.nt_broadphase_in()
    view.lower, view.upper, view.gap = lower.ctypes.data, upper.ctypes.data, gap.ctypes.data
    view.include_static_kinematic_pairs = int(v["include"])
    keep = []
    for name in ("shape_body", "body_flags"):
        if v[name] is not None:
            keep.append(np.ascontiguousarray(v[name], np.int32))
            setattr(view, name, keep[-1].ctypes.data)
    motion = L.nt_broadphase_motion()
    disp = np.ascontiguousarray(disp, np.float32)
    motion.displacement, motion.sort_axis_displacement_limit = disp.ctypes.data, -1.0 if limit is None else float(limit)
    cap = n * (n - 1) // 2 + 1 if cap is None else cap
    pairs = np.full((cap, 2), -1, dtype=np.int32)
    count = np.zeros(1, dtype=np.int32)
    if kind == "explicit":
        ep = np.ascontiguousarray(v["explicit_pairs"], np.int32)
        H.check(H.lib().nt_broadphase_explicit_swept(C.byref(view), C.byref(motion), ep.ctypes.data, len(ep), pairs.ctypes.data,
                                                     count.ctypes.data, cap, None), "nt_broadphase_explicit_swept")
        return int(count[0]), pairs[: min(cap, int(count[0]))]
    group, world = np.ascontiguousarray(v["group"], np.int32), np.ascontiguousarray(v["world"], np.int32)
    view.group, view.world = group.ctypes.data, world.ctypes.data
    if v["filter_pairs"] is not None and len(v["filter_pairs"]):
        keep.append(np.ascontiguousarray(v["filter_pairs"], np.int32))
        view.filter_pairs, view.num_filter_pairs = keep[-1].ctypes.data, len(keep[-1])
    index_map, ends = precompute_world_map(world, v["flags"])
    m = len(index_map)
    if kind == "nxn":
        H.check(H.lib().nt_broadphase_nxn_swept(C.byref(view), C.byref(motion), index_map.ctypes.data, ends.ctypes.data, len(ends),
                                                max(0, len(ends) - 1), m, pairs.ctypes.data, count.ctypes.data, cap, None),
                "nt_broadphase_nxn_swept")
    else:
        sorted_map = np.full(max(m, 1), -1, dtype=np.int32)
        proj = np.zeros((2, max(m, 1)), dtype=np.float32)
        seg_len = np.diff(np.concatenate([[0], ends])) if len(ends) else np.zeros(0, dtype=np.int64)
        H.check(H.lib().nt_broadphase_sap_device_swept(
            C.byref(view), C.byref(motion), index_map.ctypes.data, ends.ctypes.data, len(ends), max(0, len(ends) - 1), m,
            int(seg_len.max()) if len(seg_len) else 0, sorted_map.ctypes.data, proj.ctypes.data, pairs.ctypes.data, count.ctypes.data,
            cap, None), "nt_broadphase_sap_device_swept")
    return int(count[0]), pairs[: min(cap, int(count[0]))]


@pytest.mark.parametrize("case", CASES)
def test_oracle_swept_matches_independent_numpy(oracle_lib, case):
    """The checker's swept N x N against a numpy slab test written without looking at it; a zero displacement array gives the
    static result; the uncapped SAP finds the same set as N x N on these boxes (every true pair has overlapping projections)."""
    import broadphase_cases as bc

    v = bc.variants(case)["swept"]
    args = (v["lower"], v["upper"], v["gap"], v["group"], v["world"], v["flags"])
    want = brute_force_swept(*args[:5], v["displacement"], v["flags"])
    static = brute_force(*args)
    assert len(want) > 5 and want != static
    count, pairs = _oracle(oracle_lib, "nxn", *args, displacement=v["displacement"])
    assert count == len(want) and {tuple(p) for p in pairs} == want
    count, pairs = _oracle(oracle_lib, "sap", *args, displacement=v["displacement"])
    assert {tuple(p) for p in pairs} == want
    count, pairs = _oracle(oracle_lib, "nxn", *args, displacement=np.zeros_like(v["displacement"]))
    assert {tuple(p) for p in pairs} == static


@pytest.mark.parametrize("variant", SWEPT_VARIANTS)
@pytest.mark.parametrize("case", CASES)
def test_emulated_swept_kernels_reproduce_the_reference_classes(oracle_lib, case, variant):
    """nt_broadphase_{nxn,sap_device,explicit}_swept (csrc/nt_broadphase.hip on the CPU emulator) against the pair lists the
    reference classes produced with shape_displacement (and sort_axis_displacement_limit for SAP), as sets; capacity overflow
    keeps counting."""
    import broadphase_cases as bc

    ref = np.load(BP_VEC)
    v = bc.variants(case)[variant]
    for kind in ("nxn", "sap", "explicit"):
        want = {tuple(p) for p in ref[f"{case}/{variant}/{kind}_pairs"]}
        count, pairs = _emu_swept(kind, v, v["displacement"], v["limit"])
        assert count == len(want) and {tuple(p) for p in pairs} == want, (kind, count, len(want))
    count, pairs = _emu_swept("nxn", v, v["displacement"], v["limit"], cap=3)
    assert count == len(ref[f"{case}/{variant}/nxn_pairs"]) and len(pairs) == 3


def test_emulated_swept_known_answers(oracle_lib):
    """newton/tests/test_broad_phase.py:261-339 on the emulated kernels."""
    base = dict(gap=np.zeros(2, np.float32), group=np.ones(2, np.int32), world=np.zeros(2, np.int32), flags=None, filter_pairs=None,
                shape_body=None, body_flags=None, include=True, explicit_pairs=np.array([[0, 1]], np.int32))
    for what, lo, up, disp, limit, expect, classes in swept_known_answers():
        if disp is None:
            continue  # the static entry points: covered above
        v = dict(base, lower=lo, upper=up)
        for cls, kind in (("BroadPhaseAllPairs", "nxn"), ("BroadPhaseSAP", "sap"), ("BroadPhaseExplicit", "explicit")):
            if cls in classes:
                assert _emu_swept(kind, v, disp, limit)[0] == expect, (what, kind)


def _gpu_swept(cls_name, v, disp, limit=None, cap=None):
    import torch

    from newton_amd import geometry

    dev = "cuda:0"
    t = lambda a, d: torch.as_tensor(np.ascontiguousarray(a), dtype=d, device=dev) if a is not None else None  # noqa: E731
    n = v["lower"].shape[0]
    cap = n * (n - 1) // 2 + 1 if cap is None else cap
    pairs = torch.full((max(cap, 1), 2), -1, dtype=torch.int32, device=dev)
    count = torch.full((1,), 77, dtype=torch.int32, device=dev)
    kw = dict(shape_body=t(v["shape_body"], torch.int32), body_flags=t(v["body_flags"], torch.int32),
              include_static_kinematic_pairs=v["include"], shape_displacement=t(disp, torch.float32))
    lo, up, gap = t(v["lower"], torch.float32), t(v["upper"], torch.float32), t(v["gap"], torch.float32)
    if cls_name == "BroadPhaseExplicit":
        ep = t(v["explicit_pairs"], torch.int32)
        geometry.BroadPhaseExplicit(device=dev).launch(lo, up, gap, ep, int(ep.shape[0]), pairs, count, **kw)
    else:
        if cls_name == "BroadPhaseSAP":
            kw["sort_axis_displacement_limit"] = limit
        bp = getattr(geometry, cls_name)(v["world"], v["flags"], device=dev)
        bp.launch(lo, up, gap, t(v["group"], torch.int32), t(v["world"], torch.int32), n, pairs, count,
                  filter_pairs=t(v["filter_pairs"], torch.int32), **kw)
    c = int(count.cpu().numpy()[0])
    return c, pairs.cpu().numpy()[: min(c, cap)]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", SWEPT_VARIANTS)
@pytest.mark.parametrize("case", CASES)
def test_hip_swept_broad_phases_reproduce_the_reference_classes(case, variant):
    """The three classes with shape_displacement on the MI355X against the executed reference (sets) and the checker."""
    import broadphase_cases as bc
    from oracle_bridge import lib

    ref = np.load(BP_VEC)
    v = bc.variants(case)[variant]
    for cls_name, kind in (("BroadPhaseAllPairs", "nxn"), ("BroadPhaseSAP", "sap"), ("BroadPhaseExplicit", "explicit")):
        want = {tuple(p) for p in ref[f"{case}/{variant}/{kind}_pairs"]}
        count, pairs = _gpu_swept(cls_name, v, v["displacement"], v["limit"])
        assert count == len(want) and {tuple(p) for p in pairs} == want, (cls_name, count, len(want))
    ocount, opairs = _oracle(lib(), "sap", v["lower"], v["upper"], v["gap"], v["group"], v["world"], v["flags"],
                             filter_pairs=v["filter_pairs"], shape_body=v["shape_body"], body_flags=v["body_flags"],
                             include=v["include"], displacement=v["displacement"], limit=v["limit"])
    assert {tuple(p) for p in opairs} == {tuple(p) for p in ref[f"{case}/{variant}/sap_pairs"]}


@pytest.mark.gpu
def test_hip_swept_known_answers_validation_and_scale():
    """The reference's known answers (test_broad_phase.py:261-339), its argument validation (:341-390, broad_phase_sap.py:727-735)
    and 256 worlds x 40 moving shapes: N x N = uncapped SAP = checker."""
    from oracle_bridge import lib

    base = dict(gap=np.zeros(2, np.float32), group=np.ones(2, np.int32), world=np.zeros(2, np.int32), flags=None, filter_pairs=None,
                shape_body=None, body_flags=None, include=True, explicit_pairs=np.array([[0, 1]], np.int32))
    for what, lo, up, disp, limit, expect, classes in swept_known_answers():
        for cls_name in classes:
            assert _gpu_swept(cls_name, dict(base, lower=lo, upper=up), disp, limit)[0] == expect, (what, cls_name)
    v = dict(base, lower=np.zeros((2, 3), np.float32), upper=np.ones((2, 3), np.float32))
    for cls_name in ("BroadPhaseAllPairs", "BroadPhaseSAP", "BroadPhaseExplicit"):
        with pytest.raises(ValueError, match="shape_displacement length must match"):
            _gpu_swept(cls_name, v, np.zeros((1, 3), np.float32))
    for bad in (-0.1, float("nan"), float("inf")):
        with pytest.raises(ValueError, match="sort_axis_displacement_limit must be a non-negative finite number"):
            _gpu_swept("BroadPhaseSAP", v, np.zeros((2, 3), np.float32), limit=bad)
    rng = np.random.default_rng(21)
    W, per, shared = 256, 40, 6
    n = W * per + shared
    centers = rng.random((n, 3)).astype(np.float32) * 4.0
    half = rng.random((n, 3)).astype(np.float32) * 0.3 + 0.05
    scene = dict(base, lower=centers - half, upper=centers + half, gap=rng.uniform(0.0, 0.05, size=n).astype(np.float32),
                 group=rng.integers(-2, 4, size=n).astype(np.int32),
                 world=np.concatenate([np.repeat(np.arange(W, dtype=np.int32), per), np.full(shared, -1, dtype=np.int32)]))
    disp = (rng.standard_normal((n, 3)) * 0.4).astype(np.float32)
    disp[rng.random(n) < 0.3] = 0.0
    cap = 400_000
    got = {}
    for cls_name in ("BroadPhaseAllPairs", "BroadPhaseSAP"):
        count, pairs = _gpu_swept(cls_name, scene, disp, cap=cap)
        assert count < cap and len({tuple(p) for p in pairs}) == count
        got[cls_name] = {tuple(p) for p in pairs}
    ocount, opairs = _oracle(lib(), "nxn", scene["lower"], scene["upper"], scene["gap"], scene["group"], scene["world"], None, cap=cap,
                             displacement=disp)
    static = _oracle(lib(), "nxn", scene["lower"], scene["upper"], scene["gap"], scene["group"], scene["world"], None, cap=cap)[0]
    assert ocount != static
    assert got["BroadPhaseAllPairs"] == {tuple(p) for p in opairs} == got["BroadPhaseSAP"]
