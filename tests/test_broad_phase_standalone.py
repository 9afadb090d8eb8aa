"""Standalone broad phases on raw AABB arrays (newton.geometry.BroadPhaseAllPairs / BroadPhaseSAP / BroadPhaseExplicit),
restated from newton/tests/test_broad_phase.py:78-145,399-1465,2272-2330: random boxes with exclusive / shared collision
groups, several worlds plus shared (-1) shapes, visual-only shapes filtered by flags, per-shape gaps, excluded pairs, the
immovable-pair filter and capacity overflow -- the candidate SET must equal a numpy brute force exactly.

CPU: the oracle (N x N and sort-and-sweep), the host-side world map, and the kernels' own pair logic compiled for the host
(tools/broadphase_host_check.cpp shares csrc/nt_broadphase_core.hpp with the HIP kernels).  GPU: the three classes."""
import os
import subprocess
from math import sqrt

import ctypes as C
import numpy as np
import pytest

from newton_amd.enums import ShapeFlags
from newton_amd.geometry import precompute_world_map

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
            include=True):
    index_map, ends = precompute_world_map(world, flags)
    fp = np.ascontiguousarray(filter_pairs if filter_pairs is not None else np.zeros((0, 2)), dtype=np.int32)
    n = lower.shape[0]
    cap = n * (n - 1) // 2 + 1 if cap is None else cap
    out = np.zeros((max(cap, 1), 2), dtype=np.int32)
    fn = {"nxn": lib.o_broadphase_nxn, "sap": lib.o_broadphase_sap}[mode]
    fn.restype = C.c_int
    ptr = lambda a, t: a.ctypes.data_as(t) if a is not None else None  # noqa: E731
    count = fn(ptr(lower, _f), ptr(upper, _f), ptr(gap, _f), ptr(group, _i), ptr(world, _i), ptr(index_map, _i), ptr(ends, _i),
               len(ends), max(0, len(ends) - 1), ptr(fp, _i), len(fp), ptr(shape_body, _i), ptr(body_flags, _i), int(include),
               ptr(out, _i), cap)
    return count, out[: min(count, cap)]


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
