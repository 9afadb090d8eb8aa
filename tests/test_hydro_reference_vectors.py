"""The hydroelastic pipeline against tests/golden/hydro_reference_vectors.npz -- the record of the
REFERENCE's own kernels (sdf_hydroelastic.py: SAT broad phase, four octree levels with ordered scatter, generate, decode) executed
by tests/golden/make_hydro_reference_vectors.py.  Bit for bit: the float32 checker (oracle/oracle_hydro.py hydro_pipeline), the
kernel source compiled for the host (tests/emu, nt_hydro_pairs) and -- under -m gpu -- the MI355X."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import hydro_cases  # noqa: E402

from newton_amd.mc_tables import tables  # noqa: E402

REF = np.load(os.path.join(ROOT, "tests", "golden", "hydro_reference_vectors.npz"))
SCENES = hydro_cases.scenes()


def _tables():
    tr, fl = tables()
    return np.ascontiguousarray(tr, np.int32), np.ascontiguousarray(np.asarray(fl).reshape(-1, 2), np.uint8)


def _check_rows(name, rows, vox_counts):
    """rows: (pair_idx, pair-local fingerprint, shape_a, shape_b, centre, normal, separation, stiffness) in (pair, voxel, face) order
    == the reference's ContactData (whose fingerprints number the voxels across all pairs)."""
    rr = REF[f"{name}/rows"]
    assert len(rows) == len(rr) > 50
    base = np.concatenate([[0], np.cumsum(vox_counts)])
    for r, q in zip(rows, rr):
        assert (int(r[2]), int(r[3]), int(r[1]) + 5 * int(base[int(r[0])])) == (int(q[0]), int(q[1]), int(q[2]))
        assert np.array_equal(np.asarray(r[4], np.float32), q[3:6]) and np.array_equal(np.asarray(r[5], np.float32), q[6:9])
        assert np.float32(r[6]) == q[9] and np.float32(r[7]) == q[10]


@pytest.mark.parametrize("name", sorted(SCENES))
def test_checker_pipeline_is_the_executed_reference(name):
    import oracle_hydro as H

    s = SCENES[name]
    rows, vox = H.hydro_pipeline(s["pairs"], s["X"], s["data"], s["gap"], s["kh"], s["sdfs"], _tables())
    allv = [(x, y, z, p) for p, v in enumerate(vox) for (x, y, z) in v]
    assert [tuple(int(c) for c in r) for r in REF[f"{name}/voxels"]] == allv  # the octree's survivors, in the ordered-scatter order
    _check_rows(name, rows, [len(v) for v in vox])


REDUCED = {"prune_nm": dict(pre_prune=True, normal_matching=True), "full_nm": dict(pre_prune=False, normal_matching=True),
           "prune_plain": dict(pre_prune=True, normal_matching=False),
           "prune_anchor": dict(pre_prune=True, normal_matching=True, anchor_contact=True),
           "prune_moment": dict(pre_prune=True, normal_matching=True, moment_matching=True),
           "full_moment_plain": dict(pre_prune=False, normal_matching=False, moment_matching=True)}


def _check_reduced_rows(name, tag, rows, vox_counts, tol=0.0):
    """rows: (pair_idx, pair-local fingerprint, shape_a, shape_b, centre, normal, separation, stiffness, friction scale) in export
    order == the record of HydroelasticContactReduction.reduce / export executed in thread order."""
    rr = REF[f"{name}/reduced_{tag}/rows"]
    assert len(rows) == len(rr) > 10
    base = np.concatenate([[0], np.cumsum(vox_counts)])
    for r, q in zip(rows, rr):
        assert (int(r[2]), int(r[3]), int(r[1]) + 5 * int(base[int(r[0])])) == (int(q[0]), int(q[1]), int(q[2]))
        got = np.concatenate([np.asarray(r[4], np.float32), np.asarray(r[5], np.float32), [np.float32(r[6]), np.float32(r[7])]])
        want = np.concatenate([q[3:6], q[6:9], [q[9], q[10]]])
        if tol == 0.0:
            assert np.array_equal(got, want), (got, want)
        else:
            assert np.all(np.abs(got - want) <= tol * np.maximum(1.0, np.abs(want))), (got, want)
        assert np.float32(r[8]) == q[12]


@pytest.mark.parametrize("tag", sorted(REDUCED))
@pytest.mark.parametrize("name", sorted(SCENES))
def test_checker_reduced_pipeline_is_the_executed_reference(name, tag):
    """reduce_contacts=True: aggregates + local-first pruning in the generate kernel, HydroelasticContactReduction.reduce / export
    (contact_reduction_hydroelastic.py) executed on the stand-in; the checker's rows equal the record bit for bit."""
    import oracle_hydro as H

    s = SCENES[name]
    rows, vox = H.hydro_pipeline(s["pairs"], s["X"], s["data"], s["gap"], s["kh"], s["sdfs"], _tables(),
                                 reduce=dict(aabb_lo=s["aabb_lo"], aabb_hi=s["aabb_hi"], res=s["res"], **REDUCED[tag]))
    _check_reduced_rows(name, tag, rows, [len(v) for v in vox])


def run_hydro_pairs(lib, s, make_sdf, ptr, cap=4096, reduce=None):
    """nt_hydro_pairs on one world holding the scene's pairs; `make_sdf(t)` -> (nt_sdf, keepalive), `ptr(array)` -> device pointer of
    a host array (identity for the emulated library).  Returns (rows like the checker's, per-pair voxel counts are not exposed:
    the fingerprints carry the ranks)."""
    from newton_amd import _lib as L

    n = len(s["pairs"])
    descs = [make_sdf(t) for t in s["sdfs"]]
    table = (L.nt_sdf * len(descs))(*[d for d, _ in descs])
    tr, fl = _tables()
    bufs = dict(pairs=np.ascontiguousarray(s["pairs"], np.int32), prefix=np.array([0, n], np.int32), kind=np.ones(n, np.uint8),
                X=np.ascontiguousarray(s["X"], np.float32), data=np.ascontiguousarray(s["data"], np.float32),
                gap=np.ascontiguousarray(s["gap"], np.float32), kh=np.ascontiguousarray(s["kh"], np.float32),
                idx=np.arange(len(s["sdfs"]), dtype=np.int32), tr=tr, fl=fl, count=np.zeros(1, np.int32), o_pair=np.full(cap, -1, np.int32),
                o_key=np.zeros(cap, np.int32), o_data=np.zeros((cap, 9), np.float32), o_rank=np.zeros(cap, np.int32),
                o_stiff=np.zeros(cap, np.float32), blk=np.zeros((n, 2), np.int32), norm=np.zeros((n, 2), np.int32),
                table=np.frombuffer(bytes(table), np.uint8).copy())
    if reduce is not None:
        bufs.update(lo=np.ascontiguousarray(s["aabb_lo"], np.float32), hi=np.ascontiguousarray(s["aabb_hi"], np.float32),
                    res=np.ascontiguousarray(s["res"], np.int32), f_count=np.zeros(2, np.int32), f_rec=np.zeros((cap, 12), np.float32),
                    o_fric=np.zeros(cap, np.float32))
    dev = {k: ptr(v) for k, v in bufs.items()}
    a = L.nt_hydro_args()
    a.pairs, a.pair_count = dev["pairs"][0], n
    a.shape_transform, a.shape_data, a.shape_gap, a.shape_kh = dev["X"][0], dev["data"][0], dev["gap"][0], dev["kh"][0]
    a.shape_sdf_index, a.sdf_table, a.sdf_count = dev["idx"][0], dev["table"][0], len(descs)
    a.tri_range, a.flat_edge_verts, a.margin_contact_area, a.edge_clamp_min = dev["tr"][0], dev["fl"][0], 1.0e-2, 0.02
    a.out_count, a.out_pair, a.out_key, a.out_data, a.capacity = dev["count"][0], dev["o_pair"][0], dev["o_key"][0], dev["o_data"][0], cap
    a.pair_world_prefix, a.worlds, a.pairs_per_world, a.pair_kind = dev["prefix"][0], 1, n, dev["kind"][0]
    a.out_pairs_normalized, a.out_blk, a.out_rank, a.out_stiffness = dev["norm"][0], dev["blk"][0], dev["o_rank"][0], dev["o_stiff"][0]
    if reduce is not None:
        a.reduce = (1 | (2 if reduce["pre_prune"] else 0) | (4 if reduce["normal_matching"] else 0) |
                    (8 if reduce.get("anchor_contact") else 0) | (16 if reduce.get("moment_matching") else 0))
        a.shape_aabb_lower, a.shape_aabb_upper, a.shape_voxel_res = dev["lo"][0], dev["hi"][0], dev["res"][0]
        a.face_count, a.face_rec, a.face_capacity, a.out_friction = dev["f_count"][0], dev["f_rec"][0], cap, dev["o_fric"][0]
    assert lib.nt_hydro_pairs(C.byref(a), None) == 0
    out = {k: v[1]() for k, v in dev.items() if k in ("count", "o_pair", "o_key", "o_data", "o_rank", "o_stiff", "blk", "norm", "o_fric",
                                                      "f_count")}
    m = int(out["count"][0])
    assert m <= cap and np.array_equal(out["blk"][:, 1], np.bincount(out["o_pair"][:m], minlength=n))
    order = np.lexsort((out["o_rank"][:m], out["o_pair"][:m]))
    rows = [(out["o_pair"][i], out["o_key"][i], out["norm"][out["o_pair"][i]][0], out["norm"][out["o_pair"][i]][1], out["o_data"][i, 0:3],
             out["o_data"][i, 3:6], out["o_data"][i, 6], out["o_stiff"][i]) + ((out["o_fric"][i],) if reduce is not None else ())
            for i in order]
    if reduce is not None:
        assert out["f_count"][1] == 0 and 0 < out["f_count"][0] <= cap
    for p in range(n):  # ranks of a pair are 0 .. count-1
        r = np.sort(out["o_rank"][:m][out["o_pair"][:m] == p])
        assert np.array_equal(r, np.arange(len(r)))
    del descs
    return rows, out["norm"]


@pytest.fixture(scope="module")
def emu():
    import build

    return C.CDLL(build.build())


@pytest.mark.parametrize("name", sorted(SCENES))
def test_emulated_hydro_pairs_kernel_is_the_executed_reference(emu, name):
    from test_sdf_contact import _emu_sdf

    s = SCENES[name]
    rows, norm = run_hydro_pairs(emu, s, lambda t: _emu_sdf(emu, t), lambda v: (C.c_void_p(v.ctypes.data), lambda v=v: v))
    assert np.array_equal(norm, REF[f"{name}/normalized"])
    vox_counts = np.bincount(REF[f"{name}/voxels"][:, 3], minlength=len(s["pairs"]))
    _check_rows(name, rows, vox_counts)


@pytest.mark.parametrize("tag", sorted(REDUCED))
@pytest.mark.parametrize("name", sorted(SCENES))
def test_emulated_reduced_hydro_pairs_kernel_is_the_executed_reference(emu, name, tag):
    """nt_hydro_pairs with reduce_contacts: positions / depths / ids / order exact; stiffness and matched normals to 2e-6 (the
    rotation goes through acos / sin / cos of the host's libm here, numpy's in the record)."""
    from test_sdf_contact import _emu_sdf

    s = SCENES[name]
    rows, norm = run_hydro_pairs(emu, s, lambda t: _emu_sdf(emu, t), lambda v: (C.c_void_p(v.ctypes.data), lambda v=v: v),
                                 reduce=REDUCED[tag])
    assert np.array_equal(norm, REF[f"{name}/normalized"])
    _check_reduced_rows(name, tag, rows, np.bincount(REF[f"{name}/voxels"][:, 3], minlength=len(s["pairs"])), tol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENES))
def test_device_hydro_pairs_kernel_is_the_executed_reference(name):
    import torch

    from newton_amd import _lib
    from newton_amd.sdf_device import DeviceSDF

    s = SCENES[name]
    keep = []

    def make(t):
        d = DeviceSDF(t)
        keep.append(d)
        return d.desc, d

    def ptr(v):
        t = torch.from_numpy(v.view(np.int16) if v.dtype == np.uint16 else v).cuda()
        keep.append(t)
        return t.data_ptr(), (lambda t=t, v=v: t.cpu().numpy().view(v.dtype).reshape(v.shape))

    rows, norm = run_hydro_pairs(_lib.load(), s, make, ptr)
    torch.cuda.synchronize()
    assert np.array_equal(norm, REF[f"{name}/normalized"])
    _check_rows(name, rows, np.bincount(REF[f"{name}/voxels"][:, 3], minlength=len(s["pairs"])))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(REDUCED))
@pytest.mark.parametrize("name", sorted(SCENES))
def test_device_reduced_hydro_pairs_kernel_is_the_executed_reference(name, tag):
    import torch

    from newton_amd import _lib
    from newton_amd.sdf_device import DeviceSDF

    s = SCENES[name]
    keep = []

    def make(t):
        d = DeviceSDF(t)
        keep.append(d)
        return d.desc, d

    def ptr(v):
        t = torch.from_numpy(v.view(np.int16) if v.dtype == np.uint16 else v).cuda()
        keep.append(t)
        return t.data_ptr(), (lambda t=t, v=v: t.cpu().numpy().view(v.dtype).reshape(v.shape))

    rows, norm = run_hydro_pairs(_lib.load(), s, make, ptr, reduce=REDUCED[tag])
    torch.cuda.synchronize()
    assert np.array_equal(norm, REF[f"{name}/normalized"])
    _check_reduced_rows(name, tag, rows, np.bincount(REF[f"{name}/voxels"][:, 3], minlength=len(s["pairs"])), tol=2e-6)
