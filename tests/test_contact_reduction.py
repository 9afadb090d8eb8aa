"""Global contact reduction: the float32 checker oracle/oracle_reduce.py against the record of the
reference's own reducer (tests/golden/reduce_reference_vectors.npz: export_and_reduce_contact_centered_two_spatial_depths +
export_reduced_contacts_kernel executed on the stand-in, two arrival orders), and the building blocks against the reference's
closed forms.  The device twin lives in tests/test_gpu_sdf.py."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))


@pytest.mark.parametrize("name", ["patch", "two_pairs_many_normals", "duplicates_and_ties", "outer_only", "single"])
def test_checker_keeps_the_contacts_the_reference_reducer_keeps(name):
    import oracle_reduce as orr
    import reduce_cases as rc

    ref = np.load(os.path.join(HERE, "golden", "reduce_reference_vectors.npz"))
    c = rc.pack(rc.contacts(name))
    out = orr.reduce_contacts(c)
    assert len(out["fp"]) == len(ref[f"{name}/fp"]) and len(out["fp"]) <= len(c["fp"])
    assert np.array_equal(out["pair"], ref[f"{name}/pair"]) and np.array_equal(out["fp"], ref[f"{name}/fp"])
    assert np.array_equal(out["pos"], ref[f"{name}/pos"]) and np.array_equal(out["depth"], ref[f"{name}/depth"])
    assert np.array_equal(out["normal"], ref[f"{name}/normal"])  # the octahedral round trip, bit for bit
    if name == "duplicates_and_ties":  # some of the planted twins must have been suppressed, never both of a pair
        assert len(out["fp"]) < len(set(map(int, c["fp"])))
    if name == "outer_only":  # no depth slot, no voxel slot: at most 6 survivors per normal bin
        assert len(out["fp"]) <= 6 * 20


def test_building_blocks():
    import oracle_reduce as orr

    rng = np.random.default_rng(0)
    for _ in range(200):  # the pruned icosahedron scan returns the face with the largest dot product
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        b = orr.get_slot(n)
        dots = orr.FACE_NORMALS.astype(np.float64) @ n.astype(np.float64)
        assert dots[b] >= dots.max() - 1e-6
        back = orr.decode_oct(orr.encode_oct(n))
        assert np.abs(back - n).max() < 5e-7
    for b in range(20):  # orthonormal face frames
        u, v = orr.FACE_FRAMES[b]
        fn = orr.FACE_NORMALS[b]
        assert abs(u @ v) < 1e-6 and abs(u @ fn) < 1e-6 and abs(v @ fn) < 1e-6 and abs(np.linalg.norm(v) - 1) < 1e-6
    assert orr.float_flip(np.float32(-1.0)) < orr.float_flip(np.float32(-0.5)) < orr.float_flip(np.float32(0.0)) \
        < orr.float_flip(np.float32(1e-30)) < orr.float_flip(np.float32(2.0))
    # inner contacts outrank outer ones whatever the score
    assert orr.value_spatial(np.float32(-5.0), True, 0) > orr.value_spatial(np.float32(5.0), False, orr.FINGERPRINT_MASK)
    assert orr.voxel_index(np.array([9, 9, 9], np.float32), np.zeros(3, np.float32), np.ones(3, np.float32), (4, 5, 5)) == 99
    assert orr.voxel_index(np.array([-1, 0.5, 0.0], np.float32), np.zeros(3, np.float32), np.ones(3, np.float32), (4, 5, 5)) == 8


# ------------------------------------------------------------------------------------------------ the gfx950 kernels, emulated
sys.path.insert(0, os.path.join(HERE, "emu"))


@pytest.fixture(scope="module")
def emu():
    import harness

    return harness.lib()


def run_reduce_list(lib, c, stream=None, to_dev=None, to_host=None):
    """nt_contacts_reduce_list on a packed case (host arrays for the emulated library; the device twin passes converters).
    -> (indices into the list, exported normals), segments in shape-pair order."""
    import ctypes as C

    from newton_amd import _lib as L

    to_dev = to_dev or (lambda a: a)
    to_host = to_host or (lambda a: a)
    ptr = (lambda a: a.ctypes.data) if stream is None else (lambda a: a.data_ptr())
    order = np.lexsort((c["pair"][:, 1], c["pair"][:, 0]))  # group by shape pair (stable: the arrival order inside survives)
    g = {k: np.ascontiguousarray(v[order]) for k, v in c.items()}
    change = np.flatnonzero(np.any(np.diff(g["pair"], axis=0) != 0, axis=1)) + 1
    seg = np.concatenate([[0], change, [len(order)]]).astype(np.int32)
    n = len(order)
    keep = {k: to_dev(v) for k, v in g.items()}
    keep["seg"] = to_dev(seg)
    keep["count"], keep["index"] = to_dev(np.zeros(1, np.int32)), to_dev(np.full(n + 1, -1, np.int32))
    keep["onormal"] = to_dev(np.zeros((n + 1, 3), np.float32))
    a = L.nt_contact_reduce_list()
    a.segment_start, a.segments = ptr(keep["seg"]), len(seg) - 1
    for f in ("pos", "normal", "depth", "fp", "centered", "inner", "outer", "local", "aabb_lo", "aabb_hi", "res"):
        setattr(a, f, ptr(keep[f]))
    a.out_count, a.out_index, a.out_normal, a.capacity = ptr(keep["count"]), ptr(keep["index"]), ptr(keep["onormal"]), n + 1
    assert lib.nt_contacts_reduce_list(C.byref(a), stream) == 0
    cnt = int(to_host(keep["count"])[0])
    idx, nrm = to_host(keep["index"])[:cnt], to_host(keep["onormal"])[:cnt]
    return order[idx], nrm, g["pair"][idx], g["fp"][idx]


@pytest.mark.parametrize("name", ["patch", "two_pairs_many_normals", "duplicates_and_ties", "outer_only", "single"])
def test_emulated_reduction_kernel_keeps_what_the_reference_reducer_keeps(emu, name):
    """contacts_reduce_list_kernel (LDS table, 64-bit maxima, twin suppression, rank by fingerprint) against the reference's
    record: same survivors, same exported normals, bit for bit.  Workgroups finish in any order, so rows are sorted by key."""
    import reduce_cases as rc

    ref = np.load(os.path.join(HERE, "golden", "reduce_reference_vectors.npz"))
    c = rc.pack(rc.contacts(name))
    idx, nrm, pair, fp = run_reduce_list(emu, c)
    o = np.lexsort((fp, pair[:, 1], pair[:, 0]))
    assert np.array_equal(pair[o], ref[f"{name}/pair"]) and np.array_equal(fp[o], ref[f"{name}/fp"])
    assert np.array_equal(c["pos"][idx[o]], ref[f"{name}/pos"]) and np.array_equal(c["depth"][idx[o]], ref[f"{name}/depth"])
    assert np.array_equal(nrm[o], ref[f"{name}/normal"])
    # inside a segment the kernel already writes ascending fingerprints
    for p in {tuple(x) for x in pair}:
        rows = [k for k in range(len(fp)) if tuple(pair[k]) == p]
        assert rows == list(range(rows[0], rows[-1] + 1)) and np.all(np.diff(fp[rows]) > 0)
