"""MESH vs infinite plane (SURVEY.md section 8 row a24's mesh legs; newton/_src/geometry/narrow_phase.py:618-631 routing,
:1744-1992 one lane per mesh vertex) with the reducer variant that leg uses (contact_reduction_global.py:1246-1346
reduce_contact_in_hashtable, fed by write_contact_to_reducer :2059-2096).

CPU: the checker (oracle/oracle_mesh_plane.py, oracle_reduce.reduce_buffered_contacts) against the record of the reference's own
kernels (tests/golden/mesh_plane_reference_vectors.npz, tests/golden/make_mesh_plane_reference_vectors.py), and the HIP kernel
source on the emulator against both.  GPU: nt_mesh_plane_pairs on the device against the record."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
VEC = os.path.join(HERE, "golden", "mesh_plane_reference_vectors.npz")

import mesh_plane_cases as mc  # noqa: E402


@pytest.mark.parametrize("name", mc.CASES)
def test_checker_buffers_the_contacts_the_reference_kernel_buffers(name):
    """Per (mesh, plane) pair, in vertex order: which vertices become contacts, their centre, distance and the octahedral code of
    the normal -- bit for bit."""
    import oracle_mesh_plane as omp
    import oracle_reduce as orr

    ref = np.load(VEC)
    c = omp.buffered_contacts(mc.scene(name))
    assert np.array_equal(c["pair"], ref[f"{name}/buffered_pair"]) and np.array_equal(c["fp"], ref[f"{name}/buffered_fp"])
    assert np.array_equal(c["pos"], ref[f"{name}/buffered_pos"]) and np.array_equal(c["depth"], ref[f"{name}/buffered_depth"])
    oct_ = np.array([orr.encode_oct(n) for n in c["normal"]], np.float32).reshape(-1, 2)
    assert np.array_equal(oct_, ref[f"{name}/buffered_oct"])
    if name != "separated":
        assert len(c["fp"]) > 20


@pytest.mark.parametrize("name", mc.CASES)
def test_checker_keeps_the_contacts_the_reference_reducer_keeps(name):
    """reduce_contact_in_hashtable + export_reduced_contacts_kernel executed in two arrival orders vs reduce_buffered_contacts:
    the same survivors, bit-identical geometry, the shape margins and the gap sum the writer receives."""
    import oracle_mesh_plane as omp

    ref = np.load(VEC)
    s = mc.scene(name)
    out = omp.mesh_plane_rows(s)
    assert np.array_equal(out["pair"], ref[f"{name}/pair"]) and np.array_equal(out["fp"], ref[f"{name}/fp"])
    assert np.array_equal(out["pos"], ref[f"{name}/pos"]) and np.array_equal(out["depth"], ref[f"{name}/depth"])
    assert np.array_equal(out["normal"], ref[f"{name}/normal"])
    misc = ref[f"{name}/misc"]  # margin a, margin b, radius a, radius b, gap sum
    assert np.array_equal(out["margin_a"], misc[:, 0]) and np.array_equal(out["margin_b"], misc[:, 1])
    assert np.all(misc[:, 2:4] == 0.0)
    if len(misc):
        gaps = np.array([np.float32(s["shape_gap"][a]) + np.float32(s["shape_gap"][b]) for a, b in out["pair"]], np.float32)
        assert np.array_equal(gaps, misc[:, 4])
        assert len(out["fp"]) < len(ref[f"{name}/buffered_fp"])  # the reduction drops contacts on every case with contacts


def test_the_buffered_variant_is_not_the_centred_one():
    """Same list through both reducers: the buffered variant gates the directional slots by depth < 1e-4 |aabb| and lets every
    contact compete for the depth / voxel slots, so the survivors differ (this is why the SDF leg's reducer cannot serve it)."""
    import oracle_mesh_plane as omp
    import oracle_reduce as orr

    differs = []
    for name in mc.CASES[:4]:
        c = omp.buffered_contacts(mc.scene(name))
        n = len(c["fp"])
        centred = dict(c, centered=c["pos"], inner=np.full(n, 1e9, np.float32), outer=np.full(n, 1e9, np.float32),
                       local=np.array([omp.transform_point(omp.transform_inverse(c["xform_a"][i]), c["pos"][i]) for i in range(n)],
                                      np.float32))
        a, b = orr.reduce_buffered_contacts(c), orr.reduce_contacts(centred)
        differs.append(set(zip(a["pair"][:, 0].tolist(), a["fp"].tolist())) != set(zip(b["pair"][:, 0].tolist(), b["fp"].tolist())))
    assert any(differs), differs


# ------------------------------------------------------------------------------------------------ the HIP kernel (nt_mesh_plane.hip)
def run_mesh_plane(lib, s, reduce=1, capacity=None, start=0, stream=None, to_dev=None, to_host=None, world_regions=False):
    """nt_mesh_plane_pairs over a scene of mesh_plane_cases (host arrays for the emulated library; the device twin passes
    converters).  The pairs go in as the candidate lists hold them -- (smaller id, larger id) -- and must come back as (mesh, plane).
    world_regions: the same pairs laid out as ONE world's region of 2 * len(pairs) slots with interleaved foreign pairs (kind 0),
    the layout CollisionPipeline's SDF leg hands over.  -> (pairs after the call, blk, rows dict in row order, total rows)."""
    from newton_amd import _lib as L
    from newton_amd.enums import GeoType

    to_dev = to_dev or (lambda x: x)
    to_host = to_host or (lambda x: x)
    S = len(s["shape_gap"])
    pairs = np.sort(np.asarray(s["pairs"], np.int32), axis=1)
    P = len(pairs)
    keep = {}
    a = L.nt_mesh_plane_args()

    def put(name, arr):
        x = keep[name] = to_dev(np.ascontiguousarray(arr))
        return x.data_ptr() if hasattr(x, "data_ptr") else x.ctypes.data

    if world_regions:  # slot 2k: a foreign pair of kind 0, slot 2k + 1: pair k (kind 2); prefix [0, 2P]
        region = np.zeros((2 * P, 2), np.int32)
        region[1::2] = pairs
        kind = np.zeros(2 * P, np.uint8)
        kind[1::2] = 2
        a.pairs, a.pair_kind = put("pairs", region), put("kind", kind)
        a.pair_world_prefix, a.worlds, a.pairs_per_world = put("prefix", np.array([0, 2 * P], np.int32)), 1, 2 * P
        slots = 2 * P
    else:
        a.pairs, a.pair_count = put("pairs", pairs if P else np.zeros((1, 2), np.int32)), P
        slots = max(P, 1)
    stype = np.where(s["vertex_count"] > 0, int(GeoType.MESH), int(GeoType.PLANE)).astype(np.int32)
    a.shape_type, a.shape_transform, a.shape_data, a.shape_gap = (put("type", stype), put("xf", s["shape_transform"]),
                                                                   put("data", s["shape_data"]), put("gap", s["shape_gap"]))
    a.shape_vertex_range = put("vr", np.stack([s["vertex_start"], s["vertex_count"]], axis=1).astype(np.int32))
    a.vertices = put("verts", s["vertices"] if len(s["vertices"]) else np.zeros((1, 3), np.float32))
    a.shape_aabb_lower, a.shape_aabb_upper, a.shape_voxel_res = put("lo", s["aabb_lo"]), put("hi", s["aabb_hi"]), put("res", s["res"])
    a.reduce = int(reduce)
    capacity = int(s["vertex_count"].sum()) + 8 + start if capacity is None else capacity
    a.out_count = put("count", np.array([start], np.int32))
    a.out_pair, a.out_key = put("opair", np.full(capacity, -1, np.int32)), put("okey", np.full(capacity, -1, np.int32))
    a.out_data, a.capacity = put("odata", np.zeros((capacity, 9), np.float32)), capacity
    a.out_blk = put("blk", np.full((slots, 2), -7, np.int32))
    rc = lib.nt_mesh_plane_pairs(C.byref(a), stream)
    assert rc == 0, rc
    h = {k: np.asarray(to_host(v)) for k, v in keep.items()}
    total = int(h["count"][0])
    rows = dict(pair=h["opair"], key=h["okey"], data=h["odata"])
    return h["pairs"], h["blk"], rows, total


def check_against_record(name, out_pairs, blk, rows, total, ref, s, slot_of=lambda k: k, start=0):
    """Every pair's block = the reference's exported contacts of that (mesh, plane) pair, in ascending vertex order, bit for bit."""
    want_pairs = np.asarray(s["pairs"], np.int32)
    n_ref = 0
    for k, (mesh, plane) in enumerate(want_pairs):
        slot = slot_of(k)
        assert tuple(out_pairs[slot]) == (mesh, plane)  # normalised to (mesh, plane) whatever the id order
        sel = np.flatnonzero((ref[f"{name}/pair"][:, 0] == mesh) & (ref[f"{name}/pair"][:, 1] == plane))
        r0, cnt = int(blk[slot][0]), int(blk[slot][1])
        assert cnt == len(sel), (name, k, cnt, len(sel))
        n_ref += cnt
        if cnt == 0:
            continue
        assert r0 >= start
        sl = slice(r0, r0 + cnt)
        assert np.all(rows["pair"][sl] == slot)
        assert np.array_equal(rows["key"][sl], ref[f"{name}/fp"][sel])
        d = rows["data"][sl]
        assert np.array_equal(d[:, 0:3], ref[f"{name}/pos"][sel]) and np.array_equal(d[:, 3:6], ref[f"{name}/normal"][sel])
        assert np.array_equal(d[:, 6], ref[f"{name}/depth"][sel])
        assert np.array_equal(d[:, 7], ref[f"{name}/misc"][sel, 0]) and np.array_equal(d[:, 8], ref[f"{name}/misc"][sel, 1])
    assert total - start == n_ref


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import harness as H

    return H.lib()


@pytest.mark.parametrize("name", mc.CASES)
def test_emulated_kernel_reproduces_the_reference_leg(emu, name):
    """mesh_plane_pairs_kernel on the CPU emulator (LDS table, 64-bit maxima, winners recomputed from the vertex index, twin
    suppression, rank by vertex) against the executed reference: plain pair list and the per-world candidate-region layout with
    foreign pairs in between, rows appended behind a non-zero counter."""
    ref = np.load(VEC)
    s = mc.scene(name)
    check_against_record(name, *run_mesh_plane(emu, s), ref, s)
    out_pairs, blk, rows, total = run_mesh_plane(emu, s, world_regions=True, start=5)
    check_against_record(name, out_pairs, blk, rows, total, ref, s, slot_of=lambda k: 2 * k + 1, start=5)
    assert np.all(blk[0::2] == -7) and np.all(out_pairs[0::2] == 0)  # foreign pairs untouched


def test_emulated_kernel_unreduced_and_overflow(emu):
    """reduce = 0: every admitted vertex is a row, ascending vertex order = the reference's buffered list (the order its unreduced
    kernel appends on one thread); a row buffer that is too small keeps counting and clamps the block."""
    ref = np.load(VEC)
    name = "two_meshes_margins"
    s = mc.scene(name)
    out_pairs, blk, rows, total = run_mesh_plane(emu, s, reduce=0)
    n = len(ref[f"{name}/buffered_fp"])
    assert total == n
    for k, (mesh, plane) in enumerate(s["pairs"]):
        sel = np.flatnonzero(ref[f"{name}/buffered_pair"][:, 0] == mesh)
        r0, cnt = blk[k]
        assert cnt == len(sel) and np.array_equal(rows["key"][r0:r0 + cnt], ref[f"{name}/buffered_fp"][sel])
        assert np.array_equal(rows["data"][r0:r0 + cnt, 0:3], ref[f"{name}/buffered_pos"][sel])
        assert np.array_equal(rows["data"][r0:r0 + cnt, 6], ref[f"{name}/buffered_depth"][sel])
    full = len(ref[f"{name}/fp"])
    out_pairs, blk, rows, total = run_mesh_plane(emu, s, capacity=full - 3)
    assert total == full and int(blk[:, 1].sum()) == full - 3 and np.all(blk[:, 0] + blk[:, 1] <= full - 3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", mc.CASES)
def test_hip_mesh_plane_reproduces_the_reference_leg(name):
    """nt_mesh_plane_pairs on the MI355X against the executed reference, bit for bit; twice on the same buffers -> identical."""
    import torch

    from newton_amd import _lib

    lib = _lib.load()
    dev = lambda x: torch.from_numpy(x).to("cuda:0")  # noqa: E731
    host = lambda x: x.cpu().numpy()  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = np.load(VEC)
    s = mc.scene(name)
    first = run_mesh_plane(lib, s, stream=stream, to_dev=dev, to_host=host)
    check_against_record(name, *first, ref, s)
    out_pairs, blk, rows, total = run_mesh_plane(lib, s, stream=stream, to_dev=dev, to_host=host, world_regions=True, start=5)
    check_against_record(name, out_pairs, blk, rows, total, ref, s, slot_of=lambda k: 2 * k + 1, start=5)
    again = run_mesh_plane(lib, s, stream=stream, to_dev=dev, to_host=host)
    for k, (mesh, plane) in enumerate(s["pairs"]):  # the block's position may differ between runs, its rows may not
        (a0, n0), (a1, n1) = first[1][k], again[1][k]
        assert n0 == n1 and np.array_equal(first[2]["data"][a0:a0 + n0], again[2]["data"][a1:a1 + n1])


@pytest.mark.gpu
def test_hip_mesh_plane_many_worlds_unreduced_and_overflow():
    """512 replicated worlds of the two-mesh scene through the candidate-region layout: every world's blocks equal the record;
    reduce = 0 equals the buffered list; a short row buffer keeps counting and clamps."""
    import torch

    from newton_amd import _lib
    from newton_amd.enums import GeoType

    lib = _lib.load()
    ref = np.load(VEC)
    name = "two_meshes_margins"
    s = mc.scene(name)
    dev = lambda x: torch.from_numpy(x).to("cuda:0")  # noqa: E731
    host = lambda x: x.cpu().numpy()  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out_pairs, blk, rows, total = run_mesh_plane(lib, s, reduce=0, stream=stream, to_dev=dev, to_host=host)
    assert total == len(ref[f"{name}/buffered_fp"])
    for k, (mesh, plane) in enumerate(s["pairs"]):
        sel = np.flatnonzero(ref[f"{name}/buffered_pair"][:, 0] == mesh)
        r0, cnt = blk[k]
        assert cnt == len(sel) and np.array_equal(rows["key"][r0:r0 + cnt], ref[f"{name}/buffered_fp"][sel])
        assert np.array_equal(rows["data"][r0:r0 + cnt, 0:3], ref[f"{name}/buffered_pos"][sel])
    full = len(ref[f"{name}/fp"])
    out_pairs, blk, rows, total = run_mesh_plane(lib, s, capacity=full - 3, stream=stream, to_dev=dev, to_host=host)
    assert total == full and int(blk[:, 1].sum()) == full - 3
    # many worlds: the scene's shapes replicated W times (shape ids offset per world), pairs in per-world regions of 4 slots
    W, S, PPW = 512, len(s["shape_gap"]), 4
    rep = {k: np.concatenate([s[k]] * W) for k in ("shape_transform", "shape_data", "shape_gap", "aabb_lo", "aabb_hi", "res")}
    vr = np.tile(np.stack([s["vertex_start"], s["vertex_count"]], axis=1).astype(np.int32), (W, 1))  # one shared vertex table
    region = np.zeros((W * PPW, 2), np.int32)
    kind = np.zeros(W * PPW, np.uint8)
    for w in range(W):
        for k, p in enumerate(np.sort(s["pairs"], axis=1)):
            region[w * PPW + k] = p + w * S
            kind[w * PPW + k] = 2
    stype = np.tile(np.where(s["vertex_count"] > 0, int(GeoType.MESH), int(GeoType.PLANE)).astype(np.int32), W)
    cap = W * full + 16
    t = {k: dev(np.ascontiguousarray(v)) for k, v in dict(
        pairs=region, kind=kind, prefix=(np.arange(W + 1) * len(s["pairs"])).astype(np.int32), type=stype, xf=rep["shape_transform"],
        data=rep["shape_data"], gap=rep["shape_gap"], vr=vr, verts=s["vertices"], lo=rep["aabb_lo"], hi=rep["aabb_hi"], res=rep["res"],
        count=np.zeros(1, np.int32), opair=np.full(cap, -1, np.int32), okey=np.full(cap, -1, np.int32),
        odata=np.zeros((cap, 9), np.float32), blk=np.full((W * PPW, 2), -7, np.int32)).items()}
    a = _lib.nt_mesh_plane_args()
    a.pairs, a.pair_kind, a.pair_world_prefix, a.worlds, a.pairs_per_world = (t["pairs"].data_ptr(), t["kind"].data_ptr(),
                                                                               t["prefix"].data_ptr(), W, PPW)
    a.shape_type, a.shape_transform, a.shape_data, a.shape_gap = (t["type"].data_ptr(), t["xf"].data_ptr(), t["data"].data_ptr(),
                                                                   t["gap"].data_ptr())
    a.shape_vertex_range, a.vertices = t["vr"].data_ptr(), t["verts"].data_ptr()
    a.shape_aabb_lower, a.shape_aabb_upper, a.shape_voxel_res = t["lo"].data_ptr(), t["hi"].data_ptr(), t["res"].data_ptr()
    a.reduce, a.out_count, a.out_pair, a.out_key, a.out_data, a.capacity, a.out_blk = (
        1, t["count"].data_ptr(), t["opair"].data_ptr(), t["okey"].data_ptr(), t["odata"].data_ptr(), cap, t["blk"].data_ptr())
    assert lib.nt_mesh_plane_pairs(C.byref(a), stream) == 0
    blk_h, data_h, key_h = host(t["blk"]), host(t["odata"]), host(t["okey"])
    assert int(host(t["count"])[0]) == W * full
    for w in (0, 1, 255, 511):
        for k, (mesh, plane) in enumerate(s["pairs"]):
            sel = np.flatnonzero(ref[f"{name}/pair"][:, 0] == mesh)
            r0, cnt = blk_h[w * PPW + k]
            assert cnt == len(sel) and np.array_equal(key_h[r0:r0 + cnt], ref[f"{name}/fp"][sel])
            assert np.array_equal(data_h[r0:r0 + cnt, 0:3], ref[f"{name}/pos"][sel])
            assert np.array_equal(data_h[r0:r0 + cnt, 6], ref[f"{name}/depth"][sel])
    assert np.all(blk_h.reshape(W, PPW, 2)[:, 2:] == -7)  # slots past a world's live pairs stay untouched


@pytest.mark.gpu
def test_hip_mesh_plane_python_entry():
    """newton_amd.mesh_plane.mesh_plane_contacts (the stand-alone host entry) = the record; host tensors are refused."""
    import torch

    from newton_amd.enums import GeoType
    from newton_amd.mesh_plane import mesh_plane_contacts

    ref = np.load(VEC)
    name = "cube_flat"
    s = mc.scene(name)
    d = lambda x, t: torch.as_tensor(np.ascontiguousarray(x), dtype=t, device="cuda:0")  # noqa: E731
    stype = np.where(s["vertex_count"] > 0, int(GeoType.MESH), int(GeoType.PLANE))
    args = [d(np.sort(s["pairs"], axis=1), torch.int32), d(stype, torch.int32), d(s["shape_transform"], torch.float32),
            d(s["shape_data"], torch.float32), d(s["shape_gap"], torch.float32),
            d(np.stack([s["vertex_start"], s["vertex_count"]], axis=1), torch.int32), d(s["vertices"], torch.float32),
            d(s["aabb_lo"], torch.float32), d(s["aabb_hi"], torch.float32), d(s["res"], torch.int32)]
    out = mesh_plane_contacts(*args)
    n = int(out["count"].item())
    assert n == len(ref[f"{name}/fp"]) and args[0].cpu().numpy().tolist() == s["pairs"].tolist()
    r0, cnt = out["blk"].cpu().numpy()[0]
    assert cnt == n and np.array_equal(out["vertex"].cpu().numpy()[r0:r0 + cnt], ref[f"{name}/fp"])
    assert np.array_equal(out["data"].cpu().numpy()[r0:r0 + cnt, 0:3], ref[f"{name}/pos"])
    from newton_amd import _lib

    if os.path.abspath(_lib.LIB_PATH) == os.path.abspath(_lib._DEFAULT_LIB):  # (the CPU emulator's "cuda" tensors are host tensors)
        with pytest.raises(TypeError):
            mesh_plane_contacts(args[0].cpu(), *args[1:])
