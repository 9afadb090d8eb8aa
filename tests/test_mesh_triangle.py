"""MESH vs convex primitive through the triangle midphase (SURVEY.md section 8 rows a19 / a20 / a24 on triangle meshes without the
SDF route; newton/_src/geometry/narrow_phase.py:633-638 routing, :1455-1568 midphase, collision_core.py:996-1276,
contact_reduction_global.py:2299-2403 GJK / MPR + manifold per triangle into the global reducer).

CPU: the checker (oracle/oracle_mesh_triangle.py -> liboracle.so o_mesh_triangle_contacts, oracle_reduce.reduce_buffered_contacts)
against the record of the reference's own kernels (tests/golden/mesh_triangle_reference_vectors.npz,
tests/golden/make_mesh_triangle_reference_vectors.py), and the HIP kernel source on the emulator against the record.
GPU: nt_mesh_triangle_pairs on the device against the record (tests/test_gpu_mesh_triangle.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
VEC = os.path.join(HERE, "golden", "mesh_triangle_reference_vectors.npz")

import mesh_triangle_cases as mc  # noqa: E402


@pytest.mark.parametrize("name", mc.CASES + mc.HF_CASES)
def test_checker_finds_the_triangles_the_reference_midphase_finds(name):
    """(mesh, convex, triangle) triples: the support-function AABB in the unscaled mesh frame, widened by margin + gap, against
    every triangle's bounds, minus the back faces -- the same set."""
    import oracle_mesh_triangle as om

    ref = np.load(VEC)
    triples, _ = om.triangle_contacts(mc.scene(name))
    assert np.array_equal(triples.reshape(-1, 3), ref[f"{name}/tri_pairs"].reshape(-1, 3))
    assert len(triples) > 0


@pytest.mark.parametrize("name", mc.CASES + mc.HF_CASES)
def test_checker_buffers_the_contacts_the_reference_kernel_buffers(name):
    """Per (mesh, convex) pair in fingerprint order: which triangles produce which manifold contacts, centre, distance and the
    octahedral code of the normal -- bit for bit (MPR / GJK with the TRIANGLE support map and Minkowski seed, build_manifold)."""
    import oracle_mesh_triangle as om
    import oracle_reduce as orr

    ref = np.load(VEC)
    _, c = om.triangle_contacts(mc.scene(name))
    assert np.array_equal(c["pair"], ref[f"{name}/buffered_pair"]) and np.array_equal(c["fp"], ref[f"{name}/buffered_fp"])
    assert np.array_equal(c["pos"], ref[f"{name}/buffered_pos"]) and np.array_equal(c["depth"], ref[f"{name}/buffered_depth"])
    oct_ = np.array([orr.encode_oct(n) for n in c["normal"]], np.float32).reshape(-1, 2)
    assert np.array_equal(oct_, ref[f"{name}/buffered_oct"])
    assert len(c["fp"]) >= 8
    tri = c["fp"] >> 4
    assert np.all((c["fp"] >> 3) & 1 == 1)  # sort_sub_key = (triangle << 1) | 1
    if name in ("box_on_grid", "cylinder_and_cone", "mirrored_mesh"):
        assert np.any(np.bincount(tri) >= 3)  # face manifolds: several contacts per triangle


@pytest.mark.parametrize("name", mc.CASES + mc.HF_CASES)
def test_checker_keeps_the_contacts_the_reference_reducer_keeps(name):
    """reduce_contact_in_hashtable + export_reduced_contacts_kernel executed in two arrival orders vs reduce_buffered_contacts:
    the same survivors, bit-identical geometry, the margins / effective radii / gap sum the writer receives."""
    import oracle_mesh_triangle as om

    ref = np.load(VEC)
    s = mc.scene(name)
    out = om.mesh_triangle_rows(s)
    assert np.array_equal(out["pair"], ref[f"{name}/pair"]) and np.array_equal(out["fp"], ref[f"{name}/fp"])
    assert np.array_equal(out["pos"], ref[f"{name}/pos"]) and np.array_equal(out["depth"], ref[f"{name}/depth"])
    assert np.array_equal(out["normal"], ref[f"{name}/normal"])
    misc = ref[f"{name}/misc"]  # margin a, margin b, radius a, radius b, gap sum
    assert np.array_equal(out["margin_a"], misc[:, 0]) and np.array_equal(out["margin_b"], misc[:, 1])
    assert np.array_equal(out["radius_a"], misc[:, 2]) and np.array_equal(out["radius_b"], misc[:, 3])
    gaps = np.array([np.float32(s["shape_gap"][a]) + np.float32(s["shape_gap"][b]) for a, b in out["pair"]], np.float32)
    assert np.array_equal(gaps, misc[:, 4])
    assert len(out["fp"]) < len(ref[f"{name}/buffered_fp"])  # the reduction drops contacts on every case


# ------------------------------------------------------------------------------------------------ the HIP kernel (nt_mesh_triangle.hip)
def run_mesh_triangle(lib, s, reduce=1, capacity=None, start=0, stream=None, to_dev=None, to_host=None, world_regions=False,
                      blocks=False):
    """nt_mesh_triangle_pairs over a scene of mesh_triangle_cases (host arrays for the emulated library; the device twin passes
    converters).  The pairs go in as the candidate lists hold them -- (smaller id, larger id) -- and must come back as (mesh, convex).
    world_regions: the same pairs laid out as ONE world's region of 2 * len(pairs) slots with interleaved foreign pairs (kind 0),
    the layout CollisionPipeline's SDF leg hands over.  -> (pairs after the call, blk, rows dict, total rows counted)."""
    from newton_amd import _lib as L

    to_dev = to_dev or (lambda x: x)
    to_host = to_host or (lambda x: x)
    pairs = np.sort(np.asarray(s["pairs"], np.int32), axis=1)
    P = len(pairs)
    keep = {}
    a = L.nt_mesh_triangle_args()

    def put(name, arr):
        x = keep[name] = to_dev(np.ascontiguousarray(arr))
        return x.data_ptr() if hasattr(x, "data_ptr") else x.ctypes.data

    if world_regions:  # slot 2k: a foreign pair of kind 0, slot 2k + 1: pair k (kind 3); prefix [0, 2P]
        region = np.zeros((2 * P, 2), np.int32)
        region[1::2] = pairs
        kind = np.zeros(2 * P, np.uint8)
        kind[1::2] = 3
        a.pairs, a.pair_kind = put("pairs", region), put("kind", kind)
        a.pair_world_prefix, a.worlds, a.pairs_per_world = put("prefix", np.array([0, 2 * P], np.int32)), 1, 2 * P
        slots = 2 * P
    else:
        a.pairs, a.pair_count = put("pairs", pairs if P else np.zeros((1, 2), np.int32)), P
        slots = max(P, 1)
    a.shape_type, a.shape_transform, a.shape_data, a.shape_gap = (put("type", s["shape_type"]), put("xf", s["shape_transform"]),
                                                                   put("data", s["shape_data"]), put("gap", s["shape_gap"]))
    a.shape_vertex_range = put("vr", np.stack([s["vertex_start"], s["vertex_count"]], axis=1).astype(np.int32))
    a.shape_triangle_range = put("tr", np.stack([s["tri_start"], s["tri_count"]], axis=1).astype(np.int32))
    a.vertices = put("verts", s["vertices"] if len(s["vertices"]) else np.zeros((1, 3), np.float32))
    a.indices = put("idx", s["indices"] if len(s["indices"]) else np.zeros((1, 3), np.int32))
    a.shape_aabb_lower, a.shape_aabb_upper, a.shape_voxel_res = put("lo", s["aabb_lo"]), put("hi", s["aabb_hi"]), put("res", s["res"])
    a.reduce = int(reduce)
    n_tri = int(s["tri_count"].sum()) + int(sum(2 * (int(r) - 1) * (int(c) - 1) for _, r, c, *_ in s["hf_table"]))
    capacity = 5 * n_tri * max(P, 1) + 8 + start if capacity is None else capacity
    a.out_count = put("count", np.array([start], np.int32))
    a.out_pair, a.out_key = put("opair", np.full(capacity, -1, np.int32)), put("okey", np.full(capacity, -1, np.int32))
    a.out_data, a.capacity = put("odata", np.zeros((capacity, 9), np.float32)), capacity
    a.out_radius = put("oradius", np.full((capacity, 2), -1.0, np.float32))
    a.out_blk = put("blk", np.full((slots, 2), -7, np.int32))
    if len(s["hf_table"]) > 0:  # heightfields: HeightfieldData records + the concatenated elevation grids
        hf = np.zeros(len(s["hf_table"]), dtype=[("data_offset", "<i4"), ("nrow", "<i4"), ("ncol", "<i4"), ("hx", "<f4"), ("hy", "<f4"),
                                                 ("min_z", "<f4"), ("max_z", "<f4")])
        for k, row in enumerate(s["hf_table"]):
            hf[k] = (int(row[0]), int(row[1]), int(row[2]), row[3], row[4], row[5], row[6])
        a.shape_heightfield_index, a.elevations = put("hfi", s["hf_index"]), put("hfe", s["hf_elev"])
        a.heightfields = put("hfd", hf.view(np.uint8).reshape(len(hf), 28))
    if int(s["hull_count"].sum()) > 0:  # CONVEX_MESH partners: their vertex tables
        a.hull_points = put("hp", s["hull_points"])
        a.shape_hull_range = put("hr", np.stack([s["hull_start"], s["hull_count"]], axis=1).astype(np.int32))
    if blocks and int(s["tri_count"].sum()) > 0:  # the bounds of every 64 consecutive triangles: the scan skips blocks that miss the query box (same candidates)
        from newton_amd.mesh import triangle_block_bounds

        tabs, bstart, nb = [], np.zeros(len(s["shape_gap"]), np.int32), 0
        for k in range(len(s["shape_gap"])):
            if s["tri_count"][k] > 0:
                v0, nv, t0, nt = int(s["vertex_start"][k]), int(s["vertex_count"][k]), int(s["tri_start"][k]), int(s["tri_count"][k])
                tabs.append(triangle_block_bounds(s["vertices"][v0:v0 + nv], s["indices"][t0:t0 + nt]))
                bstart[k] = nb
                nb += len(tabs[-1])
        a.block_bounds, a.shape_block_start = put("bb", np.concatenate(tabs)), put("bs", bstart)
    rc = lib.nt_mesh_triangle_pairs(C.byref(a), stream)
    assert rc == 0, rc
    h = {k: np.asarray(to_host(v)) for k, v in keep.items()}
    rows = dict(pair=h["opair"], key=h["okey"], data=h["odata"], radius=h["oradius"])
    return h["pairs"], h["blk"], rows, int(h["count"][0])


def check_against_record(name, out_pairs, blk, rows, total, ref, s, slot_of=lambda k: k, start=0, reduce=1):
    """Every pair's block = the reference's contacts of that (mesh, convex) pair -- the exported survivors under the reduction, the
    whole buffer without it -- in ascending fingerprint order, bit for bit, with the margins and effective radii of the export."""
    from oracle_mesh_triangle import effective_radius

    pre = "" if reduce else "buffered_"
    n_ref = 0
    for k, (sa, sb) in enumerate(np.asarray(s["pairs"], np.int32)):
        mesh, convex = (sa, sb) if s["shape_type"][sa] in (mc.MESH, mc.HFIELD) else (sb, sa)
        slot = slot_of(k)
        assert tuple(out_pairs[slot]) == (mesh, convex)  # normalised to (mesh, convex) whatever the id order
        rp = ref[f"{name}/{pre}pair"]
        sel = np.flatnonzero((rp[:, 0] == mesh) & (rp[:, 1] == convex))
        r0, cnt = int(blk[slot][0]), int(blk[slot][1])
        assert cnt == len(sel), (name, k, cnt, len(sel))
        n_ref += cnt
        if cnt == 0:
            continue
        assert r0 >= start
        sl = slice(r0, r0 + cnt)
        assert np.all(rows["pair"][sl] == slot)
        assert np.array_equal(rows["key"][sl], ref[f"{name}/{pre}fp"][sel])
        d = rows["data"][sl]
        assert np.array_equal(d[:, 0:3], ref[f"{name}/{pre}pos"][sel]) and np.array_equal(d[:, 6], ref[f"{name}/{pre}depth"][sel])
        if reduce:  # the export decodes the buffered octahedral code
            assert np.array_equal(d[:, 3:6], ref[f"{name}/normal"][sel])
        else:       # the generated normal itself: its code is what the record's buffer holds
            import oracle_reduce as orr

            oct_ = np.array([orr.encode_oct(n) for n in d[:, 3:6]], np.float32).reshape(-1, 2)
            assert np.array_equal(oct_, ref[f"{name}/buffered_oct"][sel])
        assert np.all(d[:, 7] == s["shape_data"][mesh][3]) and np.all(d[:, 8] == s["shape_data"][convex][3])
        assert np.all(rows["radius"][sl, 0] == 0.0)
        assert np.all(rows["radius"][sl, 1] == effective_radius(s["shape_type"][convex], s["shape_data"][convex]))
    assert total - start == n_ref
    return n_ref


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import harness as H

    return H.lib()


@pytest.mark.parametrize("reduce", [1, 0])
@pytest.mark.parametrize("name", mc.CASES + mc.HF_CASES)
def test_emulated_kernel_matches_the_record(emu, name, reduce):
    """The HIP source of nt_mesh_triangle_pairs compiled for the host (tests/emu) against the executed reference: the triangle scan,
    batches of waiting triangles, MPR / GJK + manifold per lane, the LDS reduction table with winners recomputed from (triangle,
    manifold index); without the reduction every generated contact in fingerprint order.  Plain pair list and the per-world
    candidate-region layout with foreign pairs in between, rows appended behind a non-zero counter."""
    ref = np.load(VEC)
    s = mc.scene(name)
    out = run_mesh_triangle(emu, s, reduce=reduce)
    assert check_against_record(name, *out, ref, s, reduce=reduce) > 0
    out = run_mesh_triangle(emu, s, reduce=reduce, start=5, world_regions=True, blocks=True)  # ... and the block bounds in front
    check_against_record(name, *out, ref, s, slot_of=lambda k: 2 * k + 1, start=5, reduce=reduce)
    assert np.all(out[1][0::2] == -7) and np.all(out[0][0::2] == 0)  # foreign pairs untouched


def test_emulated_kernel_counts_past_the_capacity(emu):
    """Rows beyond the capacity are counted, the block is clamped, nothing is written past the arrays."""
    s = mc.scene("box_on_grid")
    _, blk, rows, total = run_mesh_triangle(emu, s, reduce=0, capacity=10)
    assert total == 52 and int(blk[0][1]) == 10 and np.all(rows["pair"] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("reduce", [1, 0])
@pytest.mark.parametrize("name", mc.CASES + mc.HF_CASES)
def test_hip_mesh_triangle_reproduces_the_reference_leg(name, reduce):
    """nt_mesh_triangle_pairs on the MI355X against the executed reference, bit for bit; twice on the same buffers -> identical."""
    import torch

    from newton_amd import _lib

    lib = _lib.load()
    dev = lambda x: torch.from_numpy(x).to("cuda:0")  # noqa: E731
    host = lambda x: x.cpu().numpy()  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = np.load(VEC)
    s = mc.scene(name)
    first = run_mesh_triangle(lib, s, reduce=reduce, stream=stream, to_dev=dev, to_host=host)
    assert check_against_record(name, *first, ref, s, reduce=reduce) > 0
    out = run_mesh_triangle(lib, s, reduce=reduce, stream=stream, to_dev=dev, to_host=host, world_regions=True, start=5, blocks=True)
    check_against_record(name, *out, ref, s, slot_of=lambda k: 2 * k + 1, start=5, reduce=reduce)
    again = run_mesh_triangle(lib, s, reduce=reduce, stream=stream, to_dev=dev, to_host=host)
    for k in range(len(s["pairs"])):  # the block's position may differ between runs, its rows may not
        (a0, n0), (a1, n1) = first[1][k], again[1][k]
        assert n0 == n1 and np.array_equal(first[2]["data"][a0:a0 + n0], again[2]["data"][a1:a1 + n1])
        assert np.array_equal(first[2]["key"][a0:a0 + n0], again[2]["key"][a1:a1 + n1])


def _large_mesh_many_batches(lib, **conv):
    import oracle_mesh_triangle as om
    import oracle_reduce as orr

    p, t = mc.grid_mesh(120, 120, 0.6, 0.6, height=lambda x, y: 0.002 * np.sin(40 * x) * np.cos(31 * y))
    q = [0.0, 0.0, float(np.sin(0.15)), float(np.cos(0.15))]
    s = mc._tables([dict(type=mc.MESH, points=p, tris=t, xform=[0, 0, 0, 0, 0, 0, 1], scale=[1, 1, 1], margin=0.0, gap=0.002),
                    dict(type=mc.BOX, xform=[0.01, 0.02, 0.0195, *q], scale=[0.09, 0.07, 0.02], margin=0.0, gap=0.002)])
    s["pairs"] = np.array([[1, 0]], np.int32)
    triples, c = om.triangle_contacts(s)
    assert len(triples) > 520  # three batches at least
    _, blk, rows, total = run_mesh_triangle(lib, s, reduce=0, blocks=True, **conv)  # (450 blocks: two rounds of the block pass)
    r0, cnt = blk[0]
    assert total == cnt == len(c["fp"]) and np.array_equal(rows["key"][r0:r0 + cnt], c["fp"])
    assert np.array_equal(rows["data"][r0:r0 + cnt, 0:3], c["pos"]) and np.array_equal(rows["data"][r0:r0 + cnt, 6], c["depth"])
    want = orr.reduce_buffered_contacts(c)
    _, blk, rows, total = run_mesh_triangle(lib, s, reduce=1, **conv)
    r0, cnt = blk[0]
    assert total == cnt == len(want["fp"]) and np.array_equal(rows["key"][r0:r0 + cnt], want["fp"])
    assert np.array_equal(rows["data"][r0:r0 + cnt, 0:3], want["pos"]) and np.array_equal(rows["data"][r0:r0 + cnt, 6], want["depth"])
    assert np.array_equal(rows["data"][r0:r0 + cnt, 3:6], want["normal"])


def test_emulated_kernel_large_mesh_many_batches(emu):
    """A 120 x 120 grid (28 800 triangles, > 520 candidates under a wide flat box: several batches per pair, leftovers carried
    between scan rounds) against the checker, bit for bit, with and without the reduction."""
    _large_mesh_many_batches(emu)


@pytest.mark.gpu
def test_hip_mesh_triangle_large_mesh_many_batches():
    """The same scene on the MI355X."""
    import torch

    from newton_amd import _lib

    _large_mesh_many_batches(_lib.load(), stream=C.c_void_p(torch.cuda.current_stream().cuda_stream),
                             to_dev=lambda x: torch.from_numpy(x).to("cuda:0"), to_host=lambda x: x.cpu().numpy())
