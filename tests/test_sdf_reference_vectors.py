"""Texture-SDF samplers, the Brent edge search and the mesh-vs-SDF narrow phase against tests/golden/sdf_reference_vectors.npz:
the record of the REFERENCE's own source (sdf_texture.py samplers, sdf_contact.py do_edge_sdf_collision and
mesh_sdf_collision_kernel) executed by tests/golden/make_sdf_reference_vectors.py.  Bit for bit: the float32 checker
(oracle/oracle_sdf.py), the kernel sources compiled for the host (tests/emu) and -- under -m gpu -- the MI355X."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import sdf_cases  # noqa: E402

REF = np.load(os.path.join(ROOT, "tests", "golden", "sdf_reference_vectors.npz"))
SDFS = sdf_cases.sdfs()
SCENES = sdf_cases.pair_scenes()


@pytest.mark.parametrize("name", sorted(SDFS))
def test_checker_samplers_and_edge_search_are_the_executed_reference(name):
    import oracle_sdf as O

    t = SDFS[name]
    o = O.OracleSDF(t)
    pts = REF[f"sample/{name}/points"]
    assert np.array_equal(pts, sdf_cases.query_points(t))  # the cases are reproducible
    assert np.array_equal(np.array([o.sample(p) for p in pts]), REF[f"sample/{name}/value"])         # texture_sample_sdf
    assert np.array_equal(np.array([o.sample_hw(p) for p in pts]), REF[f"sample/{name}/value_hw"])   # texture_sample_sdf_hw
    assert np.array_equal(np.array([o.sample_grad_fd(p) for p in pts]), REF[f"sample/{name}/grad_hw"])
    half = len(pts) // 2
    pair = np.array([[o.sample_hw(pts[i]), o.sample_hw(pts[-1 - i])] for i in range(half)])
    assert np.array_equal(pair, REF[f"sample/{name}/pair_hw"])                                         # the paired fetch = two single ones
    # the software value + analytic gradient (texture_sample_sdf_grad) through the host sampler of newton_amd.sdf
    v, g = t.sample_grad(pts)
    assert np.abs(v - REF[f"sample/{name}/grad_value"]).max() <= 1e-6
    inside = np.all((pts > t.box_lower) & (pts < t.box_upper), axis=1)
    assert np.abs(g[inside] - REF[f"sample/{name}/grad"][inside]).max() <= 2e-3 * max(1.0, np.abs(REF[f"sample/{name}/grad"]).max())
    assert np.array_equal(t.sample_at_voxel(REF[f"sample/{name}/voxels"]), REF[f"sample/{name}/at_voxel"])
    res = REF[f"edge/{name}/result"]
    for a, b, p, r in zip(REF[f"edge/{name}/v0"], REF[f"edge/{name}/v1"], REF[f"edge/{name}/precision"], res):
        mid = o.sample_hw((a + b) * np.float32(0.5))
        d, pt, ep = O.edge_search(o, a, b, mid, p)
        assert mid == r[0] and d == r[1] and np.array_equal(pt, r[2:5]) and ep == int(r[5])
    assert len(set(res[:, 5].astype(int))) == 3  # interior, v0 and v1 winners all occur


def _rows_equal(rows, pairs, ref_rows):
    """(pair index, key, centre, normal, distance, margins) rows of the checker / a kernel == the reference's ContactData, in the
    reference's (pair, mode, edge) emission order."""
    assert len(rows) == len(ref_rows)
    for x, y in zip(rows, ref_rows):
        a, b = pairs[int(x[0])]
        assert (a, b, int(x[1])) == (int(y[0]), int(y[1]), int(y[2]))
        assert np.array_equal(np.asarray(x[2], np.float32), y[3:6]) and np.array_equal(np.asarray(x[3], np.float32), y[6:9])
        assert np.float32(x[4]) == y[9] and np.float32(x[5]) == y[10] and np.float32(x[6]) == y[11]


@pytest.mark.parametrize("name", sorted(SCENES))
def test_checker_mesh_sdf_kernel_is_the_executed_reference(name):
    import oracle_sdf as O

    s = SCENES[name]
    rows = O.mesh_sdf_collide(s["pairs"], s["X"], s["data"], s["gap"], s["sdf_index"], s["sdfs"], s["er"], s["ec"], s["eh"])
    _rows_equal(rows, s["pairs"], REF[f"pair/{name}/rows"])
    if name != "cube_apart":
        assert len(rows) >= 7


def _sorted_like_reference(pair, key, data):
    """Kernel rows (appended in arrival order) -> the reference's (pair, mode, edge) order."""
    mode, edge = (key >> 1) & 1, key >> 2
    o = np.lexsort((edge, mode, pair))
    return [(pair[i], key[i], data[i, 0:3], data[i, 3:6], data[i, 6], data[i, 7], data[i, 8]) for i in o]


@pytest.fixture(scope="module")
def emu():
    import build

    return C.CDLL(build.build())


@pytest.mark.parametrize("name", sorted(SDFS))
def test_emulated_sampler_kernels_are_the_executed_reference(emu, name):
    from test_sdf_contact import _emu_sdf

    t = SDFS[name]
    d, keep = _emu_sdf(emu, t)
    pts = np.ascontiguousarray(REF[f"sample/{name}/points"])
    n = len(pts)
    dist, grad, hw = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    assert emu.nt_sdf_sample(C.byref(d), C.c_void_p(pts.ctypes.data), n, C.c_void_p(dist.ctypes.data), C.c_void_p(grad.ctypes.data), None) == 0
    assert emu.nt_sdf_sample_hw(C.byref(d), C.c_void_p(pts.ctypes.data), n, C.c_void_p(hw.ctypes.data), None) == 0
    assert np.array_equal(dist, REF[f"sample/{name}/value"]) and np.array_equal(hw, REF[f"sample/{name}/value_hw"])
    assert np.array_equal(grad, REF[f"sample/{name}/grad_hw"])
    vox = np.ascontiguousarray(REF[f"sample/{name}/voxels"])
    av = np.zeros(len(vox), np.float32)
    assert emu.nt_sdf_sample_voxels(C.byref(d), C.c_void_p(vox.ctypes.data), len(vox), C.c_void_p(av.ctypes.data), None) == 0
    assert np.array_equal(av, REF[f"sample/{name}/at_voxel"])
    del keep


@pytest.mark.parametrize("name", sorted(SCENES))
def test_emulated_mesh_sdf_kernel_is_the_executed_reference(emu, name):
    from test_sdf_contact import _emu_mesh_sdf

    s = SCENES[name]
    pair, key, data = _emu_mesh_sdf(emu, s, reduced=False)
    _rows_equal(_sorted_like_reference(pair, key, data), s["pairs"], REF[f"pair/{name}/rows"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SDFS))
def test_device_sampler_kernels_are_the_executed_reference(name):
    import torch

    from newton_amd import _lib
    from newton_amd.sdf_device import DeviceSDF

    dev = DeviceSDF(SDFS[name])
    lib = _lib.load()
    pts = torch.from_numpy(np.ascontiguousarray(REF[f"sample/{name}/points"])).cuda()
    dist, grad = dev.sample(pts, grad=True)
    hw = torch.zeros(len(pts), dtype=torch.float32, device="cuda")
    _lib.check(lib.nt_sdf_sample_hw(C.byref(dev.desc), pts.data_ptr(), len(pts), hw.data_ptr(), None), "nt_sdf_sample_hw")
    vox = torch.from_numpy(np.ascontiguousarray(REF[f"sample/{name}/voxels"])).cuda()
    av = torch.zeros(len(vox), dtype=torch.float32, device="cuda")
    _lib.check(lib.nt_sdf_sample_voxels(C.byref(dev.desc), vox.data_ptr(), len(vox), av.data_ptr(), None), "nt_sdf_sample_voxels")
    torch.cuda.synchronize()
    assert np.array_equal(dist.cpu().numpy(), REF[f"sample/{name}/value"]) and np.array_equal(hw.cpu().numpy(), REF[f"sample/{name}/value_hw"])
    assert np.array_equal(grad.cpu().numpy(), REF[f"sample/{name}/grad_hw"])
    assert np.array_equal(av.cpu().numpy(), REF[f"sample/{name}/at_voxel"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENES))
def test_device_mesh_sdf_kernel_is_the_executed_reference(name):
    from newton_amd.sdf_device import DeviceSDF, mesh_sdf_collide

    s = SCENES[name]
    r = mesh_sdf_collide(s["pairs"], s["X"], s["data"], s["gap"], s["sdf_index"], [DeviceSDF(t) for t in s["sdfs"]], s["er"],
                         s["ec"], s["eh"])
    data = np.concatenate([r["center"], r["normal"], r["distance"][:, None], r["margin0"][:, None], r["margin1"][:, None]], axis=1)
    _rows_equal(_sorted_like_reference(r["pair"], r["key"], data), s["pairs"], REF[f"pair/{name}/rows"])
