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
