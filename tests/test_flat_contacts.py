"""Contacts outside the environment tiles (SURVEY.md section 8, rows a22 / a7 for mesh-SDF rows): the float32 checker
oracle/oracle_flat_contacts.py and the emulated gfx950 kernels of nt_flat_contacts.hip against the record of the reference's own
write_contact and eval_body_contact executed on the stand-in (tests/golden/flat_contact_reference_vectors.npz)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(HERE, "emu"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
NAMES = ["dynamic_pairs", "with_static_shapes", "per_contact_properties"]
FIELDS = ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")


def _ref():
    return np.load(os.path.join(HERE, "golden", "flat_contact_reference_vectors.npz"))


@pytest.mark.parametrize("name", NAMES)
def test_checker_reproduces_the_reference_writer_and_force_kernel(name):
    import flat_contact_cases as fc
    import oracle_flat_contacts as O

    ref, case = _ref(), fc.make(name)
    w = O.write_rows(case["rows"], case["body_q"], case["shape_body"], case["shape_gap"])
    acc = np.flatnonzero(w["accepted"])
    assert len(acc) == int(ref[f"{name}/count"][0]) < len(case["rows"]["key"])  # some rows lie beyond the gap
    for k in FIELDS:
        assert np.array_equal(w[k][acc], ref[f"{name}/{k}"]), k
    if case["props"] is not None:
        assert np.array_equal(acc, ref[f"{name}/accepted"])
    f = O.eval_body_contact(w, case["body_q"], case["body_qd"], case["body_com"], case["mat"], case["shape_body"],
                            case["friction_smoothing"], case["props"])
    assert np.abs(ref[f"{name}/body_f"]).max() > 0.1
    assert np.array_equal(f, ref[f"{name}/body_f"])  # same summation order (ascending rows): bit for bit


def run_flat_stage(lib, case, to_dev=lambda a: a, to_host=lambda a: a, ptr=lambda a: a.ctypes.data, stream=None):
    """nt_contact_rows_write + nt_eval_body_contact_flat on one case -> (flat contact arrays, body_f)."""
    from newton_amd import _lib as L

    r = case["rows"]
    n = len(r["key"])
    # the rows the mesh-SDF stage would hand over: pair table + (pair index, 9 floats) per row
    pairs, inv = np.unique(np.stack([r["shape_a"], r["shape_b"]], axis=1), axis=0, return_inverse=True)
    data = np.concatenate([r["center"], r["normal"], r["distance"][:, None], r["margin_a"][:, None], r["margin_b"][:, None]],
                          axis=1).astype(np.float32)
    keep = dict(pairs=to_dev(np.ascontiguousarray(pairs, np.int32)), row_pair=to_dev(np.ascontiguousarray(inv.reshape(-1), np.int32)),
                data=to_dev(np.ascontiguousarray(data)), body_q=to_dev(case["body_q"]), body_qd=to_dev(case["body_qd"]),
                body_com=to_dev(case["body_com"]), shape_body=to_dev(case["shape_body"]), gap=to_dev(case["shape_gap"]),
                count=to_dev(np.array([n], np.int32)))
    out = {k: to_dev(np.full(n, 7, np.int32)) for k in ("shape0", "shape1")}
    out.update({k: to_dev(np.full((n, 3), 9.0, np.float32)) for k in ("point0", "point1", "offset0", "offset1", "normal")})
    out.update({k: to_dev(np.full(n, 9.0, np.float32)) for k in ("margin0", "margin1")})
    a = L.nt_contact_rows()
    a.row_count, a.row_count_device, a.row_pair, a.pairs, a.row_data = n, ptr(keep["count"]), ptr(keep["row_pair"]), ptr(keep["pairs"]), ptr(keep["data"])
    a.body_q, a.shape_body, a.shape_gap = ptr(keep["body_q"]), ptr(keep["shape_body"]), ptr(keep["gap"])
    for k in out:
        setattr(a, "out_" + k, ptr(out[k]))
    assert lib.nt_contact_rows_write(C.byref(a), stream) == 0
    mat = {k: to_dev(v) for k, v in case["mat"].items()}
    body_f = to_dev(np.zeros((len(case["body_q"]), 6), np.float32))
    f = L.nt_flat_contact_forces()
    f.body_q, f.body_qd, f.body_com = ptr(keep["body_q"]), ptr(keep["body_qd"]), ptr(keep["body_com"])
    f.shape_ke, f.shape_kd, f.shape_kf, f.shape_ka, f.shape_mu = (ptr(mat[k]) for k in ("ke", "kd", "kf", "ka", "mu"))
    f.shape_body, f.contact_count, f.contact_max = ptr(keep["shape_body"]), ptr(keep["count"]), n
    for k in ("point0", "point1", "normal", "shape0", "shape1", "margin0", "margin1"):
        setattr(f, k, ptr(out[k]))
    if case["props"] is not None:
        props = {k: to_dev(v) for k, v in case["props"].items()}
        f.contact_stiffness, f.contact_damping, f.contact_friction_scale = ptr(props["stiffness"]), ptr(props["damping"]), ptr(props["friction"])
        keep["props"] = props
    f.friction_smoothing, f.body_f = float(case["friction_smoothing"]), ptr(body_f)
    assert lib.nt_eval_body_contact_flat(C.byref(f), stream) == 0
    return {k: to_host(v) for k, v in out.items()}, to_host(body_f)


def check_flat_stage(name, out, body_f):
    import flat_contact_cases as fc
    import oracle_flat_contacts as O

    ref, case = _ref(), fc.make(name)
    acc = np.flatnonzero(out["shape0"] >= 0)
    assert len(acc) == int(ref[f"{name}/count"][0])
    for k in FIELDS:
        assert np.array_equal(out[k][acc], ref[f"{name}/{k}"]), k  # the writer: bit for bit the reference's rows, in order
    rej = np.flatnonzero(out["shape0"] < 0)
    assert np.all(out["shape1"][rej] == -1) and all(np.all(out[k][rej] == 0) for k in FIELDS[2:])
    # forces: float atomics add in arrival order -- the same numbers as the reference's up to the rounding of the sums
    want = ref[f"{name}/body_f"]
    scale = np.abs(want).max()
    assert np.abs(body_f - want).max() <= 2e-6 * scale
    # ... and bit for bit the checker's sum when the rows are added in the checker's order on a single lane is not observable
    # here; the per-row force is pinned through the single-contact launches below
    w = O.write_rows(case["rows"], case["body_q"], case["shape_body"], case["shape_gap"])
    return w


@pytest.fixture(scope="module")
def emu():
    import harness

    return harness.lib()


@pytest.mark.parametrize("name", NAMES)
def test_emulated_flat_contact_kernels_against_the_reference(emu, name):
    import flat_contact_cases as fc
    import oracle_flat_contacts as O

    case = fc.make(name)
    out, body_f = run_flat_stage(emu, case)
    w = check_flat_stage(name, out, body_f)
    # one contact per launch: no summation order left, the kernel's force must be the checker's bit for bit
    acc = np.flatnonzero(w["accepted"])[:6]
    for i in acc:
        one = dict(case, rows={k: v[i:i + 1] for k, v in case["rows"].items()},
                   props=None if case["props"] is None else {k: v[i:i + 1] for k, v in case["props"].items()})
        o1, f1 = run_flat_stage(emu, one)
        w1 = O.write_rows(one["rows"], case["body_q"], case["shape_body"], case["shape_gap"])
        want = O.eval_body_contact(w1, case["body_q"], case["body_qd"], case["body_com"], case["mat"], case["shape_body"],
                                   case["friction_smoothing"], one["props"])
        assert np.array_equal(f1, want)
