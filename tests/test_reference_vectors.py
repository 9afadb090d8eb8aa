"""The C++ checker (oracle/) against golden vectors produced by EXECUTING the reference's own SolverXPBD source
(tests/golden/make_xpbd_reference_vectors.py: /root/reference/newton/_src/solvers/xpbd/solver_xpbd.py + kernels.py +
solvers/solver.py run on a pure-Python stand-in for the Warp API, thread by thread in ascending tid order).  Teacher-forced:
every step starts from the reference's state.  This pins the checker's solver arithmetic and orchestration -- joint forces,
integration, contact and joint position solves, delta application, restitution -- to the reference code itself; the only
restated layer left is the fp32 operation order inside the Warp builtins (quat_rotate, transform products ...), which the
stand-in and oracle/wp_builtins.h share, and the libm behind sin / cos / acos (numpy float32 vs glibc).
The GPU twin (HIP path vs the same vectors) is tests/test_zx_round2_gpu.py::test_hip_path_against_reference_vectors."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = os.path.join(HERE, "golden", "xpbd_reference_vectors.npz")


sys.path.insert(0, os.path.join(HERE, "golden"))


def _cases():
    import reference_cases as rc

    return rc


def _errors(q, qd, q_ref, qd_ref):
    pos = np.abs(q[:, :3] - q_ref[:, :3]).max()
    rot = np.minimum(np.abs(q[:, 3:] - q_ref[:, 3:]).max(axis=1), np.abs(q[:, 3:] + q_ref[:, 3:]).max(axis=1)).max()
    return pos, rot, np.abs(qd[:, :3] - qd_ref[:, :3]).max(), np.abs(qd[:, 3:] - qd_ref[:, 3:]).max()


NAMES = ["quadruped_standing", "quadruped_impact_restitution", "pendulum", "joint_zoo", "joint_zoo_free_root",
         "box_stack_no_weighting", "box_stack_sunk_restitution", "quadruped_velocity_from_delta",
         "box_stack_velocity_from_delta_restitution", "quadruped_report", "box_stack_report", "semi/pendulum", "semi/joint_zoo", "semi/box_stack",
         "semi/box_stack_contact_props", "semi/quadruped", "fs/pendulum", "fs/joint_zoo", "fs/joint_zoo_free_root", "fs/quadruped", "fs/quadruped_interval3",
         "fs/joint_zoo_interval2", "fs/free_child", "fs/free_child_free_root"]


@pytest.mark.parametrize("name", NAMES)
def test_checker_reproduces_the_reference_solver_step_by_step(oracle_lib, name):
    import oracle_bridge as ob

    rc = _cases()
    assert sorted(NAMES) == sorted(rc.cases())
    ref = np.load(VEC)
    case = rc.cases()[name]
    model = rc.prepare(case)
    if case.get("report"):
        model.request_state_attributes("body_parent_f")
        model.request_contact_attributes("force")
    orc = ob.Oracle(model)
    worst_report, force_seen = np.zeros(2), 0.0
    semi, fs = case.get("solver") == "semi_implicit", case.get("solver") == "featherstone"
    worst, worst_joint = np.zeros(4), np.zeros(2)
    for k in range(case["steps"]):
        q, qd = ref[f"{name}/body_q{k}"], ref[f"{name}/body_qd{k}"]
        ct = orc.contacts()
        orc.collide(q, ct)
        n = int(ct.count[0])
        assert n == int(ref[f"{name}/contacts{k}"][0])
        if case.get("props") is not None:
            ct.set_properties(np.full(max(n, 1), case["props"][0]), np.full(max(n, 1), case["props"][1]), np.full(max(n, 1), case["props"][2]))
        s_in, s_out = ob.OracleState(model, q, qd), ob.OracleState(model, q, qd)
        if fs:
            for s_ in (s_in, s_out):
                s_.joint_q[:], s_.joint_qd[:] = ref[f"{name}/joint_q{k}"], ref[f"{name}/joint_qd{k}"]
            orc.featherstone_step(s_in, s_out, orc.control(), ct if n else None, case["dt"], **case["kw"])
            worst_joint = np.maximum(worst_joint, [np.abs(s_out.joint_q - ref[f"{name}/joint_q{k + 1}"]).max(),
                                                   np.abs(s_out.joint_qd - ref[f"{name}/joint_qd{k + 1}"]).max()])
        elif semi:
            orc.semi_implicit_step(s_in, s_out, orc.control(), ct if n else None, case["dt"], **case["kw"])
        elif case.get("report"):
            force = np.zeros((ct.max, 6), np.float32)
            orc.xpbd_step(s_in, s_out, orc.control(), ct if n else None, case["dt"], contact_force_out=force, **case["kw"], **case.get("attrs", {}))
            worst_report = np.maximum(worst_report, [np.abs(force[:n] - ref[f"{name}/contact_force{k + 1}"][:n]).max(),
                                                     np.abs(s_out.body_parent_f - ref[f"{name}/body_parent_f{k + 1}"]).max()])
            force_seen = max(force_seen, float(np.abs(ref[f"{name}/contact_force{k + 1}"][:n]).max()))
        else:
            orc.xpbd_step(s_in, s_out, orc.control(), ct if n else None, case["dt"], **case["kw"], **case.get("attrs", {}))
        e = _errors(s_out.body_q, s_out.body_qd, ref[f"{name}/body_q{k + 1}"], ref[f"{name}/body_qd{k + 1}"])
        worst = np.maximum(worst, e)
    print(name, "max abs error vs the reference run: pos %.3g rot %.3g lin vel %.3g ang vel %.3g" % tuple(worst))
    # one step from identical inputs.  Measured: bit-identical positions and linear velocities in nearly every case; rotations
    # within 1.5e-8 and angular velocities within 3e-6 where asin / acos / atan2 enter (numpy float32 vs glibc)
    lin_tol, ang_tol = 1e-6, 1e-5
    if case.get("attrs", {}).get("compute_body_velocity_from_position_delta"):
        # velocities are finite differences of the poses: the sub-ulp libm differences in the poses come back times 1 / dt
        # (measured 3.0e-5 / 5.9e-5 at dt = 1e-3 where the poses differ by 6e-8; bit-identical in the box-stack case)
        lin_tol, ang_tol = max(lin_tol, 1e-7 / case["dt"]), max(ang_tol, 2e-7 / case["dt"])
    assert worst[0] <= 1e-7 and worst[1] <= 1e-7 and worst[2] <= lin_tol and worst[3] <= ang_tol, worst
    if case.get("report"):
        print(name, "reporting: max abs error contacts.force %.3g [N], body_parent_f %.3g [N]" % tuple(worst_report))
        assert worst_report[0] <= 1e-3 and worst_report[1] <= 1e-3 and force_seen > 1.0, (worst_report, force_seen)
    if fs:
        print(name, "joint space: max abs error joint_q %.3g joint_qd %.3g" % tuple(worst_joint))
        assert worst_joint[0] <= 1e-7 and worst_joint[1] <= 1e-5, worst_joint


COLLIDE_NAMES = ["hull_bin_a", "hull_bin_b", "pair_matrix_a", "pair_matrix_b", "mixed_primitives_a", "mixed_primitives_b", "mixed_primitives_c", "box_stack_a", "box_stack_b",
                 "quadruped_cylinders", "quadruped_box_feet"]


@pytest.mark.parametrize("name", COLLIDE_NAMES)
def test_checker_collide_reproduces_the_reference_collision_kernels(oracle_lib, name):
    """tests/golden/make_collide_reference_vectors.py ran the reference's compute_shape_aabbs, primitive narrow phase, GJK / MPR +
    manifold narrow phase and write_contact (unmodified source, on the Warp stand-in) on these scenes; the checker's collide()
    must produce the same AABBs and the same flat contact arrays in the same append order."""
    import collide_cases as cc
    import oracle_bridge as ob

    assert sorted(COLLIDE_NAMES) == sorted(cc.cases())
    ref = np.load(os.path.join(HERE, "golden", "collide_reference_vectors.npz"))
    model, _ = cc.cases()[name]()
    body_q = ref[f"{name}/body_q"]
    orc = ob.Oracle(model)
    ct = orc.contacts()
    pairs, lo, hi = orc.collide(body_q, ct)
    assert np.array_equal(np.asarray(pairs, np.int32), ref[f"{name}/pairs"])
    n = int(ref[f"{name}/count"][0])
    assert int(ct.count[0]) == n and n > 0
    assert np.array_equal(ct.shape0[:n], ref[f"{name}/shape0"]) and np.array_equal(ct.shape1[:n], ref[f"{name}/shape1"])
    finite = np.abs(ref[f"{name}/aabb_lower"]) < 1e5  # (infinite planes: +-1e6 boxes)
    err = {"aabb": max(np.abs(lo - ref[f"{name}/aabb_lower"])[finite].max(), np.abs(hi - ref[f"{name}/aabb_upper"])[finite].max())}
    for k in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
        err[k] = float(np.abs(getattr(ct, k)[:n] - ref[f"{name}/{k}"]).max())
    print(name, "contacts", n, "max abs error vs the reference kernels:", {k: float("%.3g" % v) for k, v in err.items()})
    assert all(v <= 2e-6 for v in err.values()), err


@pytest.mark.parametrize("name", ["barrel_wide", "barrel_tight"])
def test_checker_collide_reproduces_the_reference_on_barrel_cylinders(oracle_lib, name):
    """Barrel cylinders (support_function.py:284-305; plane route narrow_phase.py:682-686, sphere route :847,999): the record of
    make_collide_reference_vectors.py --barrel (the reference's kernels, executed) against the checker, same append order."""
    import collide_cases as cc
    import oracle_bridge as ob

    ref = np.load(os.path.join(HERE, "golden", "collide_barrel_reference_vectors.npz"))
    model, _ = cc.barrel_cases()[name]()
    body_q = ref[f"{name}/body_q"]
    orc = ob.Oracle(model)
    ct = orc.contacts()
    pairs, lo, hi = orc.collide(body_q, ct)
    assert np.array_equal(np.asarray(pairs, np.int32), ref[f"{name}/pairs"])
    n = int(ref[f"{name}/count"][0])
    assert int(ct.count[0]) == n and n > 0
    assert 0 < int(ref[f"{name}/count_analytic"][0]) < n and len(ref[f"{name}/gjk_pairs"]) > 0  # both routes of the plane pairs occur
    assert np.array_equal(ct.shape0[:n], ref[f"{name}/shape0"]) and np.array_equal(ct.shape1[:n], ref[f"{name}/shape1"])
    finite = np.abs(ref[f"{name}/aabb_lower"]) < 1e5
    err = {"aabb": max(np.abs(lo - ref[f"{name}/aabb_lower"])[finite].max(), np.abs(hi - ref[f"{name}/aabb_upper"])[finite].max())}
    for k in ("point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
        err[k] = float(np.abs(getattr(ct, k)[:n] - ref[f"{name}/{k}"]).max())
    print(name, "contacts", n, "max abs error vs the reference kernels:", {k: float("%.3g" % v) for k, v in err.items()})
    assert all(v <= 2e-6 for v in err.values()), err


@pytest.mark.parametrize("name", ["joint_zoo", "joint_zoo_free_root", "quadruped", "pendulum"])
def test_checker_eval_fk_reproduces_the_reference(oracle_lib, name):
    """newton.eval_fk of the reference (articulation.py:500-573, executed on the stand-in) vs the checker's o_eval_fk and the
    host-side numpy FK the Python package uses for host models."""
    import oracle_bridge as ob

    import newton_amd as nt

    rc = _cases()
    ref = np.load(VEC)
    model = rc.fk_cases()[name]()
    jq, jqd = ref[f"fk/{name}/joint_q"], ref[f"fk/{name}/joint_qd"]
    assert np.array_equal(jq, np.asarray(model.joint_q, np.float32))
    bq, bqd = ob.Oracle(model).eval_fk(jq, jqd)
    e = _errors(bq, bqd, ref[f"fk/{name}/body_q"], ref[f"fk/{name}/body_qd"])
    print(name, "eval_fk checker vs reference: pos %.3g rot %.3g lin vel %.3g ang vel %.3g" % e)
    assert max(e[:2]) <= 1e-7 and max(e[2:]) <= 1e-6, e
    hq, hqd = nt.articulation.eval_fk_numpy(model, jq, jqd)
    e = _errors(np.asarray(hq, np.float32).reshape(-1, 7), np.asarray(hqd, np.float32).reshape(-1, 6), ref[f"fk/{name}/body_q"],
                ref[f"fk/{name}/body_qd"])
    print(name, "eval_fk host numpy vs reference: pos %.3g rot %.3g lin vel %.3g ang vel %.3g" % e)
    assert max(e[:2]) <= 2e-6 and max(e[2:]) <= 2e-5, e


@pytest.mark.parametrize("name,frames", [("box_stack", 6), ("mixed_primitives", 8)])
def test_match_checker_reproduces_the_reference_contact_matcher(name, frames):
    """tests/golden/make_match_reference_vectors.py ran the reference's ContactMatcher (contact_match.py:602-1055) frame by frame;
    oracle/oracle_match.py must return the same match indices (including MATCH_NOT_FOUND / MATCH_BROKEN and claim races)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import oracle_match as om
    from scenes import box_stack_scene, mixed_primitive_scene

    ref = np.load(os.path.join(HERE, "golden", "match_reference_vectors.npz"))
    model = box_stack_scene(1, n_boxes=4, seed=2, jitter=5e-3) if name == "box_stack" else mixed_primitive_scene(1, seed=4)
    shape_body = np.asarray(model.shape_body)
    prev = None
    seen = set()
    for k in range(frames):
        g = lambda f: ref[f"{name}/{k}/{f}"]  # noqa: E731
        mid = om.midpoints(g("body_q"), shape_body, g("shape0"), g("shape1"), g("point0"), g("point1"))
        want = g("match")
        if prev is None:
            assert np.all(want == -1)
        else:
            got = om.match(g("keys"), mid, g("normal"), *prev)
            assert np.array_equal(got, want), (k, got, want)
            seen |= set(np.sign(want).tolist()) | ({-2} if np.any(want == -2) else set())
        prev = (g("keys"), mid, g("normal"))
    assert 1 in seen or 0 in seen
    assert -2 in seen  # broken matches occur in both recordings


BP_VEC = os.path.join(HERE, "golden", "broadphase_reference_vectors.npz")


@pytest.mark.parametrize("variant", ["plain", "filtered", "immovable"])
@pytest.mark.parametrize("name", ["single_world", "multiple_worlds", "shape_flags", "per_shape_gap"])
def test_broad_phase_checker_reproduces_the_reference_classes(oracle_lib, name, variant):
    """World map and candidate lists of the reference's own precompute_world_map / BroadPhaseAllPairs / BroadPhaseExplicit /
    BroadPhaseSAP, executed on the stand-in (tests/golden/make_broadphase_reference_vectors.py): N x N and explicit in the
    reference's append order (ascending tid), sort-and-sweep as a set (its order depends on the segmented sort)."""
    import ctypes as C

    import broadphase_cases as bc
    from newton_amd.geometry import precompute_world_map
    from test_broad_phase_standalone import _oracle

    ref = np.load(BP_VEC)
    v = bc.variants(name)[variant]
    key = f"{name}/{variant}"
    index_map, ends = precompute_world_map(v["world"], v["flags"])
    assert np.array_equal(index_map, ref[f"{key}/index_map"]) and np.array_equal(ends, ref[f"{key}/slice_ends"])
    kw = dict(filter_pairs=v["filter_pairs"], shape_body=v["shape_body"], body_flags=v["body_flags"], include=v["include"])
    args = (v["lower"], v["upper"], v["gap"], v["group"], v["world"], v["flags"])
    count, pairs = _oracle(oracle_lib, "nxn", *args, **kw)
    want = ref[f"{key}/nxn_pairs"]
    assert len(want) > 5 and count == len(want) and np.array_equal(pairs, want)
    count, pairs = _oracle(oracle_lib, "sap", *args, **kw)
    want = ref[f"{key}/sap_pairs"]
    assert count == len(want) and len({tuple(p) for p in pairs}) == count
    assert {tuple(p) for p in pairs} == {tuple(p) for p in want}
    # explicit list (no group / world / excluded-pair test: broad_phase_nxn.py:444-535)
    ep = np.ascontiguousarray(v["explicit_pairs"], np.int32)
    out = np.zeros((len(ep) + 1, 2), np.int32)
    fi, ii = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    p = lambda a, t: a.ctypes.data_as(t) if a is not None else None  # noqa: E731
    oracle_lib.o_broadphase_explicit.restype = C.c_int
    count = oracle_lib.o_broadphase_explicit(p(v["lower"], fi), p(v["upper"], fi), p(v["gap"], fi), p(ep, ii), len(ep),
                                             p(v["shape_body"], ii), p(v["body_flags"], ii), int(v["include"]), p(out, ii),
                                             len(out))
    want = ref[f"{key}/explicit_pairs"]
    assert count == len(want) and np.array_equal(out[:count], want)


@pytest.mark.parametrize("variant", ["swept", "swept_capped", "swept_capped_zero", "swept_filtered"])
@pytest.mark.parametrize("name", ["single_world", "multiple_worlds", "shape_flags", "per_shape_gap"])
def test_broad_phase_checker_reproduces_the_reference_classes_with_displacements(oracle_lib, name, variant):
    """Swept AABBs (the broad phase of the speculative-contact mode): the reference classes launched with shape_displacement --
    check_aabb_overlap_moving (broad_phase_common.py:41-85) -- and, for BroadPhaseSAP, sort_axis_displacement_limit
    (_sap_project_aabb, broad_phase_sap.py:44-79: a capped extension of the projected interval loses fast pairs, which the checker
    must lose as well).  N x N and explicit in append order, sort-and-sweep as a set."""
    import ctypes as C

    import broadphase_cases as bc
    from test_broad_phase_standalone import _oracle

    ref = np.load(BP_VEC)
    v = bc.variants(name)[variant]
    key = f"{name}/{variant}"
    kw = dict(filter_pairs=v["filter_pairs"], shape_body=v["shape_body"], body_flags=v["body_flags"], include=v["include"],
              displacement=v["displacement"])
    args = (v["lower"], v["upper"], v["gap"], v["group"], v["world"], v["flags"])
    count, pairs = _oracle(oracle_lib, "nxn", *args, **kw)
    want = ref[f"{key}/nxn_pairs"]
    assert len(want) > 5 and count == len(want) and np.array_equal(pairs, want)
    assert not np.array_equal(want, ref[f"{name}/plain/nxn_pairs"])  # the displacements change the result
    count, pairs = _oracle(oracle_lib, "sap", *args, limit=v["limit"], **kw)
    want = ref[f"{key}/sap_pairs"]
    assert count == len(want) and len({tuple(p) for p in pairs}) == count
    assert {tuple(p) for p in pairs} == {tuple(p) for p in want}
    ep = np.ascontiguousarray(v["explicit_pairs"], np.int32)
    out = np.zeros((len(ep) + 1, 2), np.int32)
    fi, ii = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    p = lambda a, t: a.ctypes.data_as(t) if a is not None else None  # noqa: E731
    disp = np.ascontiguousarray(v["displacement"], np.float32)
    oracle_lib.o_broadphase_explicit_swept.restype = C.c_int
    count = oracle_lib.o_broadphase_explicit_swept(p(v["lower"], fi), p(v["upper"], fi), p(v["gap"], fi), p(ep, ii), len(ep),
                                                   p(v["shape_body"], ii), p(v["body_flags"], ii), int(v["include"]), p(disp, fi),
                                                   p(out, ii), len(out))
    want = ref[f"{key}/explicit_pairs"]
    assert count == len(want) and np.array_equal(out[:count], want)


def test_capped_sort_axis_extension_loses_pairs_in_the_reference_too():
    """The record itself: with sort_axis_displacement_limit the reference's SAP returns a strict subset of its N x N result on
    the gap case (the projected intervals decide which pairs are tested), and the same set without the cap."""
    ref = np.load(BP_VEC)
    s = lambda k: {tuple(p) for p in ref[k]}  # noqa: E731
    assert s("per_shape_gap/swept/sap_pairs") == s("per_shape_gap/swept/nxn_pairs")
    assert s("per_shape_gap/swept_capped/sap_pairs") < s("per_shape_gap/swept_capped/nxn_pairs")
    assert s("per_shape_gap/swept_capped_zero/sap_pairs") < s("per_shape_gap/swept_capped/sap_pairs")
