"""Host logic (no GPU): builder / URDF importer / contact-pair rules / env template / C ABI surface."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import newton_amd as nt
from newton_amd import _lib
from newton_amd.enums import GeoType, JointType
from scenes import box_stack_scene, mixed_primitive_scene, quadruped_builder, quadruped_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quadruped_model_matches_survey_counts():
    """per env B=13, J=13, D=18, coords=19, S=13 + 1 global plane, exactly 13 (shape, ground) pairs."""
    q = quadruped_builder()
    assert (q.body_count, q.joint_count, q.shape_count, q.joint_dof_count, q.joint_coord_count) == (13, 13, 13, 18, 19)
    model = quadruped_scene(4, seed=None)
    t = model.env
    assert (t.env_count, t.nb, t.nj, t.nd, t.nc, t.ns, t.ng, t.np, t.cpp) == (4, 13, 13, 18, 19, 13, 1, 13, 4)
    ground = model.shape_count - 1
    want = [(w * 13 + s, ground) for w in range(4) for s in range(13)]
    assert [tuple(p) for p in model.shape_contact_pairs] == want
    assert t.joint_type[0] == JointType.FREE and np.all(t.joint_type[1:] == JointType.REVOLUTE)
    # density-1000 cylinder masses (builder.py:491 default density, inertia.py:151-190)
    assert abs(model.body_mass[0] - 1000 * np.pi * 0.1 ** 2 * 0.75) < 1e-3
    # inverse inertia is recomputed at finalize from the armature-augmented inertia (inertia.py:1096-1110)
    assert np.allclose(model.body_inv_inertia[1] @ model.body_inertia[1], np.eye(3), atol=1e-4)
    assert model.gravity.shape == (5, 3) and np.allclose(model.gravity[:, 2], -9.81)


def test_quadruped_scene_is_the_reference_asset():
    """The bench / test quadruped carries the numbers of newton/examples/assets/quadruped.urdf exactly -- joint origins (the hind
    legs' HAA mounts are turned by rpy "0 0 3.1415" and their HFE offsets flipped, :148-215), axes, limits, collision
    geometry -- checked against tests/golden/quadruped_asset.json (the reference file read with ElementTree by
    tests/golden/make_quadruped_asset_vectors.py, independent of newton_amd.urdf)."""
    import json
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_quadruped_asset_vectors import asset_numbers
    from scenes import quadruped_urdf

    with open(os.path.join(ROOT, "tests", "golden", "quadruped_asset.json")) as f:
        want = json.load(f)
    got = asset_numbers(quadruped_urdf())
    assert [j["name"] for j in got["joints"]] == [j["name"] for j in want["joints"]]
    assert [k["name"] for k in got["links"]] == [k["name"] for k in want["links"]]
    for g, w in zip(got["joints"], want["joints"]):
        assert g == w, (g, w)
    for g, w in zip(got["links"], want["links"]):
        assert g == w, (g, w)
    assert want["joints"][6]["rpy"] == [0.0, 0.0, 3.1415] and want["joints"][7]["xyz"] == [0.0, -0.05, 0.0]


def test_box_stack_pairs_and_cpp():
    model = box_stack_scene(2, n_boxes=8, seed=None)
    t = model.env
    assert (t.nb, t.nj, t.ns, t.np) == (8, 8, 8, 36)  # 8 ground pairs + 28 box-box pairs
    assert t.cpp == 5  # box-box goes through the convex manifold (<= 5 contacts)
    # reference order per world: (global, local) pairs first, then local pairs (builder.py:12935-12962)
    first = [tuple(p) for p in model.shape_contact_pairs[:9]]
    ground = model.shape_count - 1
    assert first[:8] == [(s, ground) for s in range(8)] and first[8] == (0, 1)


def test_collision_filters():
    b = nt.ModelBuilder()
    a = b.add_link()
    c = b.add_link()
    b.add_shape_box(a)
    b.add_shape_box(a)  # same body -> filtered (builder.py:6638-6643)
    b.add_shape_sphere(c)
    j0 = b.add_joint_fixed(-1, a)
    j1 = b.add_joint_revolute(a, c)  # parent/child filtered (collision_filter_parent default True)
    b.add_articulation([j0, j1])
    m = b.finalize()
    assert len(m.shape_contact_pairs) == 0
    b2 = nt.ModelBuilder()
    x = b2.add_body()
    y = b2.add_body()
    b2.add_shape_sphere(x)
    cfg = nt.ShapeConfig(collision_group=2)
    b2.add_shape_sphere(y, cfg=cfg)  # different positive groups never pair (broad_phase_common.py:220-240)
    assert len(b2.finalize().shape_contact_pairs) == 0


def test_heterogeneous_worlds_go_through_world_groups():
    """Worlds that differ in topology (model.py:881-900) finalize into a model served through its world groups
    (newton_amd/hetero.py); a single device descriptor for it is still refused loudly."""
    a = nt.ModelBuilder()
    a.add_shape_sphere(a.add_body())
    b = nt.ModelBuilder()
    b.add_shape_box(b.add_body())
    scene = nt.ModelBuilder()
    scene.add_world(a)
    scene.add_world(b)
    model = scene.finalize()
    assert model.is_heterogeneous and model.env is None
    assert model.world_groups.ranges == [(0, 1), (1, 2)]
    model.device = "cuda:0"  # pretend: the refusal must come before any device work
    with pytest.raises(NotImplementedError, match="heterogeneous worlds"):
        model.device_model()


def test_mixed_scene_template():
    model = mixed_primitive_scene(3)
    t = model.env
    assert t.nb == 7 and t.ns == 7 and t.ng == 1
    assert t.np == 7 + 21  # ground pairs + all local pairs
    assert t.cpp == 5      # box-capsule etc. are routed to the convex path


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/*.h declare."""
    inc = os.path.join(ROOT, "include")
    header = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    declared = set(re.findall(r"\b(nt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert b"gfx950" in lib.nt_build_info()
    assert lib.nt_error_string(0) == b"ok"
    # struct layouts agree with the header (field count + size)
    assert C.sizeof(_lib.nt_model) == 17 * 4 + 4 + 32 * 8  # 17 int32 (+ 4 B alignment pad) and 32 pointers
    m = _lib.nt_model()
    m.nb, m.nj, m.np, m.ns = 13, 13, 13, 13
    m.nd, m.ntq, m.cpp, m.np_analytic = 18, 18, 4, 13
    # slot-major fields with odd strides (even component counts padded by one row):
    # persistent rows 1401 (state (7 + 7) * 13 = 182 + body 23 * 13 = 299 + joint 15 * 13 = 195 + dof 11 * 18 = 198 + shape 21 * 13 = 273
    # + control 54 + gravity 3 + derived 117 + per-pair live counts 13 + their exclusive prefix 14 + the compacted live-contact
    # list 13 * 4 = 52 + its atomic append counter 1)
    # + XPBD scratch max(collide 14 * 13 + 13 = 195 + staged candidates 19 * 13 + hit list 13 + 1 = 456, the forces 7 * 13 + 13 * 13 behind it = 716,
    #   joints 22 * 13 = 286, correction records 11 * 52 = 572); the restitution scratch (182 + 15-float records) only when enabled
    assert lib.nt_lds_bytes_per_env(C.byref(m)) == 4 * (1401 + 716)
    # Featherstone: generalized state 127 + COM/origin 78 + S 108 + I_s 468 + v/a/f/ft 312 + f_ext 78 = 1171,
    # + max(P 6*13*18 + H 18*18 = 1728, contact wrenches 780, collide scratch 429)
    m.nc, m.na, m.max_art_dofs = 19, 1, 18
    # (this layout has no live-contact list and no body-derived tile: 1401 - 53 - 117 persistent rows; the figure is its largest form --
    # per-environment parameters, dense mass-matrix region; the uniform-parameter tile of the tree mode needs 1231 - 965 + 1171 + 780)
    assert lib.nt_featherstone_lds_bytes_per_env(C.byref(m)) == 4 * (1231 + 1171 + 1728)


def test_no_silent_cpu_fallback():
    model = quadruped_scene(1, seed=None)
    with pytest.raises(_lib.NewtonHipError):
        nt.solvers.SolverXPBD(model)
    with pytest.raises(_lib.NewtonHipError):
        nt.CollisionPipeline(model)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under newton_amd/ may reference it."""
    pkg = os.path.join(ROOT, "newton_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "_lib.py" and "oracle" not in text.lower(), (dirpath, f)


def test_state_reset_masked_host():
    """State.reset(source, world_mask): only the selected worlds are overwritten (host path of the RL-style reset)."""
    model = quadruped_scene(4, seed=3)
    default = model.state()
    s = model.state()
    s.body_q = s.body_q + 1.0
    s.joint_q = s.joint_q + 2.0
    before = s.body_q.copy()
    s.reset(default, world_mask=[True, False, True, False, False])  # reference shape: world_count + 1
    q = s.body_q.reshape(4, -1)
    assert np.array_equal(q[0], default.body_q.reshape(4, -1)[0]) and np.array_equal(q[2], default.body_q.reshape(4, -1)[2])
    assert np.array_equal(q[1], before.reshape(4, -1)[1]) and np.array_equal(q[3], before.reshape(4, -1)[3])
    jq = s.joint_q.reshape(4, -1)
    assert np.array_equal(jq[0], default.joint_q.reshape(4, -1)[0]) and not np.array_equal(jq[1], default.joint_q.reshape(4, -1)[1])
    with pytest.raises(ValueError):
        s.reset(default, world_mask=[True, False])


def test_ground_plane_first_is_accepted():
    """The reference's examples call add_ground_plane() before the bodies as often as after: global shapes may sit in
    front of the env-local block (shape_local0), alone or replicated."""
    b = nt.ModelBuilder()
    b.add_ground_plane()
    for k in range(3):
        body = b.add_body(xform=[0, 0, 0.5 + k, 0, 0, 0, 1])
        b.add_shape_box(body, hx=0.5, hy=0.5, hz=0.5)
    m = b.finalize()
    t = m.env
    assert t.shape_local0 == 1 and t.ns == 3 and t.ng == 1 and list(t.gshape_id) == [0]
    assert t.np == 3 + 3
    env = nt.ModelBuilder()
    for k in range(2):
        body = env.add_body(xform=[0, 0, 0.5 + k, 0, 0, 0, 1])
        env.add_shape_sphere(body, radius=0.5)
    scene = nt.ModelBuilder()
    scene.add_ground_plane()
    scene.replicate(env, 4)
    m = scene.finalize()
    t = m.env
    assert t.shape_local0 == 1 and t.ns == 2 and t.ng == 1 and t.env_count == 4
    # Newton ids: ground 0 < locals, so the ground (template index ns + 0 = 2) is the pair's first shape
    assert sorted(zip(t.pair_a.tolist(), t.pair_b.tolist())) == [(0, 1), (2, 0), (2, 1)]


def test_articulation_view_host():
    """newton.selection.ArticulationView subset: per-world getters / masked setters / masked FK on a host model."""
    from newton_amd.selection import ArticulationView

    model = quadruped_scene(4, seed=2)
    view = ArticulationView(model, "*")
    assert view.count == 4 and view.joint_dof_count == 18 and view.joint_coord_count == 19 and view.link_count == 13
    assert view.is_floating_base
    s = model.state()
    assert view.get_root_transforms(s).shape == (4, 7) and view.get_link_transforms(s).shape == (4, 13, 7)
    q = view.get_dof_positions(s).copy()
    q[:, 7:] += 0.1
    before = s.body_q.copy()
    view.set_dof_positions(s, q, mask=[True, False, True, False])
    got = view.get_dof_positions(s)
    assert np.allclose(got[0, 7:], q[0, 7:]) and np.allclose(got[1, 7:], q[1, 7:] - 0.1)
    view.eval_fk(s, mask=[True, False, True, False])
    moved = np.abs(s.body_q - before).reshape(4, -1).max(axis=1)
    assert moved[0] > 1e-3 and moved[2] > 1e-3 and moved[1] == 0.0 and moved[3] == 0.0
    ctrl = model.control()
    view.set_dof_forces(ctrl, np.ones((4, 18), dtype=np.float32), mask=[False, True, False, False])
    assert np.array_equal(view.get_dof_forces(ctrl).sum(axis=1), [0.0, 18.0, 0.0, 0.0])


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Error behaviour of the boundary: malformed descriptors come back as negative nt_status codes (never a crash)
    before any launch is attempted, so this runs on a GPU-less host."""
    lib = _lib.load()
    NT_ERR_INVALID_ARG = -1
    m = _lib.nt_model()          # all zeros: env_count == 0
    s = _lib.nt_state()
    ctl = _lib.nt_control()
    ct = _lib.nt_contacts()
    xp = _lib.nt_xpbd_params(2, 0.7, 0.4, 0.0, 0.0, 0.8, 1, 0.0, 0)
    cp = _lib.nt_collide_params(0, 0)
    fp = _lib.nt_featherstone_params(0.05, 1.0)
    sp = _lib.nt_semi_implicit_params(0.05, 1.0, 1e4, 1e2)
    assert lib.nt_collide(C.byref(m), C.byref(s), C.byref(ct), C.byref(cp), None) == NT_ERR_INVALID_ARG
    assert lib.nt_xpbd_step(C.byref(m), C.byref(xp), C.byref(s), C.byref(s), C.byref(ctl), None, 1e-3, 0, None, None) == NT_ERR_INVALID_ARG
    assert lib.nt_xpbd_rollout(C.byref(m), C.byref(xp), C.byref(cp), C.byref(s), C.byref(s), C.byref(ctl), C.byref(ct), 1e-3, 4,
                               None) == NT_ERR_INVALID_ARG
    assert lib.nt_semi_implicit_step(C.byref(m), C.byref(sp), C.byref(s), C.byref(s), C.byref(ctl), None, 1e-3, 0, None) == NT_ERR_INVALID_ARG
    assert lib.nt_featherstone_step(C.byref(m), C.byref(fp), C.byref(s), C.byref(s), C.byref(ctl), None, 1e-3, 0, None) == NT_ERR_INVALID_ARG
    assert lib.nt_clear_forces(C.byref(m), C.byref(s), None) == NT_ERR_INVALID_ARG
    assert lib.nt_state_reset(C.byref(m), C.byref(s), C.byref(s), None, None) == NT_ERR_INVALID_ARG
    # a well-formed descriptor with an unaligned env_stride / bad cpp is rejected too
    m.env_count, m.env_stride, m.nb, m.cpp = 10, 10, 1, 4
    assert lib.nt_clear_forces(C.byref(m), C.byref(s), None) == NT_ERR_INVALID_ARG
    m.env_stride, m.cpp = 64, 7
    assert lib.nt_collide(C.byref(m), C.byref(s), C.byref(ct), C.byref(cp), None) == NT_ERR_INVALID_ARG
    assert lib.nt_error_string(NT_ERR_INVALID_ARG) == b"invalid argument"
    assert lib.nt_error_string(-3) == b"unsupported configuration"


def test_viewer_null_and_state_recorder(tmp_path):
    """ViewerNull frame accounting / benchmark result and the ViewerFile state recorder round trip
    (newton/_src/viewer/viewer_null.py, viewer_file.py:1176-1260,1479-1533)."""
    import newton_amd as nt
    from scenes import pendulum_scene

    v = nt.viewer.ViewerNull(num_frames=5, benchmark=True, benchmark_start_frame=2)
    n = 0
    while v.is_running():
        v.begin_frame(0.01 * n)
        v.log_scalar("x", n)
        v.end_frame()
        n += 1
    assert n == 5 and v.frame_count == 5
    res = v.benchmark_result()
    assert res["frames"] == 3 and res["elapsed"] > 0.0 and res["fps"] > 0.0
    assert nt.viewer.ViewerNull().benchmark_result() is None

    model = pendulum_scene(2)
    rec = nt.viewer.ViewerFile(str(tmp_path / "run.npz"))
    rec.set_model(model)
    states = []
    s = model.state()
    for k in range(4):
        s.body_q = np.asarray(s.body_q) + np.float32(0.01 * (k + 1))
        s.joint_q = np.asarray(s.joint_q) + np.float32(0.1)
        rec.begin_frame(0.5 * k)
        rec.log_state(s)
        rec.end_frame()
        states.append((np.array(s.body_q).copy(), np.array(s.joint_q).copy()))
    assert rec.get_frame_count() == 4 and rec.has_model()
    rec.close()  # auto-save
    back = nt.viewer.ViewerFile()
    back.load_recording(str(tmp_path / "run.npz"))
    assert back.get_frame_count() == 4
    t = model.state()
    back.load_state(t, 2)
    assert np.array_equal(np.asarray(t.body_q), states[2][0]) and np.array_equal(np.asarray(t.joint_q), states[2][1])
    import pytest as _pytest

    with _pytest.raises(IndexError):
        back.load_state(t, 9)
    ring = nt.viewer.ViewerFile(max_history_size=2)
    for k in range(5):
        ring.record(s)
    assert ring.get_frame_count() == 2


def test_speculative_contacts_are_rejected_loudly_and_none_is_accepted():
    """collide.py:1132,1239-1246: speculative_config=None (the default) disables speculative contacts -- accepted; a config object
    is refused before any device work (no silent fallback to plain contacts)."""
    import inspect

    sig = inspect.signature(nt.CollisionPipeline.__init__)
    assert sig.parameters["speculative_config"].default is None
    src = inspect.getsource(nt.CollisionPipeline.__init__)
    assert src.index("speculative_config is not None") < src.index("_BROAD_PHASES")  # refused before anything else is touched


def test_pmc_traffic_is_only_attached_to_the_code_object_it_was_measured_on(monkeypatch):
    """bench.measured_traffic trusts a counter record when it carries the loaded library's build id, or when the stepping
    translation unit (nt_kernels.hip + its headers + newton_hip.h + flags: the rollout kernel's code object) is byte-identical to
    the one of the measured build (`step_unit`); a library that is not the build of this tree, or an edited stepping unit, gets none."""
    import sys

    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    import bench

    assert bench.build_id().endswith("src " + g.source_hash())  # the loaded library is the build of this tree
    assert bench.step_unit_id() == g.step_unit_id()
    rec = bench.measured_traffic("quadruped", 4096)
    if rec is not None:
        assert rec["build_id"] == bench.build_id() or rec["step_unit"] == g.step_unit_id()
        assert 0.0 < rec["bytes_per_launch"] < 359342080.0  # below the algorithmic bytes: the rollout keeps its state in LDS
    monkeypatch.setattr(g, "step_unit_id", lambda: "0" * 12)  # an edited stepping unit
    rec = bench.measured_traffic("quadruped", 4096)
    assert rec is None or rec["build_id"] == bench.build_id()
    monkeypatch.setattr(g, "source_hash", lambda: "f" * 12)  # a library that does not belong to these sources
    assert bench.step_unit_id() is None


def test_triangle_mesh_shapes_route_to_the_vertex_leg_the_triangle_leg_or_are_refused():
    """ModelBuilder.add_shape_mesh (builder.py:7158-7198): mass properties like a convex hull's (inertia.py:726-757 treats MESH and
    CONVEX_MESH alike), local AABB / voxel grid / collision radius of the scaled vertices, the vertex table un-deduplicated (the
    vertex index is the contact fingerprint); a (MESH, infinite plane) pair leaves the tiles for the vertex leg
    (narrow_phase.py:618-631) while the tiles keep the mesh as a pre-computed-AABB shape; a (MESH, convex primitive) pair leaves them
    for the triangle leg (narrow_phase.py:633-638, pair kind 3); a finite plane or a second mesh without SDFs is refused at finalize."""
    hull = nt.Mesh.create_box(0.1, 0.08, 0.05)
    mesh = nt.Mesh(np.concatenate([hull.vertices] * 3), hull.indices)  # per-face vertices of a render mesh: every corner three times

    def scene(extra=None, plane=True, scale=(1.0, 1.0, 1.0)):
        env = nt.ModelBuilder()
        b = env.add_body(xform=[0, 0, 0.05, 0, 0, 0, 1])
        env.add_shape_mesh(b, mesh=mesh, scale=scale)
        if extra:
            extra(env)
        s = nt.ModelBuilder()
        s.replicate(env, 3)
        if plane:
            s.add_ground_plane()
        return s.finalize(device=None)

    m = scene(scale=(1.0, 2.0, 1.0))
    t = m.env
    assert t.np == 0 and t.sdf_pair.tolist() == [[0, 1]] and t.sdf_pair_mesh_plane.tolist() == [True] and not t.sdf_pair_hydro.any()
    assert t.shape_type.tolist() == [int(GeoType.MESH), int(GeoType.PLANE)]
    assert t.tile_shape_type.tolist() == [int(GeoType.CONVEX_MESH), int(GeoType.PLANE)]
    assert m.mesh_vertex_range.tolist() == [[0, 24]] * 3 + [[0, 0]] and m.mesh_vertices.shape == (24, 3)  # one shared, full table
    assert np.allclose(m.shape_collision_aabb_lower[0], [-0.1, -0.16, -0.05]) and np.allclose(m.shape_collision_aabb_upper[1], [0.1, 0.16, 0.05])
    assert int(np.prod(m._shape_voxel_resolution[0])) in range(60, 101) and m._shape_voxel_resolution[3].tolist() == [0, 0, 0]
    assert np.isclose(m.shape_collision_radius[0], 0.5 * np.linalg.norm([0.2, 0.32, 0.1]))
    vol = 0.2 * 0.32 * 0.1
    assert np.isclose(m.body_mass[0], 1000.0 * vol, rtol=1e-5)  # default density, the scaled box
    # non-uniform scale: the reference's own rule I_xx' = I_xx (sy^2 + sz^2) / 2 |sx sy sz| density (inertia.py:744-746), not the
    # exact tensor of the stretched box
    I0xx = (0.2 * 0.16 * 0.1) / 12.0 * (0.16 ** 2 + 0.1 ** 2)
    assert np.isclose(np.asarray(m.body_inertia[0]).reshape(3, 3)[0, 0], I0xx * (2.0 ** 2 + 1.0) / 2.0 * 2.0 * 1000.0, rtol=1e-4)
    assert nt.sdf_pipeline.model_has_sdf_pairs(m)
    # a sphere next to the mesh: (mesh, sphere) takes the triangle leg, (mesh, plane) the vertex leg, (sphere, plane) stays a tile pair
    ms = scene(lambda env: env.add_shape_sphere(env.add_body(xform=[0.3, 0, 0.05, 0, 0, 0, 1]), radius=0.05))
    assert ms.env.np == 1 and ms.env.sdf_pair.tolist() == [[0, 1], [0, 2]]
    assert ms.env.sdf_pair_mesh_tri.tolist() == [True, False] and ms.env.sdf_pair_mesh_plane.tolist() == [False, True]
    assert ms.mesh_triangle_range.tolist() == [[0, 12], [0, 0]] * 3 + [[0, 0]] and ms.mesh_indices.shape == (12, 3)
    assert int(ms.mesh_indices.max()) == 7  # wp.Mesh.indices as given (this mesh lists every corner three times and indexes the first copy)
    with pytest.raises(NotImplementedError, match="no analytic path"):  # a FINITE plane is a convex shape the triangle leg does not take
        scene(lambda env: env.add_shape(body=-1, type=GeoType.PLANE, scale=(1.0, 1.0, 0.0)), plane=False)
    assert len(scene(plane=False).env.sdf_pair) == 0  # nothing to collide with: no legs at all
    # env-range shards / world groups / tiled copies keep the per-shape vertex ranges (the vertex table itself is a shared asset)
    from newton_amd.worlds import slice_worlds, tile_worlds

    part, twice = slice_worlds(m, 1, 3), tile_worlds(m, 2)
    assert part.mesh_vertex_range.tolist() == [[0, 24]] * 2 + [[0, 0]] and part.mesh_vertices is m.mesh_vertices
    assert part.env.env_count == 2 and part.env.sdf_pair_mesh_plane.tolist() == [True]
    assert part.mesh_triangle_range.tolist() == [[0, 12]] * 2 + [[0, 0]] and part.mesh_indices is m.mesh_indices
    assert twice.mesh_vertex_range.shape == (7, 2) and twice.env.env_count == 6 and twice.env.sdf_pair_mesh_plane.tolist() == [True]


def test_create_box_with_duplicated_vertices_is_the_reference_table():
    """Mesh.create_box(duplicate_vertices=True) = create_mesh_box of the reference, vertex for vertex and triangle for triangle
    (tests/golden/mesh_box_tables.json, recorded by executing newton/_src/utils/mesh.py:2034-2131); outward winding: the solid
    mass properties are the box's."""
    import json

    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "mesh_box_tables.json")))["duplicated"]
    m = nt.Mesh.create_box(0.5, 0.25, 0.125, duplicate_vertices=True)
    assert np.array_equal(m.vertices, np.asarray(ref["positions"], np.float32))
    assert m.indices.tolist() == ref["indices"]
    assert np.isclose(m.mass, 1.0 * 0.5 * 0.25) and np.allclose(m.com, 0.0, atol=1e-7)  # unit density x (2 hx)(2 hy)(2 hz)
    assert np.isclose(m.inertia[0, 0], m.mass / 12.0 * (0.5 ** 2 + 0.25 ** 2), rtol=1e-5)
    assert len(nt.Mesh.create_box(0.5, 0.25, 0.125).vertices) == 8  # the default: this package's 8-corner hull
    # reference_layout=True: the reference's shared-vertex box and its (lat + 1) x (lon + 1) sphere grid, vertex for vertex
    tables = json.load(open(os.path.join(ROOT, "tests", "golden", "mesh_box_tables.json")))
    m = nt.Mesh.create_box(0.5, 0.25, 0.125, reference_layout=True)
    assert np.array_equal(np.asarray(m.vertices, np.float32), np.asarray(tables["shared"]["positions"], np.float32))
    assert np.array_equal(np.asarray(m.indices).reshape(-1), np.asarray(tables["shared"]["indices"]))
    m = nt.Mesh.create_sphere(0.5, 4, 6, reference_layout=True, compute_inertia=False)
    assert np.array_equal(np.asarray(m.vertices, np.float32), np.asarray(tables["sphere_4x6"]["positions"], np.float32))
    assert np.array_equal(np.asarray(m.indices).reshape(-1), np.asarray(tables["sphere_4x6"]["indices"]))


def test_outlier_explainer_needs_substeps_after_the_first_divergence():
    """tests/tolerances.py:explain_rollout_outliers on synthetic trajectories: an environment that parts from the checker in the
    LAST substep of the trajectories cannot be verified to re-join and is reported as unexplained (it used to pass as class B
    unchecked: ADVICE round 5); with substeps past the compared frame the same event is classified by an actual re-join check."""
    import pytest as _pytest

    import tolerances as tol

    nb, E, N = 2, 3, 4
    base = np.zeros((E * nb, 7))
    base[:, 0] = np.arange(E * nb) + 1.0
    base[:, 6] = 1.0
    qd = np.zeros((E * nb, 6))

    def traj(n_sub, jump_env=None, jump_at=None, jump=0.0):
        out = []
        for k in range(n_sub + 1):
            q = base.copy()
            q[:, 0] += 1e-3 * k
            if jump_env is not None and k >= jump_at:
                q[jump_env * nb, 0] += jump
            out.append((q, qd.copy()))
        return out

    # the device jumps by 1e-3 in env 1 at substep N (a threshold event inside that substep: class B).  The checker never makes the jump
    # itself; restarted from a state that already carries it, it carries it on
    def restart(n_sub):
        def f(k0, q, _qd):
            carried = q[nb, 0] - (base[nb, 0] + 1e-3 * k0)
            out = []
            for k in range(k0 + 1, n_sub + 1):
                r = base.copy()
                r[:, 0] += 1e-3 * k
                r[nb, 0] += carried
                out.append((r, qd.copy()))
            return out
        return f

    gpu, ora = traj(N, 1, N, 1e-3), traj(N)
    with _pytest.raises(AssertionError, match="do not re-join"):  # first divergence in the last substep: nothing left to re-join in
        tol.explain_rollout_outliers("synthetic", gpu, ora, restart(N), nb)
    gpu, ora = traj(N + 2, 1, N, 1e-3), traj(N + 2)
    res = tol.explain_rollout_outliers("synthetic", gpu, ora, restart(N + 2), nb, frame=N)
    assert res["outliers"] == 1 and res["class_a"] == 0 and res["class_b"] == 1 and not res["unexplained"]
    # a device that keeps drifting after the event does not re-join from identical states: a real discrepancy
    gpu = traj(N + 2, 1, N, 1e-3)
    gpu[N + 2][0][nb, 0] += 1e-3
    with _pytest.raises(AssertionError, match="do not re-join"):
        tol.explain_rollout_outliers("synthetic", gpu, ora, restart(N + 2), nb, frame=N)


def test_heightfield_shapes_route_to_the_triangle_leg():
    """ModelBuilder.add_shape_heightfield (geometry/types.py:2240-2340, builder.py:11653-11660): a static GeoType.HFIELD shape with the
    field's bounding box as local AABB (the tiles keep it as a pre-computed-AABB shape), HeightfieldData + normalised elevations in
    the model, (heightfield, convex) pairs routed to pair kind 3 (narrow_phase.py:553-583); primitives carry their local AABBs
    (builder.py:11601-11652: what the heightfield midphase reads of the partner)."""
    xs, ys = np.linspace(-1.0, 1.0, 21), np.linspace(-0.8, 0.8, 17)
    raw = np.array([[0.5 + 0.03 * np.sin(4 * x) * np.cos(3 * y) for x in xs] for y in ys], np.float32)
    field = nt.Heightfield(raw, 17, 21, hx=1.0, hy=0.8)
    assert field.data.min() == 0.0 and field.data.max() == 1.0 and np.isclose(field.min_z, raw.min()) and np.isclose(field.max_z, raw.max())
    env = nt.ModelBuilder()
    env.add_shape_box(env.add_body(xform=[0, 0, 0.6, 0, 0, 0, 1]), hx=0.05, hy=0.04, hz=0.03)
    env.add_shape_capsule(env.add_body(xform=[0.3, 0, 0.6, 0, 0, 0, 1]), radius=0.02, half_height=0.05)
    s = nt.ModelBuilder()
    s.replicate(env, 3)
    s.add_shape_heightfield(heightfield=field)
    m = s.finalize(device=None)
    t = m.env
    assert t.np == 1 and t.sdf_pair.tolist() == [[0, 2], [1, 2]] and t.sdf_pair_mesh_tri.tolist() == [True, True]
    assert t.shape_type.tolist() == [int(GeoType.BOX), int(GeoType.CAPSULE), int(GeoType.HFIELD)]
    assert t.tile_shape_type.tolist() == [int(GeoType.BOX), int(GeoType.CAPSULE), int(GeoType.CONVEX_MESH)]
    assert m.shape_heightfield_index.tolist() == [-1] * 6 + [0] and m.heightfield_count == 1
    off, nrow, ncol, hx, hy, zlo, zhi = m.heightfield_data[0]
    assert (off, nrow, ncol, hx, hy) == (0, 17, 21, 1.0, 0.8) and m.heightfield_elevations.shape == (17 * 21,)
    assert np.allclose(m.shape_collision_aabb_lower[6], [-1.0, -0.8, zlo]) and np.allclose(m.shape_collision_aabb_upper[6], [1.0, 0.8, zhi])
    assert int(np.prod(m._shape_voxel_resolution[6])) in range(50, 101)
    assert np.allclose(m.shape_collision_aabb_upper[0], [0.05, 0.04, 0.03]) and np.allclose(m.shape_collision_aabb_upper[1], [0.02, 0.02, 0.07])
    assert int(m.shape_body[6]) == -1
    with pytest.raises(NotImplementedError, match="scale"):
        nt.ModelBuilder().add_shape_heightfield(heightfield=field, scale=(2.0, 1.0, 1.0))
    from newton_amd.worlds import slice_worlds

    part = slice_worlds(m, 1, 3)
    assert part.shape_heightfield_index.tolist() == [-1] * 4 + [0] and part.env.sdf_pair_mesh_tri.tolist() == [True, True]
