"""Triangle meshes on an infinite ground plane THROUGH CollisionPipeline.collide (SURVEY.md section 8 row a24's mesh legs;
narrow_phase.py:618-631 routing, :1744-1992 vertex kernels, contact_reduction_global.py:1246-1346 the buffered reducer): the rows
of the vertex leg (csrc/nt_mesh_plane.hip as pair kind 2 of the SDF leg) against the checker chain -- oracle_mesh_plane (pinned by
the executed reference, tests/test_mesh_plane.py) + write_contact -- then the solvers consuming them."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

pytestmark = pytest.mark.gpu
FIELDS = ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")


def _rows(contacts):
    f = contacts._flat
    n = int(f.row_start[-1].item())
    d = {k: getattr(f, k)[:n].cpu().numpy() for k in (*FIELDS, "key")}
    d["row_start"] = f.row_start.cpu().numpy()
    return d


def mesh_scene(worlds, kind="box", device="cuda:0", ground_first=False, margin=0.0, gap=0.004, tilt=0.01, seed=3):
    """One free body per world carrying a triangle mesh (a box with per-face vertices = three coincident vertices per corner, or a
    UV sphere), resting slightly inside the ground plane with a small per-world tilt."""
    import newton_amd as nt

    rng = np.random.default_rng(seed)
    if kind == "box":
        hull = nt.Mesh.create_box(0.1, 0.08, 0.05)  # every corner three times (per-face vertices of a render mesh): roundoff twins
        mesh, h = nt.Mesh(np.concatenate([hull.vertices] * 3), hull.indices), 0.05
    else:
        mesh, h = nt.Mesh.create_sphere(0.08, 12, 16), 0.08
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = gap
    env.default_shape_cfg.margin = margin
    b = env.add_body(xform=[0.0, 0.0, h - 0.0008, 0.0, 0.0, 0.0, 1.0])
    env.add_shape_mesh(b, mesh=mesh)
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = gap
    if ground_first:
        scene.add_ground_plane()
    scene.replicate(env, worlds)
    if not ground_first:
        scene.add_ground_plane()
    model = scene.finalize(device=device)
    for w in range(worlds):  # per-world pose: small tilt + offset (the worlds then differ in which vertices touch)
        q = nt._np_math.quat_rpy(*(rng.uniform(-tilt, tilt, size=2)), rng.uniform(-1.0, 1.0))
        model.body_q[w, 3:] = q
        model.body_q[w, :2] = rng.uniform(-0.3, 0.3, size=2)
        model.joint_q.reshape(-1, 7)[w] = model.body_q[w]
    return model


def checker_rows(model, leg, body_q):
    """The rows the pipeline must emit, world-major, from the device's own exported shape transforms (oracle_mesh_plane ->
    oracle_flat_contacts.write_rows)."""
    import oracle_flat_contacts as F
    import oracle_mesh_plane as omp

    t = model.env
    X = leg.world_xform.cpu().numpy()
    data = np.concatenate([np.asarray(model.shape_scale, np.float32), np.asarray(model.shape_margin, np.float32)[:, None]], axis=1)
    s = dict(shape_transform=X, shape_data=data, shape_gap=np.asarray(model.shape_gap, np.float32),
             aabb_lo=np.asarray(model.shape_collision_aabb_lower, np.float32), aabb_hi=np.asarray(model.shape_collision_aabb_upper, np.float32),
             res=np.asarray(model._shape_voxel_resolution, np.int32), vertex_start=model.mesh_vertex_range[:, 0],
             vertex_count=model.mesh_vertex_range[:, 1], vertices=model.mesh_vertices)

    def gid(l, w):
        return t.shape_local0 + w * t.ns + l if l < t.ns else int(t.gshape_id[l - t.ns])

    out = {k: [] for k in ("world", "key", *FIELDS)}
    for w in range(t.env_count):
        for (a, b), mp in zip(t.sdf_pair, t.sdf_pair_mesh_plane):
            assert mp
            ga, gb = gid(int(a), w), gid(int(b), w)
            mesh, plane = (ga, gb) if s["vertex_count"][ga] > 0 else (gb, ga)
            red = omp.mesh_plane_rows(dict(s, pairs=np.array([[mesh, plane]], np.int32)))
            n = len(red["fp"])
            if n == 0:
                continue
            raw = dict(key=red["fp"], shape_a=red["pair"][:, 0], shape_b=red["pair"][:, 1], center=red["pos"], normal=red["normal"],
                       distance=red["depth"], margin_a=red["margin_a"], margin_b=red["margin_b"])
            wr = F.write_rows(raw, np.asarray(body_q, np.float32), np.asarray(model.shape_body), s["shape_gap"])
            out["world"] += [w] * n
            out["key"] += red["fp"].tolist()
            for name in FIELDS:
                out[name] += list(wr[name])
    return {k: np.asarray(v) for k, v in out.items()}


@pytest.mark.parametrize("kind,ground_first,margin", [("box", False, 0.0), ("sphere", True, 0.001)])
def test_collide_rows_of_meshes_on_the_ground_match_the_checker(kind, ground_first, margin):
    import newton_amd as nt

    E = 6
    model = mesh_scene(E, kind, ground_first=ground_first, margin=margin)
    t = model.env
    assert t.np == 0 and len(t.sdf_pair) == 1 and bool(t.sdf_pair_mesh_plane.all())  # the one pair takes the vertex leg
    assert int(t.tile_shape_type[0]) == int(nt.GeoType.CONVEX_MESH) and int(t.shape_type[0]) == int(nt.GeoType.MESH)
    pipe = nt.CollisionPipeline(model, broad_phase="nxn")
    c1, c2 = pipe.contacts(), pipe.contacts()
    state = model.state()
    pipe.collide(state, c1)
    pipe.collide(state, c2)
    got, again = _rows(c1), _rows(c2)
    for k in got:  # two collide() calls: bit-identical rows
        assert np.array_equal(got[k], again[k]), k
    leg = pipe._sdf_leg
    assert not leg.overflow(c1._flat)["overflow"]
    want = checker_rows(model, leg, model.body_q)
    assert len(want["key"]) == len(got["key"]) > 3 * E
    assert np.array_equal(got["row_start"], np.concatenate([[0], np.cumsum(np.bincount(want["world"], minlength=E))]))
    assert np.array_equal(got["key"], want["key"])
    for k in ("shape0", "shape1"):
        assert np.array_equal(got[k], want[k]), k
    plane = model.shape_count - 1 if not ground_first else 0
    assert np.all(got["shape1"] == plane) and np.all(got["shape0"] != plane)  # (mesh, plane) whatever the id order
    for k in FIELDS[2:]:  # same shape transforms in -> same rows out
        assert np.abs(got[k] - want[k]).max() <= 2e-6, (k, np.abs(got[k] - want[k]).max())
    # the world AABB of the mesh the tiles exported = its local AABB rotated (compute_shape_aabbs' pre-computed-AABB branch): it
    # must contain every vertex; and Newton's flat view lists the rows
    X = leg.world_xform.cpu().numpy()
    lo, hi = leg.aabb_lower.cpu().numpy(), leg.aabb_upper.cpu().numpy()
    import oracle_mesh_plane as omp

    for w in (0, E - 1):
        sid = t.shape_local0 + w * t.ns
        v = np.array([omp.transform_point(X[sid], p * np.asarray(model.shape_scale[sid], np.float32)) for p in model.mesh_vertices])
        assert np.all(v >= lo[sid] - 1e-6) and np.all(v <= hi[sid] + 1e-6)
        assert np.all(hi[sid] - lo[sid] < 0.5)
    assert int(c1.rigid_contact_count.item()) == int((got["shape0"] != got["shape1"]).sum())
    # reduce_contacts=False: every vertex within margin + gap is a row
    pipe2 = nt.CollisionPipeline(model, broad_phase="nxn", reduce_contacts=False)
    c3 = pipe2.contacts()
    pipe2.collide(state, c3)
    assert int(c3._flat.row_start[-1].item()) > len(got["key"])


def test_mesh_boxes_settle_on_the_ground_under_every_solver():
    """The rows feed the solvers like any SDF row: a mesh box dropped from 2 mm comes to rest on its bottom face."""
    import newton_amd as nt

    E = 4
    for make, dt, steps in ((lambda m: nt.solvers.SolverXPBD(m, iterations=4), 1.0 / 600.0, 300),
                            (lambda m: nt.solvers.SolverSemiImplicit(m), 1.0 / 4000.0, 1200)):
        model = mesh_scene(E, "box", tilt=0.0, gap=0.002)
        model.body_q[:, 2] += 0.002
        model.joint_q.reshape(-1, 7)[:, 2] += 0.002
        for k in ("ke", "kd", "kf", "mu"):
            getattr(model, "shape_material_" + k)[:] = {"ke": 2.0e4, "kd": 50.0, "kf": 50.0, "mu": 0.5}[k]
        solver = make(model)
        pipe = nt.CollisionPipeline(model, broad_phase="nxn")
        contacts = pipe.contacts()
        s0, s1 = model.state(), model.state()
        for _ in range(steps):
            s0.clear_forces()
            pipe.collide(s0, contacts)
            solver.step(s0, s1, None, contacts, dt)
            s0, s1 = s1, s0
        q, qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
        assert np.all(np.isfinite(q)) and np.all(np.abs(q[:, 2] - 0.05) < 2e-3), (type(solver).__name__, q[:, 2])
        assert np.abs(qd).max() < 0.05, (type(solver).__name__, np.abs(qd).max())


def test_mesh_box_on_ground_like_the_reference_test():
    """newton/tests/test_rigid_contact.py:512-605 (test_mesh_box_on_ground), call for call: ground plane first, a unit mesh box
    (Mesh.create_box, compute_inertia=False -> mass properties from the triangles) resting with its bottom face on z = 0, SolverXPBD
    iterations=2, CollisionPipeline defaults, 60 frames of 10 substeps at 1/600 s: stays at z ~ 0.5 with every velocity below 0.01."""
    import newton_amd as nt

    builder = nt.ModelBuilder()
    builder.default_shape_cfg.ke = 1.0e5
    builder.default_shape_cfg.kd = 1.0e3
    builder.default_shape_cfg.mu = 0.5
    builder.add_ground_plane()
    box_half = 0.5
    box_mesh = nt.Mesh.create_box(box_half, box_half, box_half, duplicate_vertices=False, compute_normals=False, compute_uvs=False,
                                  compute_inertia=False)
    body = builder.add_body(xform=[0.0, 0.0, box_half, 0.0, 0.0, 0.0, 1.0])
    builder.add_shape_mesh(body=body, mesh=box_mesh)
    model = builder.finalize(device="cuda:0")
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    state_0, state_1 = model.state(), model.state()
    control = model.control()
    pipeline = nt.CollisionPipeline(model)
    contacts = pipeline.contacts()
    nt.eval_fk(model, model.joint_q, model.joint_qd, state_0)
    sim_dt, substeps = 1.0 / 60.0, 10
    for _ in range(60):
        for _ in range(substeps):
            state_0.clear_forces()
            pipeline.collide(state_0, contacts)
            solver.step(state_0, state_1, control, contacts, sim_dt / substeps)
            state_0, state_1 = state_1, state_0
    q, qd = state_0.body_q.cpu().numpy()[body], state_0.body_qd.cpu().numpy()[body]
    assert box_half * 0.9 < q[2] < box_half * 1.1, q
    assert np.all(np.abs(qd) < 0.01), qd
    assert int(contacts.rigid_contact_count.item()) == 4  # the four bottom corners


def test_contact_matching_and_deterministic_order_over_vertex_rows():
    """contact_matching="latest" + contact_report on a model whose only pairs take the vertex leg (no slot contacts at all: the
    slot matcher has nothing to do, the row matcher carries the frame): first frame MATCH_NOT_FOUND, an unchanged second frame
    matches row for row with nothing new or broken; moving one world's body breaks / renews its rows only; deterministic=True
    keeps the (shape0, shape1, vertex) order."""
    import newton_amd as nt

    E = 3
    model = mesh_scene(E, "box")
    pipe = nt.CollisionPipeline(model, broad_phase="nxn", contact_matching="latest", contact_report=True)
    c = pipe.contacts()
    state = model.state()
    pipe.collide(state, c)
    n = int(c.rigid_contact_count.item())
    assert n >= 3 * E and np.all(c.rigid_contact_match_index[:n].cpu().numpy() == -1)
    pipe.collide(state, c)
    assert int(c.rigid_contact_count.item()) == n
    assert np.array_equal(c.rigid_contact_match_index[:n].cpu().numpy(), np.arange(n))
    assert int(c.rigid_contact_new_count.item()) == 0 and int(c.rigid_contact_broken_count.item()) == 0
    s0, s1 = c.rigid_contact_shape0[:n].cpu().numpy(), c.rigid_contact_shape1[:n].cpu().numpy()
    assert np.all(np.diff(s0) >= 0) and np.all(s1 == model.shape_count - 1)
    # world 1 moves sideways by more than the position threshold: its rows are new, the others still match
    q = state.body_q.cpu().numpy().copy()
    q[1, 0] += 0.05
    state.body_q = q  # (the State setter takes Newton's AoS array, host or device)
    pipe.collide(state, c)
    mi = c.rigid_contact_match_index[: int(c.rigid_contact_count.item())].cpu().numpy()
    s0 = c.rigid_contact_shape0[: len(mi)].cpu().numpy()
    assert np.all(mi[s0 == 1] < 0) and np.all(mi[s0 != 1] >= 0)
    assert int(c.rigid_contact_new_count.item()) == int((s0 == 1).sum())
    det = nt.CollisionPipeline(model, broad_phase="nxn", deterministic=True)
    cd = det.contacts()
    det.collide(state, cd)
    nd = int(cd.rigid_contact_count.item())
    key = cd.rigid_contact_shape0[:nd].cpu().numpy().astype(np.int64) * 10_000 + cd._flat.key[:nd].cpu().numpy()
    assert nd == len(mi) and np.all(np.diff(key) > 0)


def test_contact_matching_with_the_planes_before_and_after_the_meshes():
    """the vertex leg rewrites a candidate pair as (mesh, plane); with a ground plane added FIRST (shape 0) and a
    second plane added last, a world's candidate list [(0, 1), (0, 2), (1, 3), (2, 3)] becomes [(1, 0), (2, 0), (1, 3), (2, 3)] --
    no longer ascending in (shape0, shape1).  The row matcher searches the previous frame's list on the canonical (min, max) key, so an
    unchanged second frame still matches every row."""
    import newton_amd as nt

    hull = nt.Mesh.create_box(0.1, 0.08, 0.05)
    mesh = nt.Mesh(np.concatenate([hull.vertices] * 3), hull.indices)
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = 0.004
    b = env.add_body(xform=[0.0, 0.0, 0.05 - 0.0008, 0.0, 0.0, 0.0, 1.0])
    env.add_shape_mesh(b, mesh=mesh)
    env.add_shape_mesh(b, mesh=mesh, xform=[0.3, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    env.add_shape_collision_filter_pair(0, 1)
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = 0.004
    scene.add_ground_plane()
    scene.replicate(env, 2)
    scene.add_ground_plane()
    model = scene.finalize(device="cuda:0")
    pipe = nt.CollisionPipeline(model, broad_phase="nxn", contact_matching="latest", contact_report=True)
    c = pipe.contacts()
    state = model.state()
    pipe.collide(state, c)
    n = int(c.rigid_contact_count.item())
    assert n >= 2 * 2 * 2 * 4  # two worlds x two meshes x two planes x four bottom corners
    assert np.all(c.rigid_contact_match_index[:n].cpu().numpy() == -1)
    pipe.collide(state, c)
    assert int(c.rigid_contact_count.item()) == n
    assert np.array_equal(c.rigid_contact_match_index[:n].cpu().numpy(), np.arange(n))
    assert int(c.rigid_contact_new_count.item()) == 0 and int(c.rigid_contact_broken_count.item()) == 0


def test_mesh_worlds_inside_heterogeneous_models():
    """Worlds that differ in topology (a mesh box here, a mesh sphere plus a primitive box there) run as world groups
    (newton_amd/hetero.py): every group has its own vertex leg, the flat contact arrays keep Newton's shape ids -- slot contacts of
    the primitive boxes first, then the mesh rows as (mesh, plane) -- and the bodies stay on the ground under XPBD."""
    import newton_amd as nt

    hull = nt.Mesh.create_box(0.1, 0.08, 0.05)
    box_mesh = nt.Mesh(np.concatenate([hull.vertices] * 3), hull.indices)
    sphere_mesh = nt.Mesh.create_sphere(0.08, 8, 10)

    def env(mesh, h, extra=False):
        e = nt.ModelBuilder()
        e.default_shape_cfg.gap = 0.004
        b = e.add_body(xform=[0, 0, h - 0.0008, 0, 0, 0, 1])
        e.add_shape_mesh(b, mesh=mesh)
        if extra:
            b2 = e.add_body(xform=[0.5, 0, 0.05, 0, 0, 0, 1])
            e.add_shape_box(b2, hx=0.05, hy=0.05, hz=0.05)
            e.add_shape_collision_filter_pair(0, 1)  # (a mesh-vs-box pair would need the triangle leg)
        return e

    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = 0.004
    for k in range(4):
        scene.add_world(env(box_mesh, 0.05) if k % 2 == 0 else env(sphere_mesh, 0.08, extra=True))
    scene.add_ground_plane()
    model = scene.finalize(device="cuda:0")
    plane = model.shape_count - 1
    meshes = [s for s in range(model.shape_count) if int(model.shape_type[s]) == int(nt.GeoType.MESH)]
    assert meshes == [0, 1, 3, 4]
    pipe = nt.CollisionPipeline(model, broad_phase="nxn")
    c = pipe.contacts()
    s0, s1 = model.state(), model.state()
    pipe.collide(s0, c)
    n = int(c.rigid_contact_count.item())
    a, b = c.rigid_contact_shape0[:n].cpu().numpy(), c.rigid_contact_shape1[:n].cpu().numpy()
    slot = np.isin(b, [2, 5])  # the primitive boxes: (plane, box) slot contacts, four corners each
    assert slot.sum() == 8 and np.all(a[slot] == plane) and np.all(np.flatnonzero(slot) < 8)
    assert np.all(b[~slot] == plane) and sorted(set(a[~slot].tolist())) == meshes  # every mesh touches the ground, as (mesh, plane)
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    for _ in range(60):
        s0.clear_forces()
        pipe.collide(s0, c)
        solver.step(s0, s1, None, c, 1.0 / 600.0)
        s0, s1 = s1, s0
    z = s0.body_q.cpu().numpy()[:, 2]
    assert np.all(np.abs(z - np.array([0.05, 0.08, 0.05, 0.05, 0.08, 0.05])) < 3e-3), z
