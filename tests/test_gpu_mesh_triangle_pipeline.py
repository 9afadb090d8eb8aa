"""Convex primitives on a static triangle-mesh terrain THROUGH CollisionPipeline.collide (SURVEY.md section 8 rows a19 / a20 / a23
on triangle meshes; narrow_phase.py:633-638 routing, :1455-1665 midphase + per-triangle GJK / MPR, contact_reduction_global.py
:2299-2403 the reducer path): the rows of the triangle leg (csrc/nt_mesh_triangle.hip as pair kind 3 of the SDF leg) against the
checker chain -- oracle_mesh_triangle (pinned by the executed reference, tests/test_mesh_triangle.py) + write_contact -- then
SolverXPBD consuming them."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

pytestmark = pytest.mark.gpu
FIELDS = ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")


def _rows(contacts):
    f = contacts._flat
    n = int(f.row_start[-1].item())
    d = {k: getattr(f, k)[:n].cpu().numpy() for k in (*FIELDS, "key")}
    d["row_start"] = f.row_start.cpu().numpy()
    return d


def terrain_height(x, y):
    return 0.02 * np.sin(5.0 * x) * np.cos(4.0 * y)


def terrain_scene(worlds, kinds=("box", "sphere", "capsule", "hull"), device="cuda:0", gap=0.004, margin=0.0, seed=5, drop=0.0):
    """Every world: one free body per kind resting slightly inside a shared static terrain mesh (a global shape)."""
    import mesh_triangle_cases as mc
    import newton_amd as nt

    rng = np.random.default_rng(seed)
    p, t = mc.grid_mesh(32, 32, 1.6, 1.6, height=terrain_height)
    terrain = nt.Mesh(p, t.reshape(-1))
    half = dict(box=0.05, sphere=0.06, capsule=0.04, cylinder=0.05, hull=0.04)
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = gap
    env.default_shape_cfg.margin = margin
    for k, kind in enumerate(kinds):
        b = env.add_body(xform=[0.4 * k - 0.4, 0.0, 0.2, 0.0, 0.0, 0.0, 1.0])
        if kind == "box":
            env.add_shape_box(b, hx=0.08, hy=0.06, hz=0.05)
        elif kind == "sphere":
            env.add_shape_sphere(b, radius=0.06)
        elif kind == "capsule":
            env.add_shape_capsule(b, radius=0.04, half_height=0.08)
        elif kind == "hull":  # a convex hull (CONVEX_MESH): a wedge-topped box of 10 vertices
            pts = np.array([(sx * 0.07, sy * 0.05, sz * 0.04) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)] +
                           [(0.0, -0.03, 0.06), (0.0, 0.03, 0.06)], np.float32)
            env.add_shape_convex_hull(b, mesh=nt.Mesh.convex_hull_of(pts))
        else:
            env.add_shape_cylinder(b, radius=0.05, half_height=0.05)
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = gap
    scene.default_shape_cfg.margin = margin
    scene.replicate(env, worlds)
    scene.add_shape_mesh(-1, mesh=terrain)
    model = scene.finalize(device=device)
    nbody = len(kinds)
    for w in range(worlds):
        for k, kind in enumerate(kinds):
            x, y = 0.5 * k - 0.75 + rng.uniform(-0.1, 0.1), rng.uniform(-0.4, 0.4)
            q = nt._np_math.quat_rpy(*(rng.uniform(-0.05, 0.05, size=2)), rng.uniform(-1.0, 1.0))
            if kind == "capsule":  # lying on its side
                q = nt._np_math.quat_rpy(0.0, np.pi / 2 + rng.uniform(-0.03, 0.03), rng.uniform(-1.0, 1.0))
            i = w * nbody + k
            model.body_q[i, :3] = [x, y, terrain_height(x, y) + half[kind] - 0.001 + drop]
            model.body_q[i, 3:] = q
            model.joint_q.reshape(-1, 7)[i] = model.body_q[i]
    return model


def checker_rows(model, leg, body_q, worlds=None):
    """The rows the pipeline must emit, world-major, from the device's own exported shape transforms (oracle_mesh_triangle ->
    oracle_flat_contacts.write_rows)."""
    import oracle_flat_contacts as F
    import oracle_mesh_triangle as om

    t = model.env
    X = leg.world_xform.cpu().numpy()
    data = np.concatenate([np.asarray(model.shape_scale, np.float32), np.asarray(model.shape_margin, np.float32)[:, None]], axis=1)
    s = dict(shape_type=np.asarray(model.shape_type, np.int32), shape_transform=X, shape_data=data,
             shape_gap=np.asarray(model.shape_gap, np.float32),
             aabb_lo=np.asarray(model.shape_collision_aabb_lower, np.float32), aabb_hi=np.asarray(model.shape_collision_aabb_upper, np.float32),
             res=np.asarray(model._shape_voxel_resolution, np.int32), vertex_start=model.mesh_vertex_range[:, 0],
             vertex_count=model.mesh_vertex_range[:, 1], tri_start=model.mesh_triangle_range[:, 0],
             tri_count=model.mesh_triangle_range[:, 1], vertices=model.mesh_vertices, indices=model.mesh_indices,
             hull_start=np.asarray(model.shape_mesh_start, np.int32),
             hull_count=np.where(np.asarray(model.shape_type) == 10, np.asarray(model.shape_mesh_count), 0).astype(np.int32),
             hull_points=np.asarray(model.mesh_points, np.float32),
             hf_index=np.asarray(model.shape_heightfield_index, np.int32),
             hf_table=np.asarray(model.heightfield_data, np.float32).reshape(-1, 7), hf_elev=np.asarray(model.heightfield_elevations, np.float32))

    def gid(l, w):
        return t.shape_local0 + w * t.ns + l if l < t.ns else int(t.gshape_id[l - t.ns])

    out = {k: [] for k in ("world", "key", *FIELDS)}
    lo, hi = leg.aabb_lower.cpu().numpy(), leg.aabb_upper.cpu().numpy()
    for w in (range(t.env_count) if worlds is None else worlds):
        for (a, b), mt in zip(t.sdf_pair, t.sdf_pair_mesh_tri):
            assert mt
            ga, gb = gid(int(a), w), gid(int(b), w)
            if np.any(lo[ga] > hi[gb]) or np.any(lo[gb] > hi[ga]):  # not a broad-phase candidate
                continue
            red = om.mesh_triangle_rows(dict(s, pairs=np.array([[ga, gb]], np.int32)))
            n = len(red["fp"])
            if n == 0:
                continue
            raw = dict(key=red["fp"], shape_a=red["pair"][:, 0], shape_b=red["pair"][:, 1], center=red["pos"], normal=red["normal"],
                       distance=red["depth"], margin_a=red["margin_a"], margin_b=red["margin_b"], radius_a=red["radius_a"],
                       radius_b=red["radius_b"])
            wr = F.write_rows(raw, np.asarray(body_q, np.float32), np.asarray(model.shape_body), s["shape_gap"])
            out["world"] += [w] * n
            out["key"] += red["fp"].tolist()
            for name in FIELDS:
                out[name] += list(wr[name])
    return {k: np.asarray(v) for k, v in out.items()}


@pytest.mark.parametrize("margin", [0.0, 0.001])
def test_collide_rows_of_primitives_on_a_terrain_match_the_checker(margin):
    import newton_amd as nt

    E = 5
    model = terrain_scene(E, margin=margin)
    t = model.env
    assert len(t.sdf_pair) == 4 and bool(t.sdf_pair_mesh_tri.all())  # (box | sphere | capsule | hull, terrain): the triangle leg
    assert t.np == 6  # the six pairs among the four convex shapes stay in the tiles
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
    assert len(want["key"]) == len(got["key"]) > 4 * E
    assert np.array_equal(got["row_start"], np.concatenate([[0], np.cumsum(np.bincount(want["world"], minlength=E))]))
    assert np.array_equal(got["key"], want["key"])
    for k in ("shape0", "shape1"):
        assert np.array_equal(got[k], want[k]), k
    terrain = model.shape_count - 1
    live = got["shape0"] >= 0
    assert live.sum() >= 4 * E and np.all(got["shape0"][live] == terrain) and np.all(got["shape1"][live] != terrain)  # (mesh, convex)
    for k in FIELDS[2:]:  # same shape transforms in -> same rows out
        assert np.abs(got[k] - want[k]).max() <= 2e-6, (k, np.abs(got[k] - want[k]).max())
    # the sphere / capsule rows carry the effective radius in their margins (write_contact: offset magnitude = radius + margin)
    stype = np.asarray(model.shape_type)
    sph = live & (stype[np.maximum(got["shape1"], 0)] == int(nt.GeoType.SPHERE))
    assert sph.any() and np.allclose(got["margin1"][sph], 0.06 + margin, atol=1e-7) and np.allclose(got["margin0"][sph], margin, atol=1e-7)
    assert int(c1.rigid_contact_count.item()) >= int(live.sum())
    # reduce_contacts=False: every generated contact is a row (more rows than the reducer keeps)
    pipe2 = nt.CollisionPipeline(model, broad_phase="nxn", reduce_contacts=False)
    c3 = pipe2.contacts()
    pipe2.collide(state, c3)
    assert not pipe2._sdf_leg.overflow(c3._flat)["overflow"]
    full = _rows(c3)
    assert len(full["key"]) > len(got["key"])
    assert set(zip(np.repeat(np.arange(E), np.diff(got["row_start"])).tolist(), got["shape1"].tolist(), got["key"].tolist())) <= \
        set(zip(np.repeat(np.arange(E), np.diff(full["row_start"])).tolist(), full["shape1"].tolist(), full["key"].tolist())) | \
        {(w, -1, k) for w in range(E) for k in got["key"].tolist()}


def test_xpbd_settles_primitives_on_the_terrain():
    """Dropped from 2 cm: the bodies come to rest ON the mesh (no tunnelling, no drift), the rollout == the call-by-call loop."""
    import newton_amd as nt

    E = 16
    model = terrain_scene(E, kinds=("box", "sphere", "capsule", "cylinder", "hull"), drop=0.02)
    pipe = nt.CollisionPipeline(model, broad_phase="nxn")
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    s0, s1 = model.state(), model.state()
    dt = 1.0 / 600.0
    for _ in range(600):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, dt)
        s0, s1 = s1, s0
    q = s0.body_q.cpu().numpy().reshape(-1, 7)
    qd = s0.body_qd.cpu().numpy().reshape(-1, 6)
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd))
    ground = terrain_height(q[:, 0], q[:, 1])
    assert np.all(q[:, 2] > ground + 0.02), (q[:, 2] - ground).min()   # resting on the surface (smallest half extent 0.04)
    assert np.all(q[:, 2] < ground + 0.12)
    assert np.all(np.abs(q[:, :2]) < 1.55)                             # still over the terrain
    assert np.median(np.abs(qd[:, :3]).max(axis=1)) < 0.05             # at rest (spheres may still roll slowly down a slope)


def test_terrain_2048_worlds_rows_vs_checker_and_matching():
    """The bench's terrain scene at full size (2 048 worlds x 8 primitives on one 8 192-triangle terrain = 16 384 triangle-leg pairs
    per collide): two collide() calls give bit-identical rows, nothing overflows, sampled worlds equal the checker chain row for
    row, and frame-to-frame matching (contact_matching="latest") finds the previous frame's rows again after a small motion."""
    import newton_amd as nt
    import scenes

    E = 2048
    model = scenes.terrain_scene(E, 8, device="cuda:0", seed=6)
    t = model.env
    assert len(t.sdf_pair) == 8 and bool(t.sdf_pair_mesh_tri.all()) and t.np == 0
    pipe = nt.CollisionPipeline(model, broad_phase="sap", contact_matching="latest")
    c1, c2 = pipe.contacts(), pipe.contacts()
    state = model.state()
    state.body_q[:, 2] -= 0.007  # the scene starts 6 mm above the surface: 1 mm inside for this test
    pipe.collide(state, c1)
    first = _rows(c1)
    leg = pipe._sdf_leg
    info = leg.overflow(c1._flat)
    assert not info["overflow"], info
    assert info["candidate_pairs"] == 8 * E
    live = first["shape0"] >= 0
    assert live.sum() > 8 * E  # every primitive touches the terrain with at least one contact, most with several
    sample = [0, 1, 777, 2047]
    q = state.body_q.cpu().numpy().reshape(-1, 7)
    want = checker_rows(model, leg, q, worlds=sample)
    rs = first["row_start"]
    got_idx = np.concatenate([np.arange(rs[w], rs[w + 1]) for w in sample])
    assert len(want["key"]) == len(got_idx)
    assert np.array_equal(first["key"][got_idx], want["key"])
    for k in ("shape0", "shape1"):
        assert np.array_equal(first[k][got_idx], want[k]), k
    for k in FIELDS[2:]:
        assert np.abs(first[k][got_idx] - want[k]).max() <= 2e-6, (k, np.abs(first[k][got_idx] - want[k]).max())
    # second frame: every body moved by 0.2 mm -- the rows keep their fingerprints, the matcher finds them
    state.body_q[:, 0] += 0.0002
    pipe.collide(state, c2)
    second = _rows(c2)
    assert len(second["key"]) > 0
    mi = c2.rigid_contact_match_index
    n2 = int(c2.rigid_contact_count.item())
    matched = int((mi[:n2] >= 0).sum().item())
    assert matched > 0.8 * n2, (matched, n2)


def hfield_scene(worlds, kinds=("box", "sphere", "capsule", "hull", "cylinder"), device="cuda:0", gap=0.004, margin=0.0, seed=9, drop=0.0,
                 tilt=0.0):
    """Every world: one free body per kind resting slightly inside ONE shared static heightfield (nt.Heightfield, a global shape)."""
    import newton_amd as nt

    rng = np.random.default_rng(seed)
    nrow, ncol, hx, hy = 33, 41, 2.0, 1.6
    xs, ys = np.linspace(-hx, hx, ncol), np.linspace(-hy, hy, nrow)
    raw = np.array([[terrain_height(x, y) for x in xs] for y in ys], np.float32)
    field = nt.Heightfield(raw, nrow, ncol, hx=hx, hy=hy)
    half = dict(box=0.05, sphere=0.06, capsule=0.04, cylinder=0.05, hull=0.04)
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = gap
    env.default_shape_cfg.margin = margin
    for k, kind in enumerate(kinds):
        b = env.add_body(xform=[0.4 * k - 0.4, 0.0, 0.2, 0.0, 0.0, 0.0, 1.0])
        if kind == "box":
            env.add_shape_box(b, hx=0.08, hy=0.06, hz=0.05)
        elif kind == "sphere":
            env.add_shape_sphere(b, radius=0.06)
        elif kind == "capsule":
            env.add_shape_capsule(b, radius=0.04, half_height=0.08)
        elif kind == "hull":
            pts = np.array([(sx * 0.07, sy * 0.05, sz * 0.04) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)] +
                           [(0.0, -0.03, 0.06), (0.0, 0.03, 0.06)], np.float32)
            env.add_shape_convex_hull(b, mesh=nt.Mesh.convex_hull_of(pts))
        else:
            env.add_shape_cylinder(b, radius=0.05, half_height=0.05)
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = gap
    scene.default_shape_cfg.margin = margin
    scene.replicate(env, worlds)
    q_field = nt._np_math.quat_rpy(tilt, 0.0, 0.0)
    scene.add_shape_heightfield(heightfield=field, xform=[0.0, 0.0, 0.0, *q_field])
    model = scene.finalize(device=device)
    nbody = len(kinds)
    for w in range(worlds):
        for k, kind in enumerate(kinds):
            x, y = 0.6 * k - 1.2 + rng.uniform(-0.1, 0.1), rng.uniform(-0.8, 0.8)
            q = nt._np_math.quat_rpy(*(rng.uniform(-0.05, 0.05, size=2)), rng.uniform(-1.0, 1.0))
            if kind == "capsule":
                q = nt._np_math.quat_rpy(0.0, np.pi / 2 + rng.uniform(-0.03, 0.03), rng.uniform(-1.0, 1.0))
            i = w * nbody + k
            # (the field's bilinear-ish surface at the grid resolution: the analytic height is within a millimetre of it)
            p_local = np.array([x, y, terrain_height(x, y) + half[kind] - 0.001 + drop])
            model.body_q[i, :3] = nt._np_math.quat_rotate(q_field, p_local) if tilt else p_local
            model.body_q[i, 3:] = nt._np_math.quat_mul(q_field, q) if tilt else q
            model.joint_q.reshape(-1, 7)[i] = model.body_q[i]
    return model


@pytest.mark.parametrize("tilt,margin", [(0.0, 0.0), (0.15, 0.001)])
def test_collide_rows_of_shapes_on_a_heightfield_match_the_checker(tilt, margin):
    """GeoType.HFIELD through the triangle leg (narrow_phase.py:553-583, utils/heightfield.py:280-462): the grid-cell midphase from the
    partner's local AABB, TRIANGLE_PRISM cells, penetrating contacts moved to the physical face -- rows vs the checker chain."""
    import newton_amd as nt

    E = 4
    model = hfield_scene(E, tilt=tilt, margin=margin)
    t = model.env
    assert len(t.sdf_pair) == 5 and bool(t.sdf_pair_mesh_tri.all())
    assert int(t.shape_type[-1]) == int(nt.GeoType.HFIELD) and int(t.tile_shape_type[-1]) == int(nt.GeoType.CONVEX_MESH)
    pipe = nt.CollisionPipeline(model, broad_phase="nxn")
    c1, c2 = pipe.contacts(), pipe.contacts()
    state = model.state()
    pipe.collide(state, c1)
    pipe.collide(state, c2)
    got, again = _rows(c1), _rows(c2)
    for k in got:
        assert np.array_equal(got[k], again[k]), k
    leg = pipe._sdf_leg
    assert not leg.overflow(c1._flat)["overflow"]
    want = checker_rows(model, leg, model.body_q)
    assert len(want["key"]) == len(got["key"]) > 5 * E
    assert np.array_equal(got["key"], want["key"])
    for k in ("shape0", "shape1"):
        assert np.array_equal(got[k], want[k]), k
    field = model.shape_count - 1
    live = got["shape0"] >= 0
    assert live.sum() >= 5 * E and np.all(got["shape0"][live] == field)  # (heightfield, convex)
    for k in FIELDS[2:]:
        assert np.abs(got[k] - want[k]).max() <= 2e-6, (k, np.abs(got[k] - want[k]).max())


def test_xpbd_settles_shapes_on_a_heightfield():
    import newton_amd as nt

    E = 16
    model = hfield_scene(E, drop=0.02)
    pipe = nt.CollisionPipeline(model, broad_phase="nxn")
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=4)
    s0, s1 = model.state(), model.state()
    dt = 1.0 / 600.0
    for _ in range(600):
        s0.clear_forces()
        pipe.collide(s0, contacts)
        solver.step(s0, s1, None, contacts, dt)
        s0, s1 = s1, s0
    q = s0.body_q.cpu().numpy().reshape(-1, 7)
    qd = s0.body_qd.cpu().numpy().reshape(-1, 6)
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd))
    ground = terrain_height(q[:, 0], q[:, 1])
    assert np.all(q[:, 2] > ground + 0.02), (q[:, 2] - ground).min()
    assert np.all(q[:, 2] < ground + 0.12)
    assert np.median(np.abs(qd[:, :3]).max(axis=1)) < 0.05


@pytest.mark.parametrize("surface", ["mesh", "heightfield"])
def test_penalty_solvers_consume_the_triangle_rows(surface):
    """The rows feed SolverSemiImplicit and SolverFeatherstone like any SDF row (nt_flat_rows_forces: ordered per-body sums): boxes
    and spheres dropped from 5 mm onto the terrain come to rest on it under the penalty contact model."""
    import newton_amd as nt

    E = 4
    for make, dt, steps in ((lambda m: nt.solvers.SolverSemiImplicit(m), 1.0 / 4000.0, 1600),
                            (lambda m: nt.solvers.SolverFeatherstone(m), 1.0 / 4000.0, 1600)):
        model = (terrain_scene if surface == "mesh" else hfield_scene)(E, kinds=("box", "sphere"), drop=0.006)
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
        q, qd = s0.body_q.cpu().numpy().reshape(-1, 7), s0.body_qd.cpu().numpy().reshape(-1, 6)
        name = type(solver).__name__
        assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd)), name
        h = q[:, 2] - terrain_height(q[:, 0], q[:, 1])
        assert np.all(h > 0.03) and np.all(h < 0.08), (name, h)   # half extents 0.05 / 0.06: resting on the surface
        assert np.abs(qd[:, :3]).max() < 0.3, (name, np.abs(qd[:, :3]).max())
