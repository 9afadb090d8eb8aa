"""Texture-SDF sampling and the mesh-vs-SDF narrow phase on the MI355X (nt_sdf_sample / nt_mesh_sdf_collide through the C
ABI) against the float32 oracle (oracle/oracle_sdf.py): sampled distances and gradients within 1e-6, the contact SET of a
pair (which edges, which direction) bit-exact, contact geometry within 1e-5; plus the reference's accuracy table for the
sampler (newton/tests/test_sdf_texture.py:1347-1376: analytic sphere, mean < 5e-4, p95 < 1e-3) asserted on the device."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from test_sdf_contact import box_sdf, oracle_contacts, two_box_scene  # noqa: E402

from newton_amd import sdf as S  # noqa: E402
from newton_amd.enums import GeoType  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [S.QuantizationMode.FLOAT32, S.QuantizationMode.UINT16, S.QuantizationMode.UINT8])
def test_device_sampler_vs_oracle(mode):
    import oracle_sdf as O

    from newton_amd.sdf_device import DeviceSDF

    t = box_sdf(mode=mode)
    dev = DeviceSDF(t)
    pts = np.random.default_rng(9).uniform(-0.9, 0.9, size=(2000, 3)).astype(np.float32)
    dist, grad = dev.sample(pts, grad=True)
    dist, grad = dist.cpu().numpy(), grad.cpu().numpy()
    o = O.OracleSDF(t)
    k = np.arange(0, 2000, 7)
    assert np.max(np.abs(dist[k] - np.array([o.sample(p) for p in pts[k]]))) <= 1e-6
    assert np.max(np.abs(grad[k] - np.array([o.sample_grad_fd(p) for p in pts[k]]))) <= 2e-5  # (v0 - v1) * inv_dx, inv_dx ~ 30
    assert np.max(np.abs(dist - t.sample(pts))) <= 1e-6  # and the vectorised host sampler on every point


def test_device_sampler_reference_accuracy_table():
    from newton_amd.sdf_device import DeviceSDF

    t = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.5, 0.0, 0.0), max_resolution=64,
                                            quantization_mode=S.QuantizationMode.FLOAT32)
    rng = np.random.default_rng(123)
    d = rng.normal(size=(2000, 3))
    q = (d / np.linalg.norm(d, axis=1, keepdims=True) * (0.5 + rng.uniform(-0.08, 0.08, size=(2000, 1)))).astype(np.float32)
    got = DeviceSDF(t).sample(q).cpu().numpy()
    diff = np.abs(got - (np.linalg.norm(q, axis=1) - 0.5))
    assert diff.mean() < 5e-4 and np.percentile(diff, 95) < 1e-3


@pytest.mark.parametrize("dz,margin,mode", [(0.98, 0.0, S.QuantizationMode.UINT16), (1.02, 0.005, S.QuantizationMode.FLOAT32),
                                            (0.9, 0.005, S.QuantizationMode.UINT8)])
def test_device_mesh_sdf_contacts_vs_oracle(dz, margin, mode):
    from newton_amd.sdf_device import DeviceSDF, mesh_sdf_collide

    sc = two_box_scene(dz=dz, margin=margin, mode=mode)
    want = oracle_contacts(sc)
    got = mesh_sdf_collide(sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["sdf_index"], [DeviceSDF(sc["sdfs"][0])], sc["er"],
                           sc["ec"], sc["eh"])
    assert got["count"] == len(want) > 0
    assert list(zip(got["pair"].tolist(), got["key"].tolist())) == sorted((p, k) for p, k, *_ in want)
    by_key = {(p, k): (c, n, d) for p, k, c, n, d, _, _ in want}
    for i in range(len(want)):
        c, n, d = by_key[(int(got["pair"][i]), int(got["key"][i]))]
        assert np.max(np.abs(got["center"][i] - c)) <= 1e-5 and np.max(np.abs(got["normal"][i] - n)) <= 1e-5
        assert abs(got["distance"][i] - d) <= 1e-5


def test_device_mesh_sdf_many_pairs_scales_and_stays_deterministic():
    """64 hull-like cubes scattered in a bin: 2 016 pairs through one launch; two runs give the same sorted contact list, and a
    256-pair sample agrees with the oracle (contact set exact)."""
    from newton_amd.mesh import Mesh, mesh_edge_tables
    from newton_amd.sdf_device import DeviceSDF, mesh_sdf_collide

    rng = np.random.default_rng(4)
    n = 64
    m = Mesh.create_box(0.05, 0.04, 0.03)
    ec, eh = mesh_edge_tables(m.vertices, m.indices.reshape(-1, 3))
    t = S.create_texture_sdf_from_primitive(GeoType.BOX, (0.05, 0.04, 0.03), max_resolution=32, margin=0.02,
                                            narrow_band_range=(-0.03, 0.03))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = np.concatenate([rng.uniform(-0.15, 0.15, size=(n, 3)), q], axis=1).astype(np.float32)
    pairs = np.array([(a, b) for a in range(n) for b in range(a + 1, n)], dtype=np.int32)
    data = np.tile(np.array([1, 1, 1, 0.001], dtype=np.float32), (n, 1))
    gap = np.full(n, 0.005, dtype=np.float32)
    idx, er = np.zeros(n, dtype=np.int32), np.tile(np.array([0, len(ec)], dtype=np.int32), (n, 1))
    dev = [DeviceSDF(t)]
    a = mesh_sdf_collide(pairs, X, data, gap, idx, dev, er, ec, eh)
    b = mesh_sdf_collide(pairs, X, data, gap, idx, dev, er, ec, eh)
    assert a["count"] == b["count"] > 50
    for k in ("pair", "key", "center", "normal", "distance"):
        assert np.array_equal(a[k], b[k])
    import oracle_sdf as O

    sample = np.flatnonzero(np.isin(np.arange(len(pairs)), np.unique(a["pair"])[:40]))
    want = O.mesh_sdf_collide(pairs[sample], X, data, gap, idx, [t], er, ec, eh)
    got = {(int(p), int(k)) for p, k in zip(a["pair"], a["key"]) if p in set(sample.tolist())}
    assert got == {(int(sample[p]), int(k)) for p, k, *_ in want}


@pytest.mark.parametrize("kh,margin", [((1e6, 1e6), 0.0), ((3e6, 5e5), 0.001)])
def test_device_hydroelastic_faces_vs_oracle(kh, margin):
    """nt_hydro_collide on the device: the iso-pressure faces of a ball pressed into a slab -- the face set (voxel, face) equals
    the oracle's, geometry / separation / stiffness / area / pressure within fp32 tolerance; the patch carries the closed-form
    force for equal stiffness."""
    from test_hydroelastic import oracle_faces, sphere_on_slab

    from newton_amd.sdf_device import DeviceSDF, hydro_collide

    sc = sphere_on_slab(res=16, kh=kh, margin=margin)
    want = oracle_faces(sc)
    got = hydro_collide(sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["kh"], np.array([0, 1], dtype=np.int32),
                        [DeviceSDF(t) for t in sc["sdfs"]])
    assert got["count"] == len(want) > 10
    assert list(zip(got["pair"].tolist(), got["key"].tolist())) == [(p, k) for p, k, *_ in want]
    for i, w in enumerate(want):
        assert (int(got["shape_a"][i]), int(got["shape_b"][i])) == (w[2], w[3])
        assert np.max(np.abs(got["center"][i] - w[4])) <= 1e-5 and np.max(np.abs(got["normal"][i] - w[5])) <= 1e-4
        assert abs(got["distance"][i] - w[6]) <= 1e-6 and abs(got["stiffness"][i] - w[7]) <= 1e-3 * max(1.0, abs(w[7]))
        assert abs(got["area"][i] - w[8]) <= 1e-8 + 1e-4 * w[8] and abs(got["pressure"][i] - w[9]) <= 1e-2 + 1e-4 * w[9]


def test_device_hydroelastic_patch_closed_form_at_resolution_48():
    from test_hydroelastic import sphere_on_slab

    from newton_amd.sdf_device import DeviceSDF, hydro_collide

    R, delta, kh = 0.1, 0.005, 1e6
    sc = sphere_on_slab(delta=delta, res=48)
    got = hydro_collide(sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["kh"], np.array([0, 1], dtype=np.int32),
                        [DeviceSDF(t) for t in sc["sdfs"]])
    pen = got["distance"] < 0
    area, force = got["area"][pen].sum(), (got["stiffness"][pen] * -got["distance"][pen]).sum()
    assert abs(area - 2 * np.pi * R * delta) / (2 * np.pi * R * delta) < 0.08
    assert abs(force - kh * np.pi * R * delta ** 2 / 2) / (kh * np.pi * R * delta ** 2 / 2) < 0.08


def test_device_eval_body_contact_consumes_per_contact_stiffness():
    """Contacts(per_contact_shape_properties=True): SolverSemiImplicit on the device vs the oracle with per-slot overrides."""
    import newton_amd as nt
    from oracle_bridge import Oracle, OracleState
    from scenes import mixed_primitive_scene

    model = mixed_primitive_scene(5, device="cuda:0")
    model.body_q[:, 2] -= 0.03
    pipe = nt.CollisionPipeline(model)
    ct = pipe.contacts(per_contact_shape_properties=True)
    s0, s1 = model.state(), model.state()
    pipe.collide(s0, ct)
    t = model.env
    ns = t.np * t.cpp
    rng = np.random.default_rng(5)
    ke = rng.choice([0.0, 2.0e4, 7.5e3], size=(ns, t.env_count)).astype(np.float32)
    kd = rng.choice([0.0, 30.0], size=(ns, t.env_count)).astype(np.float32)
    mu = rng.choice([0.0, 0.5, 2.0], size=(ns, t.env_count)).astype(np.float32)
    ct.set_slot_properties(ke, kd, mu)
    nt.solvers.SolverSemiImplicit(model).step(s0, s1, None, ct, 1e-3)
    o = Oracle(model)
    os0, os1, oc = OracleState(model), OracleState(model), o.contacts()
    o.collide(os0.body_q, oc)
    shape0 = ct._shape0[:ns, : t.env_count].cpu().numpy()
    nas = t.np_analytic * t.cpp
    live = [(env, s) for env in range(t.env_count) for s in range(nas) if shape0[s, env] >= 0] + \
           [(env, s) for env in range(t.env_count) for s in range(nas, ns) if shape0[s, env] >= 0]
    assert len(live) == int(oc.count[0]) > 0
    oc.set_properties([ke[s, env] for env, s in live], [kd[s, env] for env, s in live], [mu[s, env] for env, s in live])
    o.semi_implicit_step(os0, os1, o.control(), oc, 1e-3)
    assert np.max(np.abs(s1.body_qd.cpu().numpy() - os1.body_qd)) <= 2e-4 * max(1.0, np.abs(os1.body_qd).max())


# ------------------------------------------------------------------------------------------------ global contact reduction
@pytest.mark.parametrize("name", ["patch", "two_pairs_many_normals", "duplicates_and_ties", "outer_only", "single"])
def test_device_reduction_kernel_keeps_what_the_reference_reducer_keeps(name):
    """contacts_reduce_list_kernel on the MI355X against the record of the reference's own reducer
    (tests/golden/reduce_reference_vectors.npz): survivors, points, distances and exported normals bit for bit."""
    import ctypes as C

    import torch

    sys.path.insert(0, os.path.join(HERE, "golden"))
    import reduce_cases as rc
    from test_contact_reduction import run_reduce_list

    from newton_amd import _lib as L

    ref = np.load(os.path.join(HERE, "golden", "reduce_reference_vectors.npz"))
    c = rc.pack(rc.contacts(name))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def to_host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()

    idx, nrm, pair, fp = run_reduce_list(L.load(), c, stream=stream, to_dev=lambda a: torch.from_numpy(a).cuda(), to_host=to_host)
    o = np.lexsort((fp, pair[:, 1], pair[:, 0]))
    assert np.array_equal(pair[o], ref[f"{name}/pair"]) and np.array_equal(fp[o], ref[f"{name}/fp"])
    assert np.array_equal(c["pos"][idx[o]], ref[f"{name}/pos"]) and np.array_equal(c["depth"][idx[o]], ref[f"{name}/depth"])
    assert np.array_equal(nrm[o], ref[f"{name}/normal"])


def test_device_reduced_mesh_sdf_kernel_is_the_reduction_of_the_unreduced_one():
    """nt_mesh_sdf_collide_reduced against the checker's reduction (oracle_reduce, pinned by the executed reference reducer) of
    nt_mesh_sdf_collide's own rows on the sphere-on-box scene: bit for bit; and against the float32 checker end to end."""
    import oracle_reduce as R
    from test_sdf_contact import check_reduced_against_unreduced, sphere_on_box_scene

    from newton_amd.sdf_device import DeviceSDF, mesh_sdf_collide

    sc = sphere_on_box_scene()
    dev = [DeviceSDF(t) for t in sc["sdfs"]]
    args = (sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["sdf_index"], dev, sc["er"], sc["ec"], sc["eh"])
    u = mesh_sdf_collide(*args)
    r = mesh_sdf_collide(*args, reduce=(sc["aabb_lo"], sc["aabb_hi"], sc["res"]))
    r2 = mesh_sdf_collide(*args, reduce=(sc["aabb_lo"], sc["aabb_hi"], sc["res"]), staged=True)
    assert all(np.array_equal(r[k], r2[k]) for k in r)
    pack = lambda d: (d["pair"], d["key"], np.concatenate([d["center"], d["normal"], d["distance"][:, None],  # noqa: E731
                                                           d["margin0"][:, None], d["margin1"][:, None]], axis=1))
    n_in, n_out = check_reduced_against_unreduced(sc, pack(u), pack(r))
    assert n_in > n_out + 50 and r["count"] == n_out
    rows = oracle_contacts(sc)
    c = R.reduce_inputs_from_mesh_sdf_contacts(rows, sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["sdf_index"], sc["sdfs"],
                                               sc["aabb_lo"], sc["aabb_hi"], sc["res"])
    want = R.reduce_contacts(c)
    shape_pair = sc["pairs"][r["pair"]]
    assert np.array_equal(shape_pair, want["pair"]) and np.array_equal(r["key"], want["fp"])
    assert np.abs(r["center"] - want["pos"]).max() <= 1e-6 and np.abs(r["normal"] - want["normal"]).max() <= 1e-6


def test_device_reduced_mesh_sdf_bin_scale():
    """C5's geometry class: 64 small boxes scattered in a bin, 2 016 pairs in one launch, reduced: twice the same rows; every
    pair's block is a subset of its unreduced rows, never more than 245, blocks ordered by fingerprint."""
    from newton_amd.mesh import Mesh, mesh_edge_tables
    from newton_amd.sdf_device import DeviceSDF, mesh_sdf_collide

    rng = np.random.default_rng(4)
    n = 64
    m = Mesh.create_sphere(0.05, 10, 12)
    ec, eh = mesh_edge_tables(m.vertices, m.indices.reshape(-1, 3))
    t = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.05, 0.05, 0.05), max_resolution=32, margin=0.02,
                                            narrow_band_range=(-0.03, 0.03))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    X = np.concatenate([rng.uniform(-0.12, 0.12, size=(n, 3)), q], axis=1).astype(np.float32)
    pairs = np.array([(a, b) for a in range(n) for b in range(a + 1, n)], dtype=np.int32)
    data = np.tile(np.array([1, 1, 1, 0.001], dtype=np.float32), (n, 1))
    gap = np.full(n, 0.01, dtype=np.float32)
    idx, er = np.zeros(n, dtype=np.int32), np.tile(np.array([0, len(ec)], dtype=np.int32), (n, 1))
    tables = S.mesh_reduction_tables([m.vertices] * n, [(1, 1, 1)] * n)
    dev = [DeviceSDF(t)]
    u = mesh_sdf_collide(pairs, X, data, gap, idx, dev, er, ec, eh)
    a = mesh_sdf_collide(pairs, X, data, gap, idx, dev, er, ec, eh, reduce=tables)
    b = mesh_sdf_collide(pairs, X, data, gap, idx, dev, er, ec, eh, reduce=tables)
    assert 0 < a["count"] == b["count"] < u["count"]
    for k in ("pair", "key", "center", "normal", "distance"):
        assert np.array_equal(a[k], b[k])
    # the staged variant (cull over all pairs -> one lane per survivor -> reduction per pair; 324 edges = six rounds per wave)
    c = mesh_sdf_collide(pairs, X, data, gap, idx, dev, er, ec, eh, reduce=tables, staged=True)
    assert c["count"] == a["count"]
    for k in ("pair", "key", "center", "normal", "distance", "margin0", "margin1"):
        assert np.array_equal(a[k], c[k]), k
    unreduced = set(zip(u["pair"].tolist(), u["key"].tolist()))
    assert set(zip(a["pair"].tolist(), a["key"].tolist())) <= unreduced
    assert np.bincount(a["pair"]).max() <= 245
    assert set(np.unique(a["pair"]).tolist()) == set(np.unique(u["pair"]).tolist())  # no touching pair loses all its contacts


@pytest.mark.parametrize("threads", [64, 256])
def test_device_resident_narrow_phase_takes_the_broad_phase_counter(threads):
    """MeshSdfNarrowPhase.launch: pairs and their live count stay on the device (the SAP broad phase's candidate array and
    counter), one wave or four per pair: the same rows as the host-driven wrapper, and rows beyond the counter are ignored."""
    import torch
    from test_sdf_contact import sphere_on_box_scene

    from newton_amd.sdf_device import DeviceSDF, MeshSdfNarrowPhase, mesh_sdf_collide

    sc = sphere_on_box_scene()
    dev = [DeviceSDF(t) for t in sc["sdfs"]]
    want = mesh_sdf_collide(sc["pairs"], sc["X"], sc["data"], sc["gap"], sc["sdf_index"], dev, sc["er"], sc["ec"], sc["eh"],
                            reduce=(sc["aabb_lo"], sc["aabb_hi"], sc["res"]))
    nphase = MeshSdfNarrowPhase(sc["data"], sc["gap"], sc["sdf_index"], dev, sc["er"], sc["ec"], sc["eh"],
                                (sc["aabb_lo"], sc["aabb_hi"], sc["res"]))
    pairs = torch.zeros((16, 2), dtype=torch.int32, device="cuda:0")
    pairs[:3] = torch.from_numpy(sc["pairs"]).cuda()
    pairs[3:] = torch.tensor([0, 1], dtype=torch.int32)  # stale rows past the live count
    count = torch.tensor([3], dtype=torch.int32, device="cuda:0")
    cap = 4096
    o_count = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    o_pair, o_key = torch.zeros(cap, dtype=torch.int32, device="cuda:0"), torch.zeros(cap, dtype=torch.int32, device="cuda:0")
    o_data = torch.zeros((cap, 9), dtype=torch.float32, device="cuda:0")
    nphase.launch(torch.from_numpy(sc["X"]).cuda(), pairs, count, o_count, o_pair, o_key, o_data, threads=threads)
    torch.cuda.synchronize()
    n = int(o_count.item())
    assert n == want["count"]
    o = np.lexsort((o_key[:n].cpu().numpy(), o_pair[:n].cpu().numpy()))
    assert np.array_equal(o_pair[:n].cpu().numpy()[o], want["pair"]) and np.array_equal(o_key[:n].cpu().numpy()[o], want["key"])
    d = o_data[:n].cpu().numpy()[o]
    assert np.array_equal(d[:, 0:3], want["center"]) and np.array_equal(d[:, 3:6], want["normal"])
    assert np.array_equal(d[:, 6], want["distance"])


# ------------------------------------------------------------------------------------------------ contacts outside the tiles
@pytest.mark.parametrize("name", ["dynamic_pairs", "with_static_shapes", "per_contact_properties"])
def test_device_flat_contact_kernels_against_the_reference(name):
    """nt_contact_rows_write + nt_eval_body_contact_flat on the MI355X against the record of the reference's own write_contact
    and eval_body_contact: the written rows bit for bit, the accumulated forces to the rounding of the atomic sums."""
    import ctypes as C

    import torch

    sys.path.insert(0, os.path.join(HERE, "golden"))
    import flat_contact_cases as fc
    from test_flat_contacts import check_flat_stage, run_flat_stage

    from newton_amd import _lib as L

    def to_host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()

    out, body_f = run_flat_stage(L.load(), fc.make(name), to_dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda(),
                                 to_host=to_host, ptr=lambda t: t.data_ptr(),
                                 stream=C.c_void_p(torch.cuda.current_stream().cuda_stream))
    check_flat_stage(name, out, body_f)


def test_mesh_sdf_contact_stage_holds_hulls_apart_under_semi_implicit():
    """C5's contact model end to end at small size: 4 environments x 12 convex hulls dropped into the bin; hull-hull contacts
    come from MeshSdfContactStage (SAP -> mesh-SDF + reduction -> write_contact rows -> eval_body_contact into body_f), hull-wall
    contacts from the tiles; SolverSemiImplicit integrates.  The hulls must come to rest inside the bin, not interpenetrating
    (compared with the convex MPR/GJK path's notion of contact on the final poses), and the stage's rows must satisfy the
    writer's invariants."""
    import torch
    from scenes import hull_bin_scene

    import newton_amd as nt
    from newton_amd.sdf_device import MeshSdfContactStage

    E, H = 4, 12
    # explicit penalty contacts: kf * n_contacts * dt / m < 2 must hold on the 25 g hull with its ~10 simultaneous rows, hence
    # kf = 20 (kf = 200 allows one); measured sweeps: profiles/r02h_sdf_stage_sweep.jsonl, profiles/r02j_sdf_stage_sweep2*.jsonl
    cfg = dict(ke=2.0e3, kd=10.0, kf=20.0, mu=0.5, gap=0.004)
    # + 1e-4 kg m^2 on the inertias: explicit friction is unstable on the bare 25 g hulls (scenes.hull_bin_scene docstring)
    model = hull_bin_scene(E, H, device="cuda:0", seed=2, hull_pairs=False, shape_cfg=cfg, inertia_armature=1.0e-4)
    stage = MeshSdfContactStage(model, sdf_resolution=24)
    assert stage.n == E * H and len(stage.sdfs) == H  # one SDF per hull asset, shared by the environments
    pipe = nt.CollisionPipeline(model)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverSemiImplicit(model)
    s0, s1, ctrl = model.state(), model.state(), model.control()
    dt = 1.0 / 4000.0
    seen = 0
    for k in range(6000):  # 1.5 s
        s0.clear_forces()
        stage.collide(s0)
        stage.apply_forces(s0)
        pipe.collide(s0, contacts)
        solver.step(s0, s1, ctrl, contacts, dt)
        s0, s1 = s1, s0
        if k % 500 == 499:
            seen = max(seen, int(stage.row_count.item()))
    torch.cuda.synchronize()
    q, qd = s0.body_q.cpu().numpy(), s0.body_qd.cpu().numpy()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd))
    assert seen > 0  # hulls did touch each other
    assert q[:, 2].min() > 0.0 and q[:, 2].max() < 0.6 and np.abs(q[:, :2]).max() < 0.45  # inside the bin, above the ground
    # settled: the atomic force sums make the run chaotic in its last bits, so the gate is statistical -- most hulls at rest,
    # none faster than free fall from the top of the start lattice allows (the criterion of the reference's test_box_drop,
    # newton/tests/test_rigid_contact.py:775-845); 8 measured runs ended with max speeds of 0.16 - 0.27 m/s
    speed = np.linalg.norm(qd[:, :3], axis=1)
    print(f"[mesh-sdf stage] speed median {np.median(speed):.3e} max {speed.max():.3e}  rows seen {seen}")
    assert np.median(speed) < 0.1 and speed.max() < np.sqrt(2.0 * 9.81 * 0.3)
    # writer invariants on the last rows: unit normals, both shapes on different bodies of the same environment
    n = int(stage.row_count.item())
    a, b = stage.rigid_contact_shapes()
    live = (a >= 0).cpu().numpy()
    nr = stage.normal[:n].cpu().numpy()[live]
    assert np.abs(np.linalg.norm(nr, axis=1) - 1.0).max() < 1e-5
    a, b = a.cpu().numpy()[live], b.cpu().numpy()[live]
    assert np.all(a < b) and np.all(a // H == b // H)
    # no deep interpenetration: every remaining row's separation is above -3 mm
    sep = stage._row_data[:n, 6].cpu().numpy()[live]
    assert sep.min() > -3e-3
