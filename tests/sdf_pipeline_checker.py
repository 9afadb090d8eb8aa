"""TEST INFRASTRUCTURE: the mesh-SDF leg of CollisionPipeline.collide restated with the float32 checkers of oracle/ on Newton's
flat arrays -- world transforms and AABBs (C++ checker), per-world candidate pairs (brute force over the SDF pair rule of
narrow_phase.py:620-640), edge-vs-SDF contacts (oracle_sdf), the global reduction (oracle_reduce) and write_contact
(oracle_flat_contacts) -- in the order the product emits its rows: world-major, pairs ascending, fingerprints ascending."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))

import oracle_flat_contacts as F  # noqa: E402
import oracle_reduce as R  # noqa: E402
import oracle_sdf as O  # noqa: E402
from oracle_bridge import Oracle  # noqa: E402

from newton_amd.enums import GeoType  # noqa: E402


def sdf_scene(world_count, n_hulls=5, device=None, seed=5, sdf_resolution=16, walls=False, spacing=0.07, gap=0.01, mu=0.5,
              verts=(8, 13), jitter=0.004):
    """`n_hulls` random convex hulls per world, each with a texture SDF attached (mesh.build_sdf), piled closely enough to touch,
    over a ground plane (hull-plane pairs stay in the tiles: MPR / GJK); `walls`: two static boxes with generated SDFs (global
    shapes) so that hull-box pairs take the SDF leg as well."""
    import newton_amd as nt

    rng = np.random.default_rng(seed)
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = gap
    env.default_shape_cfg.mu = mu
    for k in range(n_hulls):
        pts = rng.normal(size=(int(rng.integers(*verts)), 3))
        pts *= rng.uniform(0.03, 0.05) / np.linalg.norm(pts, axis=1).max()
        mesh = nt.Mesh.convex_hull_of(pts)
        mesh.build_sdf(max_resolution=sdf_resolution, margin=0.02, narrow_band_range=(-0.05, 0.05))
        pos = [(k % 2 - 0.5) * spacing, ((k // 2) % 2 - 0.5) * spacing, 0.05 + (k // 4) * spacing]
        b = env.add_body(xform=[*pos, *nt._np_math.quat_rpy(*rng.uniform(-1.0, 1.0, size=3))])
        env.add_shape_convex_hull(b, mesh=mesh)
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = gap
    scene.default_shape_cfg.mu = mu
    scene.replicate(env, world_count)
    scene.add_ground_plane()
    if walls:
        cfg = scene.default_shape_cfg.copy()
        cfg.configure_sdf(max_resolution=16)
        for sx in (1, -1):
            scene.add_shape_box(-1, xform=[sx * (0.5 * spacing + 0.06), 0.0, 0.1, 0.0, 0.0, 0.0, 1.0], hx=0.02, hy=0.2, hz=0.1, cfg=cfg)
    model = scene.finalize(device=device)
    if jitter > 0.0:
        off = np.random.default_rng(seed + 1000).uniform(-jitter, jitter, size=(model.body_count, 3)).astype(np.float32)
        model.body_q[:, :3] += off
        model.joint_q.reshape(-1, 7)[:, :3] += off
    return model


def checker_rows(model, body_q, world_xform=None, aabbs=None, kinds=None, worlds=None, reduce=True):
    """-> (rows dict in product order incl. `world`, `key`; candidate pairs per world [list of (a, b)]; the checker's world
    transforms and AABBs).  `world_xform` / `aabbs`: use the device's own exported arrays instead (the caller holds them against
    the checker's separately) -- the centred-difference SDF gradient amplifies a last-bit difference of a shape transform to
    1e-5 in the normal, which would blur the comparison of everything downstream.
    `kinds` (bool per template SDF pair, True = hydroelastic): those pairs are listed in the candidates -- which then hold
    ((shape0, shape1), kind) tuples -- but produce no edge-contact rows here (oracle_hydro.hydro_pipeline covers them).
    `reduce=False`: CollisionPipeline(reduce_contacts=False) -- every contact of mesh_sdf_collision_kernel, per pair in ascending
    fingerprint (the order Newton's deterministic sort gives them), the kernel's own normal."""
    t = model.env
    o = Oracle(model)
    oc = o.contacts()
    _, lo, hi = o.collide(np.asarray(body_q, np.float32), oc)  # AABBs (gap-widened) of compute_shape_aabbs
    S = model.shape_count
    X = np.zeros((S, 7), np.float32)
    for s in range(S):
        b = int(model.shape_body[s])
        xs = np.asarray(model.shape_transform[s], np.float32)
        X[s] = xs if b < 0 else O._x_mul(np.asarray(body_q[b], np.float32), xs)
    X_checker, lo_checker, hi_checker = X, lo, hi
    if world_xform is not None:
        X = np.asarray(world_xform, np.float32)
    if aabbs is not None:
        lo, hi = (np.asarray(a, np.float32) for a in aabbs)
    data = np.concatenate([np.asarray(model.shape_scale, np.float32), np.asarray(model.shape_margin, np.float32)[:, None]], axis=1)
    gap = np.asarray(model.shape_gap, np.float32)
    idx, er = np.asarray(model._shape_sdf_index), np.asarray(model.shape_edge_range)
    ec, eh = model.mesh_edge_centers, model.mesh_edge_halves
    sdfs = model._texture_sdf_data
    alo, ahi, res = model.shape_collision_aabb_lower, model.shape_collision_aabb_upper, model._shape_voxel_resolution

    def gid(l, w):
        return t.shape_local0 + w * t.ns + l if l < t.ns else int(t.gshape_id[l - t.ns])

    out = {k: [] for k in ("world", "key", "shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1")}
    cand = []
    for w in range(t.env_count):
        pairs, tagged = [], []
        if worlds is not None and w not in worlds:
            cand.append([])
            continue
        for k, (a, b) in enumerate(t.sdf_pair):
            s1, s2 = sorted((gid(int(a), w), gid(int(b), w)))
            if np.all(lo[s1] <= hi[s2]) and np.all(hi[s1] >= lo[s2]):
                tagged.append(((s1, s2), bool(kinds[k]) if kinds is not None else False))
                if kinds is None or not kinds[k]:
                    pairs.append((s1, s2))
        cand.append(tagged if kinds is not None else pairs)
        if not pairs:
            continue
        pr = np.asarray(pairs, np.int32)
        rows = O.mesh_sdf_collide(pr, X, data, gap, idx, sdfs, er, ec, eh)
        if not rows:
            continue
        if not reduce:
            rows = sorted(rows, key=lambda r: (int(r[0]), int(r[1])))
            pi = np.asarray([int(r[0]) for r in rows])
            red = dict(fp=np.asarray([int(r[1]) for r in rows], np.int32), pair=pr[pi], pos=np.asarray([r[2] for r in rows], np.float32),
                       normal=np.asarray([r[3] for r in rows], np.float32), depth=np.asarray([r[4] for r in rows], np.float32),
                       index=None)
        else:
            c = R.reduce_inputs_from_mesh_sdf_contacts(rows, pr, X, data, gap, idx, sdfs, alo, ahi, res)
            red = R.reduce_contacts(c)
        k = red["index"]
        raw = dict(key=red["fp"], shape_a=red["pair"][:, 0], shape_b=red["pair"][:, 1], center=red["pos"], normal=red["normal"],
                   distance=red["depth"], margin_a=data[red["pair"][:, 0], 3], margin_b=data[red["pair"][:, 1], 3])
        del k
        wr = F.write_rows(raw, np.asarray(body_q, np.float32), np.asarray(model.shape_body), gap)
        n = len(red["fp"])
        out["world"] += [w] * n
        out["key"] += red["fp"].tolist()
        for name in ("shape0", "shape1", "point0", "point1", "offset0", "offset1", "normal", "margin0", "margin1"):
            out[name] += list(wr[name])
    return {k: np.asarray(v) for k, v in out.items()}, cand, (X_checker, lo_checker, hi_checker)


def hydro_checker_rows(model, body_q, world_xform, aabbs, reduce, worlds=None):
    """The rows CollisionPipeline.collide emits for a model with hydroelastic pairs, from the checker chain
    (oracle_hydro.hydro_pipeline: SAT, octree, marching cubes, optional reduction; the mesh-SDF edge leg for the other SDF pairs;
    write_contact): dict of world / key / shape0 / shape1 / point0 / normal / stiffness / friction in the product's row order
    (world-major, pairs ascending; a hydroelastic pair's faces in traversal order resp. the reduction's export order, an edge
    pair's contacts in fingerprint order).  `reduce`: False, True (HydroelasticSDF.Config() as it comes) or "moment".
    `worlds`: only these worlds (their candidates are still taken from the given AABBs)."""
    import oracle_hydro as H

    from newton_amd.mc_tables import tables

    t = model.env
    X, (lo, hi) = np.asarray(world_xform, np.float32), aabbs
    mesh_rows, cand, _ = checker_rows(model, np.asarray(body_q), world_xform=X, aabbs=(lo, hi), kinds=t.sdf_pair_hydro, worlds=worlds)
    tr, fl = tables()
    tab = (np.asarray(tr), np.asarray(fl).reshape(-1, 2))
    data = np.concatenate([np.asarray(model.shape_scale, np.float32), np.asarray(model.shape_margin, np.float32)[:, None]], axis=1)
    gap, kh = np.asarray(model.shape_gap, np.float32), np.asarray(model.shape_material_kh, np.float32)
    sdfs = [model._texture_sdf_data[i] if i >= 0 else None for i in np.asarray(model._shape_sdf_index)]
    want = {k: [] for k in ("world", "key", "shape0", "shape1", "point0", "normal", "stiffness", "friction")}
    for w in (range(t.env_count) if worlds is None else worlds):
        hp = [p for p, kind in cand[w] if kind]
        red = dict(aabb_lo=np.asarray(model.shape_collision_aabb_lower, np.float32), aabb_hi=np.asarray(model.shape_collision_aabb_upper, np.float32),
                   res=np.asarray(model._shape_voxel_resolution, np.int32), pre_prune=True, normal_matching=True,
                   moment_matching=reduce == "moment") if reduce else None
        rows, _ = H.hydro_pipeline(np.asarray(hp, np.int32), X, data, gap, kh, sdfs, tab, reduce=red) if hp else ([], None)
        per_pair = {}
        for r in rows:
            per_pair.setdefault(hp[r[0]], []).append(r)
        m = {k: np.asarray(v)[np.asarray(mesh_rows["world"]) == w] for k, v in mesh_rows.items()}
        for p, kind in cand[w]:
            if kind:
                rs = per_pair.get(p, [])
                raw = dict(key=np.array([r[1] for r in rs]), shape_a=np.array([r[2] for r in rs]), shape_b=np.array([r[3] for r in rs]),
                           center=np.array([r[4] for r in rs], np.float32).reshape(-1, 3), normal=np.array([r[5] for r in rs], np.float32).reshape(-1, 3),
                           distance=np.array([r[6] for r in rs], np.float32), margin_a=np.zeros(len(rs), np.float32), margin_b=np.zeros(len(rs), np.float32))
                wr = F.write_rows(raw, np.asarray(body_q, np.float32), np.asarray(model.shape_body), np.full(model.shape_count, 1e9, np.float32))
                want["world"] += [w] * len(rs)
                want["key"] += [r[1] for r in rs]
                want["stiffness"] += [r[7] for r in rs]
                want["friction"] += [r[8] if reduce else 0.0 for r in rs]
                for k in ("shape0", "shape1", "point0", "normal"):
                    want[k] += list(wr[k])
            else:
                sel = (m["shape0"] == p[0]) & (m["shape1"] == p[1]) if len(m["key"]) else np.zeros(0, bool)
                want["world"] += [w] * int(sel.sum())
                want["key"] += m["key"][sel].tolist()
                want["stiffness"] += [0.0] * int(sel.sum())
                want["friction"] += [0.0] * int(sel.sum())
                for k in ("shape0", "shape1", "point0", "normal"):
                    want[k] += list(m[k][sel])
    return want, cand
