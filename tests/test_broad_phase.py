"""Broad phase: explicit-pair, NxN (world / collision-group / filter-pair rules of broad_phase_common.py:217-268,132-162,
broad_phase_nxn.py:132-218) and a numpy brute force must produce the same candidate-pair SET (integer output, exact), in the
spirit of newton/tests/test_broad_phase.py:91-145.  The HIP path evaluates the explicit per-env list for every mode; its
candidate mask is checked against the same brute force on the GPU."""
import numpy as np
import pytest

import newton_amd as nt
from newton_amd.enums import ShapeFlags


def _scene(world_count, seed, device=None):
    """Free bodies with mixed primitives, random collision groups (0 = off, >0 exclusive, <0 collide-with-others) and a
    few explicit filter pairs, plus two global static shapes."""
    rng = np.random.default_rng(seed)
    env = nt.ModelBuilder()
    groups = [1, 1, 2, -1, -2, 0, 3, -1, 2, 1]
    kinds = ["sphere", "box", "capsule", "sphere", "box", "sphere", "capsule", "box", "sphere", "box"]
    for k, (g, kind) in enumerate(zip(groups, kinds)):
        p = rng.uniform(-0.45, 0.45, size=3) + np.array([0.0, 0.0, 0.6])
        b = env.add_body(xform=[*p, 0.0, 0.0, 0.0, 1.0])
        cfg = nt.ModelBuilder.ShapeConfig(collision_group=g, gap=0.02)
        if kind == "sphere":
            env.add_shape_sphere(b, radius=0.12, cfg=cfg)
        elif kind == "box":
            env.add_shape_box(b, hx=0.12, hy=0.1, hz=0.08, cfg=cfg)
        else:
            env.add_shape_capsule(b, radius=0.06, half_height=0.12, cfg=cfg)
    env.add_shape_collision_filter_pair(0, 1)
    env.add_shape_collision_filter_pair(3, 7)
    env.add_shape_collision_filter_pair(4, 9)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    scene.add_ground_plane(cfg=nt.ModelBuilder.ShapeConfig(gap=0.02))
    scene.add_shape_box(-1, xform=[0.3, 0.0, 0.4, 0.0, 0.0, 0.0, 1.0], hx=0.1, hy=0.5, hz=0.4,
                        cfg=nt.ModelBuilder.ShapeConfig(collision_group=-3, gap=0.02))
    model = scene.finalize(device=device)
    off = rng.uniform(-0.05, 0.05, size=(model.body_count, 3)).astype(np.float32)
    model.body_q[:, :3] += off
    model.joint_q.reshape(-1, 7)[:, :3] += off
    return model


def _group_pair(a, b):
    if a == 0 or b == 0:
        return False
    if a > 0:
        return a == b or b < 0
    return a != b


def _brute_force(model, lo, hi):
    S = model.shape_count
    world, group, flags = np.asarray(model.shape_world), np.asarray(model.shape_collision_group), np.asarray(model.shape_flags)
    filt = set(model.shape_collision_filter_pairs)
    out = set()
    for a in range(S):
        for b in range(a + 1, S):
            if not (flags[a] & ShapeFlags.COLLIDE_SHAPES and flags[b] & ShapeFlags.COLLIDE_SHAPES):
                continue
            if world[a] != world[b] and world[a] != -1 and world[b] != -1:
                continue
            if not _group_pair(group[a], group[b]) or (a, b) in filt:
                continue
            if np.all(lo[a] <= hi[b]) and np.all(hi[a] >= lo[b]):
                out.add((a, b))
    return out


@pytest.mark.parametrize("world_count,seed", [(1, 0), (4, 1), (7, 2)])
def test_explicit_nxn_and_brute_force_agree(oracle_lib, world_count, seed):
    from oracle_bridge import Oracle

    model = _scene(world_count, seed)
    o = Oracle(model)
    ct = o.contacts()
    pairs_explicit, lo, hi = o.collide(model.body_q, ct, broad_phase="explicit")
    n_explicit = int(ct.count[0])
    pairs_nxn, _, _ = o.collide(model.body_q, ct, broad_phase="nxn")
    want = _brute_force(model, lo, hi)
    # static-vs-static (ground, wall) pairs never enter the builder's pair list nor any per-world NxN segment
    want = {p for p in want if not (model.shape_body[p[0]] < 0 and model.shape_body[p[1]] < 0)}
    got_e = {tuple(p) for p in pairs_explicit}
    got_n = {tuple(p) for p in pairs_nxn if not (model.shape_body[p[0]] < 0 and model.shape_body[p[1]] < 0)}
    assert len(want) >= 2 * world_count
    assert got_e == want
    assert got_n == want
    assert int(ct.count[0]) == n_explicit  # same contacts either way


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["explicit", "nxn", "sap"])
def test_hip_candidate_set_matches_brute_force(mode):
    from oracle_bridge import Oracle

    model = _scene(9, 3, device="cuda:0")
    o = Oracle(model)
    oc = o.contacts()
    _, lo, hi = o.collide(model.body_q, oc)
    want = {p for p in _brute_force(model, lo, hi) if not (model.shape_body[p[0]] < 0 and model.shape_body[p[1]] < 0)}
    pipe = nt.CollisionPipeline(model, broad_phase=mode)
    contacts = pipe.contacts()
    pipe.collide(model.state(), contacts)
    mask = contacts.candidate_pair_mask.cpu().numpy()
    t = model.env
    all_pairs = np.asarray(model.shape_contact_pairs).reshape(t.env_count, t.np, 2)
    got = {tuple(p) for p in all_pairs[mask]}
    assert got == want
    assert int(contacts.rigid_contact_count.cpu().numpy()[0]) == int(oc.count[0])
