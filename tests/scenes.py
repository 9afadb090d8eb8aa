"""Synthetic scenes shared by tests, bench.py and smoke().

The quadruped URDF below restates the *numbers* of newton/examples/assets/quadruped.urdf (13 links, cylinder
colliders, 12 revolute joints) -- geometry/topology data, generated programmatically, not a file copy.
"""
from __future__ import annotations

import numpy as np

import newton_amd as nt

# (leg, HAA x, HAA y, HAA yaw, HFE yaw sign, HFE y offset): newton/examples/assets/quadruped.urdf:24-29,86-91,148-153,210-215 (HAA
# origins; the hind legs are mounted turned by rpy = "0 0 3.1415") and :44-47,106-109,168-171,230-233 (HFE origins; the hind
# legs' y offsets are flipped with the mount)
_LEGS = [("LF", 0.2999, 0.104, "0", 1.0, 0.05), ("RF", 0.2999, -0.104, "0", -1.0, -0.05),
         ("LH", -0.2999, 0.104, "3.1415", 1.0, -0.05), ("RH", -0.2999, -0.104, "3.1415", -1.0, 0.05)]
_HALF_PI = "1.57079632679"


def quadruped_urdf(colliders: str = "cylinder") -> str:
    """Anymal-class stand-in: base cylinder + 4 x {HAA, THIGH, SHANK}.  colliders="box" swaps every
    cylinder for its bounding box (config C4's convex-convex variant)."""
    urdf = _quadruped_urdf_cylinders()
    if colliders == "cylinder":
        return urdf
    if colliders != "box":
        raise ValueError(colliders)
    import re

    def box(m):
        length, radius = float(m.group(1)), float(m.group(2))
        return f'<box size="{2 * radius} {2 * radius} {length}"/>'

    return re.sub(r'<cylinder length="([0-9.]+)" radius="([0-9.]+)"/>', box, urdf)


def _quadruped_urdf_cylinders() -> str:
    out = ['<?xml version="1.0" encoding="utf-8"?>', '<robot name="quadruped">',
           '<link name="base"><collision><origin rpy="0 %s 0" xyz="0 0 0"/><geometry><cylinder length="0.75" radius="0.1"/>'
           '</geometry></collision></link>' % _HALF_PI]
    for leg, x, y, haa_yaw, sgn, hfe_y in _LEGS:
        out.append(f'<joint name="{leg}_HAA" type="revolute"><parent link="base"/><child link="{leg}_HAA"/><axis xyz="1 0 0"/>'
                   f'<limit effort="80.0" velocity="20."/><origin rpy="0 0 {haa_yaw}" xyz="{x} {y} 0.0"/></joint>')
        out.append(f'<link name="{leg}_HAA"><collision><origin rpy="{_HALF_PI} 0 0" xyz="0 0 0"/><geometry>'
                   '<cylinder length="0.05" radius="0.04"/></geometry></collision></link>')
        yaw = _HALF_PI if sgn > 0 else "-" + _HALF_PI
        out.append(f'<joint name="{leg}_HFE" type="revolute"><parent link="{leg}_HAA"/><child link="{leg}_THIGH"/>'
                   f'<origin rpy="0 0 {yaw}" xyz="0 {hfe_y} 0"/><axis xyz="1 0 0"/><limit effort="80.0" velocity="20."/>'
                   '<dynamics damping="0.0" friction="0.0"/></joint>')
        out.append(f'<link name="{leg}_THIGH"><collision><origin rpy="0 0 0" xyz="0 0 -0.125"/><geometry>'
                   '<cylinder length="0.25" radius="0.02"/></geometry></collision></link>')
        out.append(f'<joint name="{leg}_KFE" type="revolute"><parent link="{leg}_THIGH"/><child link="{leg}_SHANK"/>'
                   '<origin rpy="0 0 0" xyz="0 0.0 -0.25"/><axis xyz="1 0 0"/><limit effort="80.0" velocity="20."/>'
                   '<dynamics damping="0.0" friction="0.0"/></joint>')
        out.append(f'<link name="{leg}_SHANK"><collision><origin rpy="0 0 0" xyz="0 0 -0.125"/><geometry>'
                   '<cylinder length="0.25" radius="0.02"/></geometry></collision></link>')
    out.append("</robot>")
    return "\n".join(out)


def quadruped_builder(colliders: str = "cylinder"):
    """The per-world builder of newton/examples/basic/example_basic_urdf.py:38-75 (XPBD branch)."""
    q = nt.ModelBuilder()
    q.default_joint_cfg.armature = 0.01
    q.default_joint_cfg.target_ke = 2000.0
    q.default_joint_cfg.target_kd = 1.0
    q.default_shape_cfg.mu = 1.0
    q.add_urdf(quadruped_urdf(colliders), xform=[0.0, 0.0, 0.7, 0.0, 0.0, 0.0, 1.0], floating=True, enable_self_collisions=False,
               ignore_inertial_definitions=True)
    for b in range(q.body_count):
        q.body_inertia[b] = q.body_inertia[b] + np.eye(3) * 0.01
        # (the example edits body_inertia only; body_inv_inertia keeps the pre-armature value, builder semantics)
    pose = [0.2, 0.4, -0.6, -0.2, -0.4, 0.6, -0.2, 0.4, -0.6, 0.2, -0.4, 0.6]
    q.joint_q[-12:] = pose
    q.joint_target_q[-12:] = pose
    return q


def quadruped_scene(world_count: int, device=None, seed: int | None = 1, height_jitter: float = 0.05,
                    colliders: str = "cylinder", ground: str = "plane", world_range=None):
    """C3/C4 scene: `world_count` quadrupeds + one global ground plane.  Per-env root-height jitter U(0, jitter)
    (seeded) de-correlates the environments.  colliders="box" + ground="box" is C4's convex-convex
    variant: box links on a static 2 m x 2 m x 0.5 m slab whose top face is z = 0 (MPR portals on a 100 m slab are
    conditioned 1e4 : 1 in fp32 -- a 6e-8 pose difference moved its contact normals by 7e-4), so all 13 candidate pairs of an
    environment are box-box and go through MPR/GJK + the manifold (narrow_phase.py:642-655).
    `world_range = (begin, end)`: build only worlds [begin, end) of the `world_count`-world scene -- the same model
    newton_amd.sharding.shard_model cuts out of the global one (the per-world jitter is drawn for all worlds and sliced), without
    paying the Python builder for the other ranks' worlds."""
    total = world_count
    begin, end = (0, world_count) if world_range is None else world_range
    world_count = end - begin
    q = quadruped_builder(colliders)
    scene = nt.ModelBuilder()
    scene.replicate(q, world_count)
    if ground == "plane":
        scene.add_ground_plane(cfg=q.default_shape_cfg)
    elif ground == "box":
        scene.add_shape_box(-1, xform=[0.0, 0.0, -0.25, 0.0, 0.0, 0.0, 1.0], hx=1.0, hy=1.0, hz=0.25, cfg=q.default_shape_cfg)
    else:
        raise ValueError(ground)
    model = scene.finalize(device=device)
    if seed is not None and height_jitter > 0.0:
        rng = np.random.default_rng(seed)
        jq = model.joint_q.reshape(world_count, -1)
        jq[:, 2] += rng.uniform(0.0, height_jitter, size=total).astype(np.float32)[begin:end]
    bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    model.body_q, model.body_qd = bq, bqd
    return model


def quadruped_convex_scene(world_count: int, device=None, seed: int | None = 1, height_jitter: float = 0.05, world_range=None):
    """Config C4, convex-convex variant (box links on a box slab)."""
    return quadruped_scene(world_count, device=device, seed=seed, height_jitter=height_jitter, colliders="box", ground="box",
                           world_range=world_range)


def box_stack_scene(world_count: int, n_boxes: int = 8, device=None, seed: int | None = 0, jitter: float = 1e-3):
    """C2 scene: `n_boxes` boxes (h = 0.5) stacked at z = 0.5 + k on an infinite plane
    (pattern of newton/tests/test_solver_xpbd.py:1798-1804)."""
    env = nt.ModelBuilder()
    for k in range(n_boxes):
        b = env.add_body(xform=[0.0, 0.0, 0.5 + k, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_box(b, hx=0.5, hy=0.5, hz=0.5)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    scene.add_ground_plane()
    model = scene.finalize(device=device)
    if seed is not None and jitter > 0.0:
        rng = np.random.default_rng(seed)
        off = rng.uniform(-jitter, jitter, size=(model.body_count, 2)).astype(np.float32)
        model.body_q[:, :2] += off
        model.joint_q.reshape(-1, 7)[:, :2] += off
    return model


def mixed_primitive_scene(world_count: int, device=None, seed: int = 3):
    """Free bodies with sphere / capsule / box / cylinder / ellipsoid colliders dropped near a ground plane:
    exercises every analytic plane-* pair plus sphere-sphere / sphere-capsule / capsule-capsule / sphere-box."""
    rng = np.random.default_rng(seed)
    env = nt.ModelBuilder()
    specs = ["sphere", "capsule", "box", "cylinder", "ellipsoid", "sphere", "capsule"]
    for k, kind in enumerate(specs):
        ang = rng.uniform(-1.0, 1.0, size=3)
        q = nt._np_math.quat_rpy(*ang)
        b = env.add_body(xform=[0.35 * (k % 3) - 0.3, 0.4 * (k // 3) - 0.3, 0.12 + 0.02 * k, *q])
        if kind == "sphere":
            env.add_shape_sphere(b, radius=0.1)
        elif kind == "capsule":
            env.add_shape_capsule(b, radius=0.07, half_height=0.15)
        elif kind == "box":
            env.add_shape_box(b, hx=0.1, hy=0.08, hz=0.06)
        elif kind == "cylinder":
            env.add_shape_cylinder(b, radius=0.08, half_height=0.1)
        else:
            env.add_shape_ellipsoid(b, rx=0.12, ry=0.08, rz=0.06)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    scene.add_ground_plane()
    model = scene.finalize(device=device)
    off = rng.uniform(-0.02, 0.02, size=(model.body_count, 3)).astype(np.float32)
    model.body_q[:, :3] += off
    model.joint_q.reshape(-1, 7)[:, :3] += off
    return model


def tiled_floor_scene(world_count: int, tiles: int = 600, device=None, seed: int = 5, per_env: int = 2):
    """`tiles` static boxes (world -1: global shapes, one row along +x, 0.2 m apart) and `per_env` free spheres per world resting on
    tiles spread over the whole row -- more global shapes than any tile kernel's workgroup has lanes (64 ... 512), so the
    per-launch staging of the global shapes' world data (nt_collide.hpp: stage_global_world) has to loop."""
    rng = np.random.default_rng(seed)
    env = nt.ModelBuilder()
    for _ in range(per_env):
        b = env.add_body(xform=[0.0, 0.0, 0.145, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_sphere(b, radius=0.05)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    for k in range(tiles):
        scene.add_shape_box(-1, xform=[0.2 * k, 0.0, 0.05, 0.0, 0.0, 0.0, 1.0], hx=0.09, hy=0.09, hz=0.05)
    model = scene.finalize(device=device)
    # every sphere over its own tile: the first / last tiles and a spread in between (indices beyond every workgroup size included)
    idx = rng.integers(0, tiles, size=model.body_count)
    idx[0], idx[-1] = tiles - 1, 0
    off = rng.uniform(-0.03, 0.03, size=(model.body_count, 2)).astype(np.float32)
    model.body_q[:, 0] = (0.2 * idx).astype(np.float32) + off[:, 0]
    model.body_q[:, 1] = off[:, 1]
    model.joint_q.reshape(-1, 7)[:, :2] = model.body_q[:, :2]
    return model


def pendulum_scene(world_count: int, device=None, seed: int | None = None):
    """C1 scene: double pendulum of newton/examples/basic/example_basic_pendulum.py:34-64 (2 box links hx=1, hy=hz=0.1,
    revolute-Y joints, first anchor at (0,0,5) rotated -90 deg about Z) + ground plane; optional per-env joint-angle jitter."""
    from newton_amd import _np_math as nm

    hx, hy, hz = 1.0, 0.1, 0.1
    env = nt.ModelBuilder()
    l0 = env.add_link()
    env.add_shape_box(l0, hx=hx, hy=hy, hz=hz)
    l1 = env.add_link()
    env.add_shape_box(l1, hx=hx, hy=hy, hz=hz)
    rot = nm.quat_from_axis_angle([0.0, 0.0, 1.0], -np.pi * 0.5)
    j0 = env.add_joint_revolute(-1, l0, axis=[0.0, 1.0, 0.0], parent_xform=nm.transform([0.0, 0.0, 5.0], rot),
                                child_xform=nm.transform([-hx, 0.0, 0.0]))
    j1 = env.add_joint_revolute(l0, l1, axis=[0.0, 1.0, 0.0], parent_xform=nm.transform([hx, 0.0, 0.0]),
                                child_xform=nm.transform([-hx, 0.0, 0.0]))
    env.add_articulation([j0, j1])
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    scene.add_ground_plane()
    model = scene.finalize(device=device)
    if seed is not None:
        rng = np.random.default_rng(seed)
        model.joint_q[:] = rng.uniform(-0.5, 0.5, size=model.joint_q.shape).astype(np.float32)
        model.joint_qd[:] = rng.uniform(-1.0, 1.0, size=model.joint_qd.shape).astype(np.float32)
    bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
    model.body_q, model.body_qd = bq, bqd
    return model


def joint_zoo_scene(world_count: int, device=None, seed: int | None = 0, free_root: bool = False):
    """One articulation per env exercising every supported joint type:
    (world | FREE) -> revolute -> prismatic -> ball -> fixed -> D6(linear x, angular z), plus limits, drives and damping."""
    from newton_amd import _np_math as nm

    env = nt.ModelBuilder()
    cfg = nt.ModelBuilder.ShapeConfig(has_shape_collision=False)
    links = []
    for k in range(6):
        b = env.add_link(xform=[0.3 * k, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_box(b, hx=0.12, hy=0.05 + 0.01 * k, hz=0.04, cfg=cfg)
        links.append(b)
    X = lambda p, q=(0.0, 0.0, 0.0, 1.0): nm.transform(p, q)  # noqa: E731
    joints = []
    if free_root:
        joints.append(env.add_joint_free(links[0]))
    else:
        joints.append(env.add_joint_revolute(-1, links[0], axis=[0.0, 1.0, 0.0], parent_xform=X([0.0, 0.0, 1.0]),
                                             child_xform=X([-0.15, 0.0, 0.0]), target_ke=50.0, target_kd=2.0))
    joints.append(env.add_joint_revolute(links[0], links[1], axis=[0.0, 0.0, 1.0], parent_xform=X([0.15, 0.0, 0.0]),
                                         child_xform=X([-0.15, 0.0, 0.0]), limit_lower=-0.2, limit_upper=0.4,
                                         target_ke=100.0, target_kd=1.0, armature=0.01))
    joints.append(env.add_joint_prismatic(links[1], links[2], axis=[1.0, 0.0, 0.0], parent_xform=X([0.15, 0.0, 0.0]),
                                          child_xform=X([-0.15, 0.0, 0.0]), limit_lower=-0.05, limit_upper=0.1,
                                          target_ke=200.0, target_kd=5.0, armature=0.02))
    joints.append(env.add_joint_ball(links[2], links[3], parent_xform=X([0.15, 0.0, 0.0]), child_xform=X([-0.15, 0.0, 0.0])))
    joints.append(env.add_joint_fixed(links[3], links[4], parent_xform=X([0.15, 0.0, 0.0], nm.quat_rpy(0.1, 0.0, 0.2)),
                                      child_xform=X([-0.15, 0.0, 0.0])))
    D = nt.ModelBuilder.JointDofConfig
    joints.append(env.add_joint_d6(links[4], links[5], linear_axes=[D(axis=0, limit_lower=-0.05, limit_upper=0.05, armature=0.01)],
                                   angular_axes=[D(axis=2, target_ke=20.0, target_kd=0.5, armature=0.01)],
                                   parent_xform=X([0.15, 0.0, 0.0]), child_xform=X([-0.15, 0.0, 0.0])))
    env.add_articulation(joints)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    model = scene.finalize(device=device)
    if seed is not None:
        rng = np.random.default_rng(seed)
        E = world_count
        jq = model.joint_q.reshape(E, -1).copy()
        jqd = rng.normal(0.0, 0.5, size=model.joint_qd.shape).astype(np.float32)
        t = model.env
        for j in range(t.nj):
            qs, jt = int(t.joint_q_start[j]), int(t.joint_type[j])
            if jt == nt.JointType.BALL:
                q = rng.normal(size=(E, 4)) * 0.2 + np.array([0, 0, 0, 1.0])
                jq[:, qs:qs + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
            elif jt == nt.JointType.FREE:
                q = rng.normal(size=(E, 4)) * 0.3 + np.array([0, 0, 0, 1.0])
                jq[:, qs + 3:qs + 7] = q / np.linalg.norm(q, axis=1, keepdims=True)
                jq[:, qs:qs + 3] += rng.normal(0, 0.05, size=(E, 3))
            elif jt != nt.JointType.FIXED:
                n = int(t.joint_lin_count[j] + t.joint_ang_count[j])
                jq[:, qs:qs + n] = rng.uniform(-0.15, 0.15, size=(E, n))
        model.joint_q = jq.reshape(-1).astype(np.float32)
        model.joint_qd = jqd
        bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
        model.body_q, model.body_qd = bq, bqd
    return model


def free_child_scene(world_count: int, device=None, seed: int | None = 0, free_root: bool = False):
    """FREE and DISTANCE joints below the articulation root (SolverFeatherstone keeps them in internal parent-origin
    coordinates and re-integrates the child pose at the end of the step, solver_featherstone.py:229-265,1006-1046):
    (world revolute | FREE) -> link0;  link0 -FREE-> link1 -revolute-> link2;  link0 -DISTANCE-> link3 -prismatic-> link4.
    Off-centre COMs and anchors on both sides so every offset in the conversions is exercised."""
    from newton_amd import _np_math as nm

    env = nt.ModelBuilder()
    cfg = nt.ModelBuilder.ShapeConfig(has_shape_collision=False)
    links = []
    for k in range(5):
        b = env.add_link(xform=[0.35 * k, 0.1 * (k % 2), 1.0 + 0.05 * k, 0.0, 0.0, 0.0, 1.0])
        env.add_shape_box(b, xform=nm.transform([0.03 * (k - 2), 0.02, -0.01 * k]), hx=0.12, hy=0.05 + 0.01 * k, hz=0.04 + 0.005 * k, cfg=cfg)
        links.append(b)
    X = lambda p, q=(0.0, 0.0, 0.0, 1.0): nm.transform(p, q)  # noqa: E731
    joints = []
    if free_root:
        joints.append(env.add_joint_free(links[0]))
    else:
        joints.append(env.add_joint_revolute(-1, links[0], axis=[0.0, 1.0, 0.0], parent_xform=X([0.0, 0.0, 1.0]),
                                             child_xform=X([-0.15, 0.0, 0.0]), target_ke=20.0, target_kd=1.0))
    joints.append(env.add_joint_free(links[1], parent=links[0], parent_xform=X([0.1, 0.02, 0.0], nm.quat_rpy(0.2, -0.1, 0.3)),
                                     child_xform=X([-0.05, 0.01, 0.02], nm.quat_rpy(0.0, 0.15, -0.1))))
    joints.append(env.add_joint_revolute(links[1], links[2], axis=[0.0, 0.0, 1.0], parent_xform=X([0.15, 0.0, 0.0]),
                                         child_xform=X([-0.15, 0.0, 0.0]), target_ke=30.0, target_kd=0.5, armature=0.01))
    joints.append(env.add_joint_distance(links[0], links[3], parent_xform=X([-0.1, 0.0, 0.05]), child_xform=X([0.02, 0.0, 0.0]),
                                         min_distance=-1.0, max_distance=-1.0))
    joints.append(env.add_joint_prismatic(links[3], links[4], axis=[1.0, 0.0, 0.0], parent_xform=X([0.15, 0.0, 0.0]),
                                          child_xform=X([-0.15, 0.0, 0.0]), limit_lower=-0.05, limit_upper=0.1, target_ke=100.0,
                                          target_kd=2.0, armature=0.02))
    env.add_articulation(joints)
    scene = nt.ModelBuilder()
    scene.replicate(env, world_count)
    model = scene.finalize(device=device)
    if seed is not None:
        rng = np.random.default_rng(seed)
        E = world_count
        jq = model.joint_q.reshape(E, -1).copy()
        jqd = rng.normal(0.0, 0.5, size=model.joint_qd.shape).astype(np.float32)
        t = model.env
        for j in range(t.nj):
            qs, jt = int(t.joint_q_start[j]), int(t.joint_type[j])
            if jt in (nt.JointType.FREE, nt.JointType.DISTANCE):
                q = rng.normal(size=(E, 4)) * 0.3 + np.array([0, 0, 0, 1.0])
                jq[:, qs + 3:qs + 7] = q / np.linalg.norm(q, axis=1, keepdims=True)
                jq[:, qs:qs + 3] += rng.normal(0, 0.05, size=(E, 3))
            else:
                jq[:, qs:qs + 1] = rng.uniform(-0.15, 0.15, size=(E, 1))
        model.joint_q = jq.reshape(-1).astype(np.float32)
        model.joint_qd = jqd
        bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
        model.body_q, model.body_qd = bq, bqd
    return model


def hull_bin_scene(world_count: int, n_hulls: int = 64, device=None, seed: int = 2, jitter: float = 0.005, mu=None,
                   hull_pairs: bool = True, shape_cfg=None, inertia_armature: float = 0.0, sdf: bool = False,
                   sdf_resolution: int = 24, hydroelastic: bool = False, kh: float = 1.0e10):
    """Config C5 without the SDF / hydroelastic contact models: `n_hulls` random convex hulls (16-32 vertices, radius
    U(0.03, 0.06)) dropped into a five-wall bin (ground plane + four static boxes); every hull pair and every hull-wall pair
    is a candidate, so one environment has n(n-1)/2 + 5n pairs (2 336 for 64 hulls) and its per-contact solver records no
    longer fit LDS (nt_model.contact_scratch_in_hbm).  Every environment shares the hull set; `jitter` moves each hull of
    each environment by U(-jitter, jitter) m in x, y, z (seeded), so the environments' contact sets differ.  More than 32
    environments are produced by tiling a 32-environment build (newton_amd.worlds.tile_worlds) -- the Python builder
    would spend minutes on 131 072 bodies.

    `inertia_armature` adds that much to the diagonal of every hull's inertia (like the + 0.01 of the reference's quadruped
    example): the lightest hull weighs 25 g with a smallest principal inertia of 2.8e-6 kg m^2, and the EXPLICIT penalty
    contacts of SolverSemiImplicit / Featherstone are only stable while kf * r^2 * dt / I < 2 (kf = 200, r = 0.04 m,
    dt = 1/4000 s gives 28 on the bare hull: it spins up to 200 rad/s on the ground before any hull touches another --
    measured, profiles/r02h_sdf_stage_sweep.jsonl).  XPBD (implicit positions) needs none."""
    from newton_amd.worlds import slice_worlds, tile_worlds

    # sdf=True: config C5 with its contact model -- every hull carries a uint16 texture SDF (Mesh.build_sdf) and the five walls of
    # the bin are static boxes with generated SDFs (ShapeConfig.configure_sdf), so EVERY pair (hull-hull, hull-wall) takes the
    # mesh-SDF leg of CollisionPipeline.collide and the environment tiles carry no pair at all
    rng = np.random.default_rng(seed)
    env = nt.ModelBuilder()
    if mu is not None:
        env.default_shape_cfg.mu = float(mu)
    for k, v in (shape_cfg or {}).items():
        setattr(env.default_shape_cfg, k, v)
    if hydroelastic:  # config C5's second contact model: every shape HYDROELASTIC, kh = 1e10 (builder.py:571)
        env.default_shape_cfg.is_hydroelastic, env.default_shape_cfg.kh = True, float(kh)
    side = 0.07 * np.ceil(np.sqrt(n_hulls))
    # hulls start on a lattice with 0.135 m pitch (> twice the largest hull radius): no initial interpenetration -- randomly
    # overlapping hulls make XPBD eject them at 10^3 rad/s (oracle and device alike) until the state overflows
    n_side = int(np.ceil(n_hulls ** (1.0 / 3.0)))
    pitch = 0.135
    for k in range(n_hulls):
        pts = rng.normal(size=(int(rng.integers(16, 33)), 3))
        pts *= rng.uniform(0.03, 0.06) / np.linalg.norm(pts, axis=1).max()
        ix, iy, iz = k % n_side, (k // n_side) % n_side, k // (n_side * n_side)
        pos = [(ix - 0.5 * (n_side - 1)) * pitch, (iy - 0.5 * (n_side - 1)) * pitch, 0.08 + iz * pitch]
        b = env.add_body(xform=[*pos, *nt._np_math.quat_rpy(*rng.uniform(-1.0, 1.0, size=3))])
        mesh = nt.Mesh.convex_hull_of(pts)
        if sdf:
            mesh.build_sdf(max_resolution=sdf_resolution, margin=0.02, narrow_band_range=(-0.1, 0.1), texture_format="uint16")
        env.add_shape_convex_hull(b, mesh=mesh)
    if inertia_armature > 0.0:
        for b in range(env.body_count):
            env.body_inertia[b] = env.body_inertia[b] + np.eye(3) * inertia_armature
            env.body_inv_inertia[b] = np.linalg.inv(env.body_inertia[b])
    if not hull_pairs:  # hull-hull contacts come from somewhere else (the mesh-SDF stage): only hull-wall pairs stay in the tiles
        for a in range(n_hulls):
            for b2 in range(a + 1, n_hulls):
                env.add_shape_collision_filter_pair(a, b2)
    base_count = min(world_count, 32)
    scene = nt.ModelBuilder()
    scene.replicate(env, base_count)
    w = 0.5 * side + 0.1
    wall_cfg = None
    if sdf:
        wall_cfg = scene.default_shape_cfg.copy()
        for k, v in (shape_cfg or {}).items():
            setattr(wall_cfg, k, v)
        if mu is not None:
            wall_cfg.mu = float(mu)
        wall_cfg.configure_sdf(max_resolution=64, is_hydroelastic=hydroelastic, kh=float(kh))
        scene.add_shape_box(-1, xform=[0.0, 0.0, -0.05, 0.0, 0.0, 0.0, 1.0], hx=w + 0.06, hy=w + 0.06, hz=0.05, cfg=wall_cfg)
    else:
        scene.add_ground_plane()
    for sx, sy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        scene.add_shape_box(-1, xform=[w * sx, w * sy, 0.3, 0.0, 0.0, 0.0, 1.0], hx=0.03 if sx else w + 0.03,
                            hy=0.03 if sy else w + 0.03, hz=0.3, cfg=wall_cfg)
    model = scene.finalize(device=device)
    if world_count > base_count:
        reps = -(-world_count // base_count)
        model = tile_worlds(model, reps, device=device, filter_pairs=False)
        if model.world_count != world_count:
            model = slice_worlds(model, 0, world_count, device=device)
    if jitter > 0.0:
        off = np.random.default_rng(seed + 1000).uniform(-jitter, jitter, size=(model.body_count, 3)).astype(np.float32)
        model.body_q[:, :3] += off
        model.joint_q.reshape(-1, 7)[:, :3] += off
    return model


def mesh_ground_scene(world_count: int, n_meshes: int = 8, device=None, seed: int = 4, jitter: float = 0.01, gap: float = 0.004):
    """`n_meshes` triangle-mesh bodies per world resting on the infinite ground plane, spaced so that they never meet (the mesh-vs-
    mesh / mesh-vs-primitive legs are not built): alternately a UV sphere of 32 x 32 = 994 vertices and a box whose corners appear
    three times (per-face vertices of a render mesh).  Every (mesh, plane) pair goes through the vertex leg of
    CollisionPipeline.collide; worlds differ by a seeded pose jitter.  bench.py --workload mesh_ground."""
    import newton_amd as nt

    rng = np.random.default_rng(seed)
    hull = nt.Mesh.create_box(0.1, 0.08, 0.05)
    box = nt.Mesh(np.concatenate([hull.vertices] * 3), hull.indices)
    sphere = nt.Mesh.create_sphere(0.08, 32, 32)
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = gap
    env.default_shape_cfg.mu = 0.5
    side = int(np.ceil(np.sqrt(n_meshes)))
    shapes = []
    for k in range(n_meshes):
        mesh, h = (sphere, 0.08) if k % 2 == 0 else (box, 0.05)
        b = env.add_body(xform=[0.5 * (k % side), 0.5 * (k // side), h - 0.0005, 0.0, 0.0, 0.0, 1.0])
        shapes.append(env.add_shape_mesh(b, mesh=mesh))
    for i in range(n_meshes):  # the meshes are too far apart to touch: no mesh-mesh candidate pairs at all
        for j in range(i + 1, n_meshes):
            env.add_shape_collision_filter_pair(shapes[i], shapes[j])
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = gap
    scene.replicate(env, world_count)
    scene.add_ground_plane()
    model = scene.finalize(device=device)
    if jitter > 0.0:
        off = rng.uniform(-jitter, jitter, size=(model.body_count, 2)).astype(np.float32)
        model.body_q[:, :2] += off
        model.joint_q.reshape(-1, 7)[:, :2] += off
    return model


def terrain_height(x, y):
    return 0.012 * np.sin(4.0 * x) * np.cos(3.0 * y)


def terrain_scene(world_count: int, n_shapes: int = 8, device=None, seed: int = 6, cells: int = 64, half: float = 1.6, gap: float = 0.004,
                  heightfield: bool = False):
    """`n_shapes` convex primitives per world (box, sphere, capsule, cylinder in turn) resting on ONE shared static terrain: a triangle
    mesh of cells x cells x 2 triangles (a global shape without an SDF).  Every (primitive, terrain) pair goes through the triangle leg
    of CollisionPipeline.collide (narrow_phase.py:633-638,1455-1665: midphase over the mesh's triangles, GJK / MPR + manifold per
    triangle, the global contact reduction); the primitives are filtered against each other.  Worlds differ by a seeded pose jitter.
    bench.py --workload terrain; heightfield=True: the same surface as a GeoType.HFIELD shape (--workload terrain_hfield)."""
    import newton_amd as nt

    rng = np.random.default_rng(seed)
    xs = np.linspace(-half, half, cells + 1)
    pts = np.array([(x, y, terrain_height(x, y)) for y in xs for x in xs], np.float32)
    idx = []
    for j in range(cells):
        for i in range(cells):
            a, b, c, d = j * (cells + 1) + i, j * (cells + 1) + i + 1, (j + 1) * (cells + 1) + i, (j + 1) * (cells + 1) + i + 1
            idx += [a, b, d, a, d, c]
    terrain = nt.Mesh(pts, np.array(idx, np.int32))
    env = nt.ModelBuilder()
    env.default_shape_cfg.gap = gap
    env.default_shape_cfg.mu = 0.5
    side = int(np.ceil(np.sqrt(n_shapes)))
    shapes, halves = [], []
    for k in range(n_shapes):
        b = env.add_body(xform=[0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0])
        kind = k % 4
        if kind == 0:
            shapes.append(env.add_shape_box(b, hx=0.08, hy=0.06, hz=0.05)); halves.append(0.05)
        elif kind == 1:
            shapes.append(env.add_shape_sphere(b, radius=0.06)); halves.append(0.06)
        elif kind == 2:
            shapes.append(env.add_shape_capsule(b, radius=0.04, half_height=0.08)); halves.append(0.04)
        else:
            shapes.append(env.add_shape_cylinder(b, radius=0.05, half_height=0.05)); halves.append(0.05)
    for i in range(n_shapes):
        for j in range(i + 1, n_shapes):
            env.add_shape_collision_filter_pair(shapes[i], shapes[j])
    scene = nt.ModelBuilder()
    scene.default_shape_cfg.gap = gap
    scene.default_shape_cfg.mu = 0.5
    scene.replicate(env, world_count)
    if heightfield:  # the same surface as a newton.Heightfield (GeoType.HFIELD): grid-cell midphase, TRIANGLE_PRISM cells
        raw = np.array([[terrain_height(x, y) for x in xs] for y in xs], np.float32)
        scene.add_shape_heightfield(heightfield=nt.Heightfield(raw, cells + 1, cells + 1, hx=half, hy=half))
    else:
        scene.add_shape_mesh(-1, mesh=terrain)
    model = scene.finalize(device=device)
    pitch = 2.0 * half / (side + 1)
    for w in range(world_count):
        for k in range(n_shapes):
            x = -half + pitch * (1 + k % side) + rng.uniform(-0.05, 0.05)
            y = -half + pitch * (1 + k // side) + rng.uniform(-0.05, 0.05)
            q = nt._np_math.quat_rpy(0.0, np.pi / 2 if k % 4 == 2 else 0.0, rng.uniform(-1.0, 1.0))  # capsules lie on their side
            i = w * n_shapes + k
            model.body_q[i, :3] = [x, y, terrain_height(x, y) + halves[k] + 0.006]  # (a 6 mm drop: the slopes are gentle, 4 %)
            model.body_q[i, 3:] = q
            model.joint_q.reshape(-1, 7)[i] = model.body_q[i]
    return model
