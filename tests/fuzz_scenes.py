"""Random replicated scenes for the kernel-vs-oracle fuzz tests: random joint trees over every supported joint type, several
shapes per body (or none), every primitive / hull type, static world shapes, collision groups, filter pairs, kinematic and
massless links, disabled joints, per-world parameter jitter and random initial state."""
import numpy as np

import newton_amd as nt
from newton_amd import _np_math as nm

D = nt.ModelBuilder.JointDofConfig


def _rand_quat(rng, spread=1.0):
    return nm.quat_rpy(*(rng.uniform(-spread, spread, size=3)))


def _add_random_shape(env, rng, body, cfg, allow_hull=True):
    kind = rng.choice(["sphere", "box", "capsule", "cylinder", "ellipsoid", "cone", "hull"] if allow_hull else
                      ["sphere", "box", "capsule", "cylinder", "ellipsoid"])
    xf = [*rng.uniform(-0.05, 0.05, size=3), *_rand_quat(rng, 0.6)]
    if kind == "sphere":
        env.add_shape_sphere(body, xform=xf, radius=rng.uniform(0.05, 0.12), cfg=cfg)
    elif kind == "box":
        env.add_shape_box(body, xform=xf, hx=rng.uniform(0.04, 0.12), hy=rng.uniform(0.04, 0.12), hz=rng.uniform(0.04, 0.12), cfg=cfg)
    elif kind == "capsule":
        env.add_shape_capsule(body, xform=xf, radius=rng.uniform(0.03, 0.07), half_height=rng.uniform(0.05, 0.15), cfg=cfg)
    elif kind == "cylinder":
        env.add_shape_cylinder(body, xform=xf, radius=rng.uniform(0.04, 0.09), half_height=rng.uniform(0.05, 0.12), cfg=cfg)
    elif kind == "ellipsoid":
        env.add_shape_ellipsoid(body, xform=xf, rx=rng.uniform(0.05, 0.12), ry=rng.uniform(0.04, 0.1), rz=rng.uniform(0.03, 0.08), cfg=cfg)
    elif kind == "cone":
        env.add_shape_cone(body, xform=xf, radius=rng.uniform(0.05, 0.1), half_height=rng.uniform(0.05, 0.12), cfg=cfg)
    else:
        pts = rng.normal(size=(int(rng.integers(8, 20)), 3)) * rng.uniform(0.04, 0.09)
        env.add_shape_convex_hull(body, xform=xf, mesh=nt.Mesh.convex_hull_of(pts), cfg=cfg)


def random_scene(seed, world_count=None, articulated=True, allow_hull=True, featherstone_compatible=False, param_jitter=False,
                 extras=False):
    rng = np.random.default_rng(seed)
    env = nt.ModelBuilder(gravity=(0.0, 0.0, -9.81) if rng.random() < 0.8 else tuple(rng.normal(0, 5.0, size=3)))
    nb = int(rng.integers(2, 7))
    bodies = []
    for k in range(nb):
        pos = [rng.uniform(-0.35, 0.35), rng.uniform(-0.35, 0.35), rng.uniform(0.08, 0.5)]
        b = env.add_link(xform=[*pos, *_rand_quat(rng)]) if articulated else env.add_body(xform=[*pos, *_rand_quat(rng)])
        bodies.append(b)
        cfg = nt.ModelBuilder.ShapeConfig(collision_group=int(rng.choice([1, 1, 1, 2, -1, -2, 0])), gap=float(rng.uniform(0.0, 0.05)),
                                          mu=float(rng.uniform(0.0, 1.2)), margin=float(rng.choice([0.0, 0.0, 0.01])),
                                          density=float(rng.uniform(300.0, 2000.0)))
        for _ in range(int(rng.choice([1, 1, 1, 2, 0]))):
            _add_random_shape(env, rng, b, cfg, allow_hull)
        if not env.body_shapes[b]:  # massless link: give it inertia so that it can be simulated
            env.body_mass[b] = 0.3
            env.body_inertia[b] = np.eye(3) * 2e-3
    split = int(rng.integers(1, nb)) if (extras and nb >= 3) else nb  # bodies [split, nb) form a second articulation
    if extras and rng.random() < 0.5:
        env.body_flags[bodies[0]] = int(nt.BodyFlags.KINEMATIC)
    if extras:
        for _ in range(int(rng.integers(1, 3))):  # static shapes that belong to the world (body -1) of every env
            cfg = nt.ModelBuilder.ShapeConfig(collision_group=int(rng.choice([1, -1, -2])), gap=float(rng.uniform(0.0, 0.03)))
            xf = [rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(0.0, 0.2), *_rand_quat(rng, 0.4)]
            if rng.random() < 0.5:
                env.add_shape_box(-1, xform=xf, hx=0.15, hy=0.1, hz=0.05, cfg=cfg)
            else:
                env.add_shape_sphere(-1, xform=xf, radius=0.1, cfg=cfg)
    if articulated:
        joints = []
        for k, b in enumerate(bodies):
            first = 0 if k < split else split
            parent = -1 if k == first else int(bodies[rng.integers(first, k)])
            Xp = [*rng.uniform(-0.15, 0.15, size=3), *_rand_quat(rng, 0.5)]
            Xc = [*rng.uniform(-0.1, 0.1, size=3), *_rand_quat(rng, 0.5)]
            if k == first:
                kind = rng.choice(["free", "revolute", "fixed", "distance"] if featherstone_compatible else ["free", "revolute", "fixed"])
                if int(env.body_flags[b]) & int(nt.BodyFlags.KINEMATIC):
                    kind = rng.choice(["free", "fixed"])
            else:
                kinds = ["revolute", "prismatic", "ball", "fixed", "d6", "d6_full"]
                if not featherstone_compatible:
                    kinds += ["distance"]
                kind = rng.choice(kinds)
            axis = rng.normal(size=3)
            axis /= np.linalg.norm(axis)
            lim = dict(limit_lower=float(rng.uniform(-0.6, -0.1)), limit_upper=float(rng.uniform(0.1, 0.6))) if rng.random() < 0.6 else {}
            drive = dict(target_ke=float(rng.uniform(10.0, 200.0)), target_kd=float(rng.uniform(0.1, 3.0))) if rng.random() < 0.5 else {}
            arm = dict(armature=float(rng.uniform(0.0, 0.02)))
            if kind == "free":
                j = env.add_joint_free(b)
            elif kind == "revolute":
                j = env.add_joint_revolute(parent, b, axis=axis, parent_xform=Xp, child_xform=Xc, **lim, **drive, **arm)
            elif kind == "prismatic":
                j = env.add_joint_prismatic(parent, b, axis=axis, parent_xform=Xp, child_xform=Xc, **lim, **drive, **arm)
            elif kind == "ball":
                j = env.add_joint_ball(parent, b, parent_xform=Xp, child_xform=Xc)
            elif kind == "fixed":
                j = env.add_joint_fixed(parent, b, parent_xform=Xp, child_xform=Xc)
            elif kind == "distance":
                j = env.add_joint_distance(parent, b, parent_xform=Xp, child_xform=Xc, min_distance=0.05, max_distance=0.4)
            elif kind == "d6":
                j = env.add_joint_d6(parent, b, linear_axes=[D(axis=int(rng.integers(0, 3)), limit_lower=-0.1, limit_upper=0.1)],
                                     angular_axes=[D(axis=int(rng.integers(0, 3)), target_ke=20.0, target_kd=0.5)],
                                     parent_xform=Xp, child_xform=Xc)
            else:
                j = env.add_joint_d6(parent, b, linear_axes=[D(axis=0, limit_lower=-0.05, limit_upper=0.05)],
                                     angular_axes=[D(axis=0), D(axis=1, limit_lower=-0.3, limit_upper=0.3), D(axis=2)],
                                     parent_xform=Xp, child_xform=Xc)
            joints.append(j)
        env.add_articulation(joints[:split])
        if split < nb:
            env.add_articulation(joints[split:])
        if not featherstone_compatible and rng.random() < 0.3 and len(joints) > 2:
            env.joint_enabled[joints[-1]] = False
    if rng.random() < 0.4 and env.shape_count >= 3:
        a, c = rng.choice(env.shape_count, size=2, replace=False)
        env.add_shape_collision_filter_pair(int(a), int(c))
    E = int(rng.integers(1, 20)) if world_count is None else world_count
    scene = nt.ModelBuilder(gravity=env._gravity_vector())
    ground_first = rng.random() < 0.3
    if ground_first:
        scene.add_ground_plane()
    scene.replicate(env, E)
    if not ground_first:
        scene.add_ground_plane()
    if rng.random() < 0.5:
        scene.add_shape_box(-1, xform=[0.5, 0.0, 0.2, *_rand_quat(rng, 0.3)], hx=0.1, hy=0.6, hz=0.3,
                            cfg=nt.ModelBuilder.ShapeConfig(collision_group=-3))
    model = scene.finalize()
    # per-world parameter jitter and random state
    model.body_mass *= rng.uniform(0.8, 1.25, size=model.body_mass.shape).astype(np.float32)
    model.body_inv_mass = np.where(model.body_mass > 0, 1.0 / np.maximum(model.body_mass, 1e-12), 0.0).astype(np.float32)
    model.shape_material_mu *= rng.uniform(0.7, 1.3, size=model.shape_material_mu.shape).astype(np.float32)
    if param_jitter:  # domain randomisation: every world gets its own geometry / frames / gains / gravity
        S = model.shape_count
        sc = rng.uniform(0.8, 1.25, size=(S, 1)).astype(np.float32)
        sc[np.asarray(model.shape_type) == int(nt.GeoType.PLANE)] = 1.0
        model.shape_scale = (model.shape_scale * sc).astype(np.float32)
        # the local AABBs carry the scale (builder.py:11575-11612): keep them consistent, like a re-finalize would
        model.shape_collision_aabb_lower = (model.shape_collision_aabb_lower * sc).astype(np.float32)
        model.shape_collision_aabb_upper = (model.shape_collision_aabb_upper * sc).astype(np.float32)
        model.shape_gap = (model.shape_gap * rng.uniform(0.5, 1.5, size=S)).astype(np.float32)
        model.shape_margin = (model.shape_margin * rng.uniform(0.5, 1.5, size=S)).astype(np.float32)
        loc = np.asarray(model.shape_body) >= 0
        model.shape_transform[loc, :3] += rng.normal(0, 0.01, size=(int(loc.sum()), 3)).astype(np.float32)
        model.body_com += rng.normal(0, 0.01, size=model.body_com.shape).astype(np.float32)
        if model.joint_count:
            model.joint_X_p[:, :3] += rng.normal(0, 0.01, size=(model.joint_count, 3)).astype(np.float32)
            model.joint_limit_lower -= rng.uniform(0, 0.05, size=model.joint_limit_lower.shape).astype(np.float32)
            model.joint_target_ke *= rng.uniform(0.5, 2.0, size=model.joint_target_ke.shape).astype(np.float32)
        model.gravity[:-1] += rng.normal(0, 1.0, size=model.gravity[:-1].shape).astype(np.float32)
    if articulated:
        t = model.env
        jq = model.joint_q.reshape(E, -1).copy()
        for j in range(t.nj):
            qs, jt = int(t.joint_q_start[j]), int(t.joint_type[j])
            if jt == nt.JointType.BALL:
                q = rng.normal(size=(E, 4)) * 0.2 + np.array([0, 0, 0, 1.0])
                jq[:, qs:qs + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
            elif jt in (nt.JointType.FREE, nt.JointType.DISTANCE):
                q = rng.normal(size=(E, 4)) * 0.3 + np.array([0, 0, 0, 1.0])
                jq[:, qs + 3:qs + 7] = q / np.linalg.norm(q, axis=1, keepdims=True)
                jq[:, qs:qs + 3] += rng.normal(0, 0.03, size=(E, 3))
            elif jt != nt.JointType.FIXED:
                n = int(t.joint_lin_count[j] + t.joint_ang_count[j])
                jq[:, qs:qs + n] = rng.uniform(-0.2, 0.2, size=(E, n))
        model.joint_q = jq.reshape(-1).astype(np.float32)
        model.joint_qd = rng.normal(0.0, 0.4, size=model.joint_qd.shape).astype(np.float32)
        bq, bqd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
        model.body_q, model.body_qd = bq, bqd
    else:
        model.body_qd = rng.normal(0.0, 0.4, size=model.body_qd.shape).astype(np.float32)
    return model
