"""Scenes for the collision reference vectors (shared by tests/golden/make_collide_reference_vectors.py and the tests)."""
import numpy as np


def _settled(model, steps, dt=2e-3, iterations=2):
    """A few checker XPBD steps so that shapes rest on / penetrate each other in generic (tilted) poses."""
    import oracle_bridge as ob

    orc = ob.Oracle(model)
    a, b = ob.OracleState(model), ob.OracleState(model)
    for _ in range(steps):
        ct = orc.contacts()
        orc.collide(a.body_q, ct)
        orc.xpbd_step(a, b, orc.control(), ct if ct.count[0] else None, dt, iterations=iterations)
        a, b = b, a
    return np.array(a.body_q, np.float32)


def cases():
    from scenes import box_stack_scene, mixed_primitive_scene, quadruped_convex_scene, quadruped_scene

    def mixed(seed, steps):
        def make():
            m = mixed_primitive_scene(1, seed=seed)
            return m, _settled(m, steps)
        return make

    def boxes(seed, steps):
        def make():
            m = box_stack_scene(1, n_boxes=4, seed=seed, jitter=0.02)
            return m, _settled(m, steps, dt=1.0 / 240.0, iterations=4)
        return make

    def quad(convex, steps):
        def make():
            m = (quadruped_convex_scene if convex else quadruped_scene)(1, seed=5)
            jq = np.array(m.joint_q, copy=True)
            jq[2] -= 0.2
            import newton_amd as nt

            m.joint_q = jq
            m.body_q, m.body_qd = nt.articulation.eval_fk_numpy(m, m.joint_q, m.joint_qd)
            return m, _settled(m, steps, dt=1e-3)
        return make

    def pair_matrix(seed, with_hull=False):
        """Every unordered pair of collider types overlapping once (plus each type on the ground plane): all analytic routines
        and every MPR / GJK type pair of the convex path in one scene."""
        def make():
            import newton_amd as nt

            rng = np.random.default_rng(seed)
            kinds = ["sphere", "capsule", "ellipsoid", "cylinder", "box", "cone"] + (["hull"] if with_hull else [])
            env = nt.ModelBuilder()
            hull = None
            if with_hull:
                pts = rng.normal(size=(20, 3))
                pts = 0.09 * pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.7, 1.0, size=(20, 1))
                hull = nt.Mesh.create_convex_hull(pts.astype(np.float32)) if hasattr(nt.Mesh, "create_convex_hull") else None

            def add(kind, pos):
                q = nt._np_math.quat_rpy(*rng.uniform(-1.2, 1.2, size=3))
                b = env.add_body(xform=[*pos, *q])
                if kind == "sphere":
                    env.add_shape_sphere(b, radius=0.09)
                elif kind == "capsule":
                    env.add_shape_capsule(b, radius=0.05, half_height=0.1)
                elif kind == "ellipsoid":
                    env.add_shape_ellipsoid(b, rx=0.11, ry=0.08, rz=0.06)
                elif kind == "cylinder":
                    env.add_shape_cylinder(b, radius=0.07, half_height=0.09)
                elif kind == "box":
                    env.add_shape_box(b, hx=0.09, hy=0.07, hz=0.06)
                elif kind == "cone":
                    env.add_shape_cone(b, radius=0.08, half_height=0.1)
                else:
                    env.add_shape_convex_hull(b, mesh=hull)

            cell = 0
            for i, ka in enumerate(kinds):
                for kb in kinds[i:]:
                    x, y = 0.6 * (cell % 6), 0.6 * (cell // 6)
                    d = rng.uniform(-1.0, 1.0, size=3)
                    d = 0.09 * d / np.linalg.norm(d)  # centres 9 cm apart: overlapping for every pair of these sizes
                    add(ka, [x, y, 1.0])
                    add(kb, [x + d[0], y + d[1], 1.0 + d[2]])
                    cell += 1
            for k, kind in enumerate(kinds):  # and each type touching the ground
                add(kind, [0.6 * k, -0.8, 0.05])
            scene = nt.ModelBuilder()
            scene.replicate(env, 1)
            scene.add_ground_plane()
            m = scene.finalize()
            return m, np.array(m.body_q, np.float32)
        return make

    def hulls(n, steps):
        def make():
            from scenes import hull_bin_scene

            m = hull_bin_scene(1, n_hulls=n, seed=3)
            return m, _settled(m, steps, dt=1.0 / 600.0)
        return make

    return {"hull_bin_a": hulls(6, 0), "hull_bin_b": hulls(8, 120), "pair_matrix_a": pair_matrix(11), "pair_matrix_b": pair_matrix(12), "mixed_primitives_a": mixed(3, 0), "mixed_primitives_b": mixed(4, 60), "mixed_primitives_c": mixed(5, 150),
            "box_stack_a": boxes(1, 0), "box_stack_b": boxes(2, 40), "quadruped_cylinders": quad(False, 30),
            "quadruped_box_feet": quad(True, 30)}


def barrel_cases():
    """Barrel cylinders (scale z = radius of the side arc; builder.py:7050-7089, support_function.py:284-305): standing on an end cap
    (the analytic plane route, narrow_phase.py:682-686), tilted and lying on the ground (plane proxy through MPR / GJK), and overlapping
    every other collider type incl. spheres (always MPR / GJK, narrow_phase.py:847) and another barrel.  Kept apart from cases(): the
    record is collide_barrel_reference_vectors.npz (make_collide_reference_vectors.py --barrel)."""
    def barrel(seed, arc):
        def make():
            import newton_amd as nt

            rng = np.random.default_rng(seed)
            env = nt.ModelBuilder()
            R, HH = 0.05, 0.08
            BR = HH if arc == "tight" else 0.13  # barrel radius == half height: the arc's centre sits on the axis end ring

            def body(pos, q=None):
                q = nt._np_math.quat_rpy(*rng.uniform(-1.2, 1.2, size=3)) if q is None else q
                return env.add_body(xform=[*pos, *q])

            def other(kind, b):
                if kind == "sphere":
                    env.add_shape_sphere(b, radius=0.07)
                elif kind == "capsule":
                    env.add_shape_capsule(b, radius=0.04, half_height=0.08)
                elif kind == "ellipsoid":
                    env.add_shape_ellipsoid(b, rx=0.09, ry=0.07, rz=0.05)
                elif kind == "cylinder":
                    env.add_shape_cylinder(b, radius=0.06, half_height=0.07)
                elif kind == "barrel":
                    env.add_shape_cylinder(b, radius=0.04, half_height=0.06, barrel_radius=0.09)
                elif kind == "box":
                    env.add_shape_box(b, hx=0.07, hy=0.06, hz=0.05)
                else:
                    env.add_shape_cone(b, radius=0.07, half_height=0.08)

            for cell, kind in enumerate(["sphere", "capsule", "ellipsoid", "cylinder", "barrel", "box", "cone"]):
                x = 0.6 * cell
                d = rng.uniform(-1.0, 1.0, size=3)
                d = 0.08 * d / np.linalg.norm(d)
                env.add_shape_cylinder(body([x, 0.0, 1.0]), radius=R, half_height=HH, barrel_radius=BR)
                other(kind, body([x + d[0], d[1], 1.0 + d[2]]))
            ident = [0.0, 0.0, 0.0, 1.0]
            # on the ground: upright on the end cap, slightly tilted (still the cap route), tilted past it (MPR / GJK), on its side, random
            poses = [ident, nt._np_math.quat_rpy(0.05, -0.03, 0.4), nt._np_math.quat_rpy(0.9, 0.2, 0.0), nt._np_math.quat_rpy(np.pi / 2, 0.0, 0.3), None]
            heights = [HH - 0.004, HH - 0.002, 0.07, R + 0.015, 0.08]
            for k, (q, z) in enumerate(zip(poses, heights)):
                env.add_shape_cylinder(body([0.6 * k, -0.8, z], q), radius=R, half_height=HH, barrel_radius=BR)
            scene = nt.ModelBuilder()
            scene.replicate(env, 1)
            scene.add_ground_plane()
            m = scene.finalize()
            return m, np.array(m.body_q, np.float32)
        return make

    return {"barrel_wide": barrel(21, "wide"), "barrel_tight": barrel(22, "tight")}
