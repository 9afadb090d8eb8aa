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
