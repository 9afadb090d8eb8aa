"""Two-shape scenes mirroring the geometry lists of newton/tests/test_narrow_phase.py (`_run_narrow_phase`):
each shape sits on its own free body, so the body frame equals the shape frame and contacts decode to world space."""
import numpy as np

import newton_amd as nt
from newton_amd import _np_math as nm

I4 = [0.0, 0.0, 0.0, 1.0]


def pair_model(geoms, device=None):
    """geoms: list of (kind, kwargs, xform7, margin, gap)."""
    b = nt.ModelBuilder()
    for kind, kwargs, xf, margin, gap in geoms:
        body = b.add_body(xform=list(xf))
        cfg = nt.ModelBuilder.ShapeConfig(margin=margin, gap=gap)
        getattr(b, "add_shape_" + kind)(body, cfg=cfg, **kwargs)
    return b.finalize(device=device)


def decode_world(model, body_q, shape0, shape1, point0, point1, normal, margin0, margin1):
    """-> list of (center, normal, penetration) like the reference test writer (penetration = distance - margins)."""
    out = []
    for i in range(len(shape0)):
        X0 = body_q[model.shape_body[shape0[i]]]
        X1 = body_q[model.shape_body[shape1[i]]]
        p0 = nm.transform_point(X0, point0[i])
        p1 = nm.transform_point(X1, point1[i])
        n = np.asarray(normal[i], dtype=np.float64)
        d = float(np.dot(p1 - p0, n) - (margin0[i] + margin1[i]))
        out.append((0.5 * (p0 + p1), n, d))
    return out


def box(h, pos, quat=I4, margin=0.0, gap=0.0):
    h = [h] * 3 if np.isscalar(h) else h
    return ("box", dict(hx=h[0], hy=h[1], hz=h[2]), [*pos, *quat], margin, gap)


def ellipsoid(r, pos, quat=I4, margin=0.0, gap=0.0):
    return ("ellipsoid", dict(rx=r[0], ry=r[1], rz=r[2]), [*pos, *quat], margin, gap)


def sphere(r, pos, margin=0.0, gap=0.0):
    return ("sphere", dict(radius=r), [*pos, *I4], margin, gap)


def capsule(r, hh, pos, quat=I4, margin=0.0, gap=0.0):
    return ("capsule", dict(radius=r, half_height=hh), [*pos, *quat], margin, gap)


def cylinder(r, hh, pos, quat=I4, margin=0.0, gap=0.0):
    return ("cylinder", dict(radius=r, half_height=hh), [*pos, *quat], margin, gap)


def cone(r, hh, pos, quat=I4, margin=0.0, gap=0.0):
    return ("cone", dict(radius=r, half_height=hh), [*pos, *quat], margin, gap)


def hull_box(h, pos, quat=I4, margin=0.0, gap=0.0, scale=(1.0, 1.0, 1.0)):
    h = [h] * 3 if np.isscalar(h) else h
    return ("convex_hull", dict(mesh=nt.Mesh.create_box(*h), scale=scale), [*pos, *quat], margin, gap)


def hull_sphere(r, pos, quat=I4, margin=0.0, gap=0.0, scale=(1.0, 1.0, 1.0)):
    return ("convex_hull", dict(mesh=nt.Mesh.create_sphere(r, 10, 12), scale=scale), [*pos, *quat], margin, gap)


def quat_z(angle):
    return [0.0, 0.0, float(np.sin(angle / 2.0)), float(np.cos(angle / 2.0))]


# (name, geoms) -- convex-path pairs used by both the oracle known-answer tests and the GPU parity tests
CONVEX_CASES = {
    "box_box_face": [box(1.0, [0, 0, 0]), box(1.0, [1.8, 0, 0])],
    "box_box_edge": [box(0.5, [0, 0, 0]), box(0.5, [1.2, 0, 0], quat_z(np.pi / 4))],
    "box_box_overlap_0p01": [box(0.5, [0, 0, 0]), box(0.5, [0, 0, 0.99])],
    "box_box_touching": [box(0.5, [0, 0, 0], gap=0.01), box(0.5, [0, 0, 1.0], gap=0.01)],
    "box_box_overlap_0p05": [box(0.5, [0, 0, 0]), box(0.5, [0, 0, 0.95])],
    "box_box_small_thickness": [box(0.5, [0, 0, 0], margin=2.5e-5), box(0.5, [0, 0, 0.99], margin=2.5e-5)],
    "box_box_large_thickness": [box(0.5, [0, 0, 0], margin=0.005), box(0.5, [0, 0, 0.99], margin=0.005)],
    "box_box_tilted": [box([0.3, 0.2, 0.1], [0, 0, 0]), box([0.2, 0.25, 0.15], [0.1, 0.05, 0.22], [0.1, 0.2, 0.05, 0.97])],
    "ell_ell_separated": [ellipsoid([1.0, 0.5, 0.3], [0, 0, 0]), ellipsoid([1.0, 0.5, 0.3], [3.0, 0, 0])],
    "ell_ell_penetrating": [ellipsoid([1.0, 0.5, 0.3], [0, 0, 0]), ellipsoid([1.0, 0.5, 0.3], [1.8, 0, 0])],
    "ell_sphere": [ellipsoid([1.0, 0.5, 0.3], [0, 0, 0]), sphere(0.5, [1.4, 0, 0])],
    "ell_box": [ellipsoid([1.0, 0.5, 0.3], [0, 0, 0]), box(0.5, [1.4, 0, 0])],
    "ell_capsule": [ellipsoid([1.0, 0.5, 0.3], [0, 0, 0]), capsule(0.5, 1.0, [0, 0.9, 0])],
    "ell_ell_rotated": [ellipsoid([1.0, 0.3, 0.3], [0, 0, 0]), ellipsoid([1.0, 0.3, 0.3], [1.3, 0, 0], quat_z(np.pi / 2))],
    "ell_ell_spherelike": [ellipsoid([1.0, 1.0, 1.0], [0, 0, 0]), ellipsoid([1.0, 1.0, 1.0], [1.8, 0, 0])],
    "capsule_box": [capsule(0.1, 0.3, [0, 0, 0.55], [0.7071068, 0, 0, 0.7071068]), box([0.5, 0.5, 0.5], [0, 0, 0])],
    "cylinder_box_flat": [cylinder(0.2, 0.1, [0.1, 0, 0.58]), box(0.5, [0, 0, 0])],
    "cylinder_box_rolling": [cylinder(0.2, 0.3, [0, 0, 0.69], [0.7071068, 0, 0, 0.7071068]), box(0.5, [0, 0, 0])],
    "cylinder_cylinder": [cylinder(0.2, 0.3, [0, 0, 0]), cylinder(0.15, 0.2, [0.05, 0, 0.48])],
    "cone_box": [cone(0.2, 0.3, [0, 0, 0.79]), box(0.5, [0, 0, 0])],
    "sphere_cone": [sphere(0.2, [0, 0, 0.42]), cone(0.3, 0.25, [0, 0, 0])],
    "hull_hull_face": [hull_box(0.5, [0, 0, 0]), hull_box(0.5, [0.2, 0.1, 0.99])],
    "hull_box_tilted": [hull_box([0.3, 0.2, 0.1], [0, 0, 0]), box([0.2, 0.25, 0.15], [0.1, 0.05, 0.22], [0.1, 0.2, 0.05, 0.97])],
    "hull_sphere_scaled": [hull_sphere(0.5, [0, 0, 0], scale=(1.0, 0.6, 0.4)), sphere(0.3, [0.1, 0.05, 0.45])],
    "hull_capsule": [hull_sphere(0.4, [0, 0, 0]), capsule(0.1, 0.3, [0.3, 0, 0.42], [0.7071068, 0, 0, 0.7071068])],
    "hull_hull_separated_gap": [hull_box(0.5, [0, 0, 0], gap=0.05), hull_box(0.5, [0, 0, 1.05], quat_z(0.3), gap=0.05)],
    "capsule_cylinder": [capsule(0.1, 0.2, [0.25, 0, 0], [0, 0.7071068, 0, 0.7071068]), cylinder(0.2, 0.3, [0, 0, 0])],
    # rolling stabilisation also applies on convex hulls (is_discrete_shape, collision_core.py:39-48): a cylinder lying on a
    # hull box (axis perpendicular to the normal) and a cone lying on its slant line
    "cylinder_hull_rolling": [cylinder(0.2, 0.3, [0.03, 0.02, 0.69], [0.7071068, 0, 0, 0.7071068]), hull_box(0.5, [0, 0, 0])],
    "cone_hull_rolling": [cone(0.2, 0.2, [0.0, 0.05, 0.5 + 0.0894427 - 0.004], [0.8506508, 0, 0, 0.5257311]), hull_box(0.5, [0, 0, 0])],
}

# cases added after the last on-hardware run of the GPU suite: the device test for them lives in the late-sorting
# tests/test_zy_recent_gpu.py (oracle and emulator tests cover them like any other case)
RECENT_CONVEX_CASES = ("cylinder_hull_rolling", "cone_hull_rolling")
