"""Shape mass properties for the model builder (host only).

Formulas restate newton/_src/geometry/inertia.py:78-300,570-605 (solid primitives, density-based) and
the bounding radius rule of newton/_src/geometry/utils.py:73-118.
"""
import numpy as np

from ._np_math import quat_to_matrix
from .enums import GeoType


def compute_inertia_shape(geo_type, scale, density, src=None):
    """(mass, com[3], inertia[3,3]) of a solid primitive about its own COM, z-up local axes."""
    if density == 0.0 or geo_type == GeoType.PLANE:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    sx, sy, sz = (float(s) for s in scale)
    if geo_type == GeoType.SPHERE:
        m = density * 4.0 / 3.0 * np.pi * sx ** 3
        Ia = 2.0 / 5.0 * m * sx * sx
        return m, np.zeros(3), np.diag([Ia, Ia, Ia])
    if geo_type == GeoType.BOX:
        m = density * 8.0 * sx * sy * sz
        return m, np.zeros(3), np.diag([
            1.0 / 3.0 * m * (sy * sy + sz * sz),
            1.0 / 3.0 * m * (sx * sx + sz * sz),
            1.0 / 3.0 * m * (sx * sx + sy * sy),
        ])
    if geo_type == GeoType.CAPSULE:
        r, h = sx, 2.0 * sy
        ms = density * (4.0 / 3.0) * np.pi * r ** 3
        mc = density * np.pi * r * r * h
        m = ms + mc
        Ia = mc * (0.25 * r * r + (1.0 / 12.0) * h * h) + ms * (0.4 * r * r + 0.375 * r * h + 0.25 * h * h)
        Ib = (mc * 0.5 + ms * 0.4) * r * r
        return m, np.zeros(3), np.diag([Ia, Ia, Ib])
    if geo_type == GeoType.CYLINDER and sz != 0.0:
        # barrel cylinder (geometry/inertia.py:166-185): solid of revolution of the arc profile, 32-point Gauss-Legendre in z
        if sz < sy:
            raise ValueError("barrel_radius must be zero or at least half_height")
        nodes, wts = np.polynomial.legendre.leggauss(32)
        z, weights = sy * nodes, sy * wts
        end_offset = np.sqrt(sz * sz - sy * sy)
        profile_offset = np.sqrt(np.maximum(sz * sz - z * z, 0.0))
        radius_profile = sx + (sy * sy - z * z) / (profile_offset + end_offset)
        r2 = radius_profile * radius_profile
        r4 = r2 * r2
        m = float(density * np.pi * np.dot(weights, r2))
        Ia = float(0.5 * density * np.pi * np.dot(weights, r4))
        Ir = float(density * np.pi * np.dot(weights, 0.25 * r4 + r2 * z * z))
        return m, np.zeros(3), np.diag([Ir, Ir, Ia])
    if geo_type == GeoType.CYLINDER:
        r, h = sx, 2.0 * sy
        m = density * np.pi * r * r * h
        Ir = 1.0 / 12.0 * m * (3.0 * r * r + h * h)
        Ia = 0.5 * m * r * r
        return m, np.zeros(3), np.diag([Ir, Ir, Ia])
    if geo_type == GeoType.ELLIPSOID:
        m = density * (4.0 / 3.0) * np.pi * sx * sy * sz
        return m, np.zeros(3), np.diag([
            0.2 * m * (sy * sy + sz * sz),
            0.2 * m * (sx * sx + sz * sz),
            0.2 * m * (sx * sx + sy * sy),
        ])
    if geo_type == GeoType.CONE:
        r, h = sx, 2.0 * sy
        m = density * np.pi * r * r * h / 3.0
        Ia = 3 / 20 * m * r * r + 3 / 80 * m * h * h
        Ib = 3 / 10 * m * r * r
        return m, np.array([0.0, 0.0, -h / 4.0]), np.diag([Ia, Ia, Ib])
    if geo_type in (GeoType.CONVEX_MESH, GeoType.MESH):
        # scaled unit-density mass properties of the source mesh (newton/_src/geometry/inertia.py:726-757: MESH and CONVEX_MESH alike)
        if src is None:
            raise ValueError("mesh and convex hull shapes need a Mesh")
        if not src.has_inertia:  # fall back to the mass properties of the scaled geometry (inertia.py:759-764)
            from .mesh import solid_mesh_mass_properties  # noqa: PLC0415

            V, com, I = solid_mesh_mass_properties(np.asarray(src.vertices, dtype=np.float64) * np.array([sx, sy, sz]),
                                                   src.indices)
            return density * V, com, density * I
        mass_ratio = abs(sx * sy * sz) * density
        I = np.asarray(src.inertia, dtype=np.float64)
        Ixx = I[0, 0] * (sy ** 2 + sz ** 2) / 2 * mass_ratio
        Iyy = I[1, 1] * (sx ** 2 + sz ** 2) / 2 * mass_ratio
        Izz = I[2, 2] * (sx ** 2 + sy ** 2) / 2 * mass_ratio
        Ixy, Ixz, Iyz = I[0, 1] * sx * sy * mass_ratio, I[0, 2] * sx * sz * mass_ratio, I[1, 2] * sy * sz * mass_ratio
        return (src.mass * mass_ratio, np.asarray(src.com, dtype=np.float64) * np.array([sx, sy, sz]),
                np.array([[Ixx, Ixy, Ixz], [Ixy, Iyy, Iyz], [Ixz, Iyz, Izz]]))
    raise NotImplementedError(f"inertia for shape type {geo_type} not supported")


def transform_inertia(mass, inertia, offset, quat):
    """R I R^T + m (|p|^2 1 - p p^T)   (newton/_src/geometry/inertia.py:570-605)."""
    R = quat_to_matrix(quat)
    offset = np.asarray(offset, dtype=np.float64)
    return R @ inertia @ R.T + mass * (np.dot(offset, offset) * np.eye(3) - np.outer(offset, offset))


def compute_shape_radius(geo_type, scale, src=None):
    if geo_type in (GeoType.CONVEX_MESH, GeoType.MESH, GeoType.HFIELD):  # bounding sphere of the scaled local AABB (geometry/utils.py:86-98)
        verts = np.asarray(src.vertices, dtype=np.float64) * np.asarray(scale, dtype=np.float64)
        return float(0.5 * np.linalg.norm(verts.max(axis=0) - verts.min(axis=0)))
    sx, sy, sz = (abs(float(s)) for s in scale)
    if geo_type == GeoType.SPHERE:
        return sx
    if geo_type == GeoType.BOX:
        return float(np.linalg.norm([sx, sy, sz]))
    if geo_type in (GeoType.CAPSULE, GeoType.CYLINDER, GeoType.CONE):
        return sx + sy
    if geo_type == GeoType.ELLIPSOID:
        return max(sx, sy, sz)
    if geo_type == GeoType.PLANE:
        return float(np.linalg.norm([sx, sy, sz])) * 0.5 if (sx > 0.0 and sy > 0.0) else 1.0e6
    raise NotImplementedError(f"radius for shape type {geo_type} not supported")
