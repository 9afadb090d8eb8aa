"""Host-side (numpy, float64) transform helpers used only by the model builder / importers.

Quaternion layout xyzw, transform = (p[3], q[4]) -- docs/concepts/conventions.rst:105-146.
"""
import numpy as np


def quat_identity():
    return np.array([0.0, 0.0, 0.0, 1.0])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + bw * ax + ay * bz - by * az,
        aw * by + bw * ay + az * bx - bz * ax,
        aw * bz + bw * az + ax * by - bx * ay,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def quat_inverse(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def quat_rotate(q, v):
    qv = np.asarray(q[:3], dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    w = q[3]
    return v * (2.0 * w * w - 1.0) + np.cross(qv, v) * w * 2.0 + qv * np.dot(qv, v) * 2.0


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    h = 0.5 * angle
    return np.array([*(axis * np.sin(h)), np.cos(h)])


def quat_rpy(roll, pitch, yaw):
    """wp.quat_rpy: rotation = Rz(yaw) * Ry(pitch) * Rx(roll)."""
    cy, sy = np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    cr, sr = np.cos(roll * 0.5), np.sin(roll * 0.5)
    cp, sp = np.cos(pitch * 0.5), np.sin(pitch * 0.5)
    return np.array([
        sr * cp * cy - cr * sp * sy,
        cr * sp * cy + sr * cp * sy,
        cr * cp * sy - sr * sp * cy,
        cr * cp * cy + sr * sp * sy,
    ])


def quat_to_matrix(q):
    return np.stack([quat_rotate(q, e) for e in np.eye(3)], axis=1)


def quat_between_vectors(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    a = a / np.linalg.norm(a)
    b = b / np.linalg.norm(b)
    d = float(np.dot(a, b))
    if d > 1.0 - 1e-12:
        return quat_identity()
    if d < -1.0 + 1e-12:
        axis = np.cross(a, [1.0, 0.0, 0.0])
        if np.linalg.norm(axis) < 1e-6:
            axis = np.cross(a, [0.0, 1.0, 0.0])
        axis /= np.linalg.norm(axis)
        return quat_from_axis_angle(axis, np.pi)
    c = np.cross(a, b)
    q = np.array([c[0], c[1], c[2], 1.0 + d])
    return q / np.linalg.norm(q)


def transform(p=(0.0, 0.0, 0.0), q=(0.0, 0.0, 0.0, 1.0)):
    return np.array([*p, *q], dtype=np.float64)


def transform_identity():
    return transform()


def transform_mul(a, b):
    return np.array([*(quat_rotate(a[3:], b[:3]) + a[:3]), *quat_mul(a[3:], b[3:])])


def transform_inverse(t):
    qi = quat_inverse(t[3:])
    return np.array([*(-quat_rotate(qi, t[:3])), *qi])


def transform_point(t, x):
    return t[:3] + quat_rotate(t[3:], x)
