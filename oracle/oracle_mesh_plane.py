"""TEST INFRASTRUCTURE ONLY (never imported by newton_amd/ or by bench.py's timed path).

CPU restatement, in float32 numpy scalars, of the reference's mesh-vs-infinite-plane leg under reduce_contacts=True:

  * contact generation   newton/_src/geometry/narrow_phase.py:1866-1990 (narrow_phase_process_mesh_plane_contacts_reduce_kernel;
                         the unreduced twin :1760-1860 computes the same contact per vertex): vertex -> world, projection on the
                         plane through the plane's frame, distance = (v - proj) . n, admitted when distance < gap sum + margin sum,
                         centre = midpoint, normal = -n (mesh -> plane), fingerprint = vertex index
  * buffering            contact_reduction_global.py:2059-2096 (write_contact_to_reducer -> export_contact_to_buffer: position,
                         depth, octahedral normal code)
  * reduction + export   oracle_reduce.reduce_buffered_contacts (reduce_contact_in_hashtable :1246-1346 + the export kernel)

Warp builtins in the operand order of oracle/wp_builtins.h (quat_rotate, transform_point / _vector / _inverse, dot left to right).
Pinned by tests/golden/mesh_plane_reference_vectors.npz, the record of the reference's own kernels executed on five scenes
(tests/golden/make_mesh_plane_reference_vectors.py)."""
import numpy as np

import oracle_reduce as orr

f32 = np.float32


def _v(a):
    return [f32(a[0]), f32(a[1]), f32(a[2])]


def _dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross(a, b):
    return [f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])), f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))]


def quat_rotate(q, v):
    """v * (2 w^2 - 1) + cross(qv, v) * w * 2 + qv * dot(qv, v) * 2"""
    qv, w = _v(q[:3]), f32(q[3])
    k = f32(f32(f32(f32(2.0) * w) * w) - f32(1.0))
    c, d = _cross(qv, v), _dot(qv, v)
    return [f32(f32(f32(v[i] * k) + f32(f32(c[i] * w) * f32(2.0))) + f32(f32(qv[i] * d) * f32(2.0))) for i in range(3)]


def transform_point(t, x):
    r = quat_rotate(t[3:], x)
    return [f32(f32(t[i]) + r[i]) for i in range(3)]


def transform_inverse(t):
    qi = [f32(-t[3]), f32(-t[4]), f32(-t[5]), f32(t[6])]
    r = quat_rotate(qi, _v(t[:3]))
    return [f32(-r[0]), f32(-r[1]), f32(-r[2]), *qi]


def mesh_plane_contacts(s, mesh_shape, plane_shape):
    """Every vertex of the mesh within margin + gap of the plane, in vertex order:
    -> list of (vertex index, centre[3], normal[3] mesh -> plane, distance)."""
    X_mesh, X_plane = s["shape_transform"][mesh_shape], s["shape_transform"][plane_shape]
    X_plane_sw = transform_inverse(X_plane)
    plane_normal = quat_rotate(X_plane[3:], [f32(0.0), f32(0.0), f32(1.0)])
    scale = _v(s["shape_data"][mesh_shape][:3])
    total_margin = f32(f32(s["shape_data"][mesh_shape][3]) + f32(s["shape_data"][plane_shape][3]))
    gap_sum = f32(f32(s["shape_gap"][mesh_shape]) + f32(s["shape_gap"][plane_shape]))
    threshold = f32(gap_sum + total_margin)
    v0, n = int(s["vertex_start"][mesh_shape]), int(s["vertex_count"][mesh_shape])
    out = []
    for vi in range(n):
        p = s["vertices"][v0 + vi]
        local = [f32(f32(p[k]) * scale[k]) for k in range(3)]
        world = transform_point(X_mesh, local)
        in_plane = transform_point(X_plane_sw, world)
        on_plane = transform_point(X_plane, [in_plane[0], in_plane[1], f32(0.0)])
        diff = [f32(world[k] - on_plane[k]) for k in range(3)]
        distance = _dot(diff, plane_normal)
        if distance < threshold:
            centre = [f32(f32(world[k] + on_plane[k]) * f32(0.5)) for k in range(3)]
            out.append((vi, centre, [f32(-plane_normal[k]) for k in range(3)], distance))
    return out


def buffered_contacts(s):
    """The unreduced list of a scene (mesh_plane_cases.scene), pair after pair, the way the reducer's buffer holds it."""
    rows = dict(pair=[], fp=[], pos=[], depth=[], normal=[], xform_a=[], aabb_lo=[], aabb_hi=[], res=[])
    for mesh_shape, plane_shape in s["pairs"]:
        for vi, centre, normal, distance in mesh_plane_contacts(s, int(mesh_shape), int(plane_shape)):
            rows["pair"].append((int(mesh_shape), int(plane_shape)))
            rows["fp"].append(vi)
            rows["pos"].append(centre)
            rows["depth"].append(distance)
            rows["normal"].append(normal)
            rows["xform_a"].append(s["shape_transform"][mesh_shape])
            rows["aabb_lo"].append(s["aabb_lo"][mesh_shape])
            rows["aabb_hi"].append(s["aabb_hi"][mesh_shape])
            rows["res"].append(s["res"][mesh_shape])
    f = lambda k, w: np.asarray(rows[k], np.float32).reshape(-1, w) if w > 1 else np.asarray(rows[k], np.float32)  # noqa: E731
    return dict(pair=np.asarray(rows["pair"], np.int32).reshape(-1, 2), fp=np.asarray(rows["fp"], np.int32), pos=f("pos", 3),
                depth=f("depth", 1), normal=f("normal", 3), xform_a=f("xform_a", 7), aabb_lo=f("aabb_lo", 3), aabb_hi=f("aabb_hi", 3),
                res=np.asarray(rows["res"], np.int32).reshape(-1, 3))


def mesh_plane_rows(s):
    """Scene -> the reduced contacts sorted by (mesh shape, plane shape, vertex): dict(pair, fp, pos, normal, depth, margin_a,
    margin_b) -- what export_reduced_contacts_kernel hands to the contact writer."""
    c = buffered_contacts(s)
    out = orr.reduce_buffered_contacts(c)
    out["margin_a"] = np.asarray([s["shape_data"][a][3] for a, _ in out["pair"]], np.float32)
    out["margin_b"] = np.asarray([s["shape_data"][b][3] for _, b in out["pair"]], np.float32)
    return out
