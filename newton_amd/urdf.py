"""URDF-subset importer (links, primitive collision geometry, revolute/continuous/prismatic/fixed/floating joints).

Restates the rules of newton/_src/utils/import_urdf.py:62-935 that determine the builder calls for a URDF such as
newton/examples/assets/quadruped.urdf: collision shapes with default density (ignore_inertial_definitions=True)
or <inertial> overrides, FREE/FIXED base joint, joint frames from <origin>, limits from <limit>, damping/friction
from <dynamics>, and the enable_self_collisions=False filter (import_urdf.py:897-904).
Meshes, visuals, mimic joints and package:// URIs are out of scope.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

import numpy as np

from . import _np_math as nm
from .enums import ShapeFlags


def _parse_origin(element, scale=1.0):
    if element is None or element.find("origin") is None:
        return nm.transform()
    origin = element.find("origin")
    xyz = [float(x) * scale for x in (origin.get("xyz") or "0 0 0").split()]
    rpy = [float(x) for x in (origin.get("rpy") or "0 0 0").split()]
    return nm.transform(xyz, nm.quat_rpy(*rpy))


def parse_urdf(builder, source, *, xform=None, floating=None, scale=1.0, enable_self_collisions=True,
               ignore_inertial_definitions=False, collapse_fixed_joints=False):
    if collapse_fixed_joints:
        raise NotImplementedError("collapse_fixed_joints is not supported by this importer subset")
    if os.path.isfile(str(source)):
        root = ET.parse(str(source)).getroot()
    else:
        root = ET.fromstring(source)
    xform = nm.transform() if xform is None else np.asarray(xform, dtype=np.float64)
    name = root.attrib.get("name")

    def label(n):
        return f"{name}/{n}" if name else n

    default_density = builder.default_shape_cfg.density
    start_shape = builder.shape_count
    link_index = {}

    def parse_shapes(link, geoms, density):
        cfg = builder.default_shape_cfg.copy()
        cfg.density = density
        for geom_group in geoms:
            geo = geom_group.find("geometry")
            if geo is None:
                continue
            tf = _parse_origin(geom_group, scale)
            for box in geo.findall("box"):
                size = [float(x) for x in (box.get("size") or "1 1 1").split()]
                builder.add_shape_box(link, xform=tf, hx=size[0] * 0.5 * scale, hy=size[1] * 0.5 * scale,
                                      hz=size[2] * 0.5 * scale, cfg=cfg)
            for sphere in geo.findall("sphere"):
                builder.add_shape_sphere(link, xform=tf, radius=float(sphere.get("radius") or "1") * scale, cfg=cfg)
            for cyl in geo.findall("cylinder"):
                builder.add_shape_cylinder(link, xform=tf, radius=float(cyl.get("radius") or "1") * scale,
                                           half_height=float(cyl.get("length") or "1") * 0.5 * scale, cfg=cfg)
            for cap in geo.findall("capsule"):
                builder.add_shape_capsule(link, xform=tf, radius=float(cap.get("radius") or "1") * scale,
                                          half_height=float(cap.get("height") or "1") * 0.5 * scale, cfg=cfg)
            if geo.find("mesh") is not None:
                raise NotImplementedError("URDF mesh geometry is not supported by this importer subset")

    joints = []
    for joint in root.findall("joint"):
        jd = {
            "name": joint.get("name"),
            "parent": joint.find("parent").get("link"),
            "child": joint.find("child").get("link"),
            "type": joint.get("type"),
            "origin": _parse_origin(joint, scale),
            "damping": builder.default_joint_cfg.target_kd,
            "friction": builder.default_joint_cfg.friction,
            "axis": np.array([1.0, 0.0, 0.0]),
            "limit_lower": builder.default_joint_cfg.limit_lower,
            "limit_upper": builder.default_joint_cfg.limit_upper,
            "effort": builder.default_joint_cfg.effort_limit,
        }
        el_axis = joint.find("axis")
        if el_axis is not None:
            jd["axis"] = np.array([float(x) for x in (el_axis.get("xyz") or "1 0 0").split()])
        el_dyn = joint.find("dynamics")
        if el_dyn is not None:
            jd["damping"] = float(el_dyn.get("damping", jd["damping"]))
            jd["friction"] = float(el_dyn.get("friction", jd["friction"]))
        el_limit = joint.find("limit")
        if el_limit is not None:
            jd["limit_lower"] = float(el_limit.get("lower", jd["limit_lower"]))
            jd["limit_upper"] = float(el_limit.get("upper", jd["limit_upper"]))
            jd["effort"] = float(el_limit.get("effort", jd["effort"]))
        joints.append(jd)

    # DFS topological order, ties broken by file order (newton/_src/utils/topology.py:18-92); bodies follow the
    # joint order: [root, child(j0), child(j1), ...]  (import_urdf.py:606-623)
    if joints:
        outgoing = {}
        has_parent = set()
        for jid, jd in enumerate(joints):
            outgoing.setdefault(jd["parent"], []).append((jid, jd["child"]))
            has_parent.add(jd["child"])
        roots = sorted({jd["parent"] for jd in joints} - has_parent)
        order = []

        def visit(node):
            for jid, child in sorted(outgoing.get(node, [])):
                order.append(jid)
                visit(child)

        for r in roots:
            visit(r)
        joints = [joints[i] for i in order]
        body_order = [joints[0]["parent"]] + [jd["child"] for jd in joints]
        urdf_links = [root.find(f"link[@name='{b}']") for b in body_order]
        if any(l is None for l in urdf_links):
            raise ValueError("URDF joint references a missing link")
    else:
        urdf_links = root.findall("link")

    for urdf_link in urdf_links:
        lname = urdf_link.get("name")
        link = builder.add_link(label=label(lname))
        link_index[lname] = link
        parse_shapes(link, urdf_link.findall("collision"), default_density)
        el_inertia = urdf_link.find("inertial")
        if not ignore_inertial_definitions and el_inertia is not None:
            frame = _parse_origin(el_inertia, scale)
            builder.body_com[link] = frame[:3].copy()
            el_i = el_inertia.find("inertia")
            if el_i is not None:
                I = np.zeros((3, 3))
                I[0, 0] = float(el_i.get("ixx", 0)) * scale ** 2
                I[1, 1] = float(el_i.get("iyy", 0)) * scale ** 2
                I[2, 2] = float(el_i.get("izz", 0)) * scale ** 2
                I[0, 1] = I[1, 0] = float(el_i.get("ixy", 0)) * scale ** 2
                I[0, 2] = I[2, 0] = float(el_i.get("ixz", 0)) * scale ** 2
                I[1, 2] = I[2, 1] = float(el_i.get("iyz", 0)) * scale ** 2
                R = nm.quat_to_matrix(frame[3:])
                I = R @ I @ R.T
                builder.body_inertia[link] = I
                builder.body_inv_inertia[link] = np.linalg.inv(I) if I.any() else I.copy()
            el_mass = el_inertia.find("mass")
            if el_mass is not None:
                mval = float(el_mass.get("value", 0))
                builder.body_mass[link] = mval
                builder.body_inv_mass[link] = 1.0 / mval if mval > 0.0 else 0.0
    end_shape = builder.shape_count

    base_link = joints[0]["parent"] if joints else next(iter(link_index))
    root_body = link_index[base_link]
    joint_indices = []
    if floating:
        j = builder.add_joint_free(root_body, label=label("floating_base"))
        joint_indices.append(j)
        start = builder.joint_q_start[j]
        builder.joint_q[start:start + 7] = list(xform)
    else:
        joint_indices.append(builder.add_joint_fixed(-1, root_body, parent_xform=xform, label=label("fixed_base")))

    for jd in joints:
        parent, child = link_index[jd["parent"]], link_index[jd["child"]]
        common = dict(parent_xform=jd["origin"], label=label(jd["name"]))
        if jd["type"] in ("revolute", "continuous"):
            j = builder.add_joint_revolute(parent, child, axis=jd["axis"], target_kd=jd["damping"], friction=jd["friction"],
                                           limit_lower=jd["limit_lower"], limit_upper=jd["limit_upper"],
                                           effort_limit=jd["effort"], **common)
        elif jd["type"] == "prismatic":
            j = builder.add_joint_prismatic(parent, child, axis=jd["axis"], target_kd=jd["damping"], friction=jd["friction"],
                                            limit_lower=jd["limit_lower"] * scale, limit_upper=jd["limit_upper"] * scale,
                                            effort_limit=jd["effort"], **common)
        elif jd["type"] == "fixed":
            j = builder.add_joint_fixed(parent, child, **common)
        elif jd["type"] == "floating":
            j = builder.add_joint_free(child, parent=parent, **common)
        else:
            raise NotImplementedError(f"Unsupported URDF joint type: {jd['type']}")
        joint_indices.append(j)

    builder.add_articulation(joint_indices, label=name)

    if not enable_self_collisions:
        colliding = [i for i in range(start_shape, end_shape) if builder.shape_flags[i] & ShapeFlags.COLLIDE_SHAPES]
        for a, i in enumerate(colliding):
            for j in colliding[a + 1:]:
                builder.add_shape_collision_filter_pair(i, j)
    return joint_indices
