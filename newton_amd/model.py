"""Model / State / Control: the reference's flat-array data model (the drop-in boundary).

Attribute names, shapes and dtypes follow newton/_src/sim/model.py:808-1364, state.py:113-171 and
control.py:31-68.  On the host every array is a numpy AoS array exactly as Newton lays it out
(``body_q[B,7]``, ``body_qd[B,6]`` ...).  On an MI355X the *resident* representation is env-major SoA
(``[component][slot][env]``, include/newton_hip.h) and the AoS arrays are materialised on demand.
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from . import _lib
from .enums import BodyFlags, GeoType, JointType, ShapeFlags


def _is_gpu(device) -> bool:
    return device is not None and str(device).startswith(("cuda", "hip"))


def _torch():
    import torch  # noqa: PLC0415

    return torch


def pair_types_analytic(type_a: int, type_b: int) -> bool:
    """True when the reference's primitive narrow-phase kernel owns this type pair (narrow_phase.py:458-1014); every other
    supported pair goes through MPR / GJK + manifold in the second launch."""
    ta, tb = sorted((int(type_a), int(type_b)))
    if ta == GeoType.PLANE:
        return tb in (GeoType.SPHERE, GeoType.CAPSULE, GeoType.ELLIPSOID, GeoType.CYLINDER, GeoType.BOX)
    if ta == GeoType.SPHERE:
        return tb in (GeoType.SPHERE, GeoType.CAPSULE, GeoType.CYLINDER, GeoType.BOX)
    return ta == GeoType.CAPSULE and tb == GeoType.CAPSULE


class EnvTemplate:
    """Env-uniform topology extracted from a replicated model (what ModelBuilder.replicate() produces)."""

    def __init__(self, model: Model):
        m = model
        E = m.world_count if m.world_count > 0 else 1
        self.env_count = E
        self.env_stride = ((E + 63) // 64) * 64

        def per_env(world_arr, what):
            world_arr = np.asarray(world_arr)
            if m.world_count == 0:
                return np.arange(len(world_arr)), np.zeros(0, dtype=np.int64)
            local = np.flatnonzero(world_arr >= 0)
            glob = np.flatnonzero(world_arr < 0)
            if len(local) % E != 0:
                raise NotImplementedError(f"heterogeneous worlds: {what} count is not a multiple of world_count")
            n = len(local) // E
            expect = np.repeat(np.arange(E), n)
            if not np.array_equal(world_arr[local], expect) or (len(local) and not np.array_equal(local, np.arange(len(local)))):
                raise NotImplementedError(
                    f"{what}s must be ordered world-major with global (world -1) entries at the tail")
            return local, glob

        body_local, body_glob = per_env(m.body_world, "body")
        if len(body_glob):
            raise NotImplementedError("bodies in the global world (-1) are not supported together with worlds")
        joint_local, joint_glob = per_env(m.joint_world, "joint")
        if len(joint_glob):
            raise NotImplementedError("joints in the global world (-1) are not supported together with worlds")
        if m.world_count == 0:
            shape_world = np.where(np.asarray(m.shape_body) >= 0, 0, -1)  # body-attached shapes are env-local
        else:
            shape_world = np.asarray(m.shape_world)
        shape_local = np.flatnonzero(shape_world >= 0)
        shape_glob = np.flatnonzero(shape_world < 0)
        # local shapes must form one contiguous world-major block; global (static) shapes may sit before and / or after
        # it (the reference's examples call add_ground_plane() first as often as last)
        L0 = int(shape_local[0]) if len(shape_local) else 0
        if len(shape_local) % E or (len(shape_local) and not np.array_equal(shape_local, np.arange(L0, L0 + len(shape_local)))):
            raise NotImplementedError("env-local shapes must form one contiguous, world-major block")
        if m.world_count > 0 and len(shape_local):
            expect = np.repeat(np.arange(E), len(shape_local) // E)
            if not np.array_equal(shape_world[shape_local], expect):
                raise NotImplementedError("shapes must be ordered world-major")
        if np.any(np.asarray(m.shape_body)[shape_glob] >= 0):
            raise NotImplementedError("global shapes must be static (shape_body == -1)")
        self.shape_local0 = L0
        self.gshape_id = shape_glob.astype(np.int32)

        self.nb = len(body_local) // E
        self.nj = len(joint_local) // E
        self.ns = len(shape_local) // E
        self.ng = len(shape_glob)
        self.nd = m.joint_dof_count // E
        self.nc = m.joint_coord_count // E
        self.ntq = len(m.joint_target_q) // E
        nb, nj, ns = self.nb, self.nj, self.ns

        def uniform(a, n, what, offset=None):
            a = np.asarray(a).reshape(E, n, *np.asarray(a).shape[1:]) if n else np.zeros((E, 0), dtype=np.int32)
            if offset is not None:
                a = a - offset
            if E > 1 and not np.all(a == a[0:1]):
                raise NotImplementedError(f"heterogeneous worlds: {what} differs between worlds")
            return np.ascontiguousarray(a[0]).astype(np.int32)

        env_ids = np.arange(E).reshape(E, 1)
        self.body_flags = uniform(m.body_flags, nb, "body_flags")
        self.joint_type = uniform(m.joint_type, nj, "joint_type")
        self.joint_enabled = uniform(np.asarray(m.joint_enabled, dtype=np.int32), nj, "joint_enabled")
        jp = np.asarray(m.joint_parent).reshape(E, nj) if nj else np.zeros((E, 0), dtype=np.int32)
        jp = np.where(jp >= 0, jp - env_ids * nb, -1)
        if E > 1 and not np.all(jp == jp[0:1]):
            raise NotImplementedError("heterogeneous worlds: joint_parent differs between worlds")
        self.joint_parent = jp[0].astype(np.int32)
        self.joint_child = uniform(m.joint_child, nj, "joint_child", env_ids * nb)
        self.joint_q_start = uniform(m.joint_q_start, nj, "joint_q_start", env_ids * self.nc)
        self.joint_qd_start = uniform(m.joint_qd_start, nj, "joint_qd_start", env_ids * self.nd)
        self.joint_tq_start = uniform(m.joint_target_q_start, nj, "joint_target_q_start", env_ids * self.ntq)
        dd = np.asarray(m.joint_dof_dim).reshape(E, nj, 2) if nj else np.zeros((E, 0, 2), dtype=np.int32)
        if E > 1 and not np.all(dd == dd[0:1]):
            raise NotImplementedError("heterogeneous worlds: joint_dof_dim differs between worlds")
        self.joint_lin_count = dd[0, :, 0].astype(np.int32)
        self.joint_ang_count = dd[0, :, 1].astype(np.int32)

        # articulations (SolverFeatherstone): env-local joint ranges, same in every world
        A = int(getattr(m, "articulation_count", 0))
        if A and nj and A % E == 0:
            na = A // E
            starts = np.asarray(m.articulation_start).reshape(E, na) - env_ids * nj
            ends = np.asarray(m.articulation_end).reshape(E, na) - env_ids * nj
            if E > 1 and not (np.all(starts == starts[0:1]) and np.all(ends == ends[0:1])):
                raise NotImplementedError("heterogeneous worlds: articulations differ between worlds")
            contiguous = na > 0 and starts[0, 0] == 0 and ends[0, -1] == nj and np.all(starts[0, 1:] == ends[0, :-1])
            self.na = na if contiguous else 0
        else:
            self.na = 0
        if self.na:
            self.art_start = np.concatenate([starts[0], [nj]]).astype(np.int32)
            dof_edges = np.concatenate([self.joint_qd_start, [self.nd]])
            self.max_art_dofs = int(max(dof_edges[self.art_start[k + 1]] - dof_edges[self.art_start[k]]
                                        for k in range(self.na)))
        else:
            self.art_start = np.zeros(1, dtype=np.int32)
            self.max_art_dofs = 0

        sb = np.asarray(m.shape_body)[L0:L0 + E * ns].reshape(E, ns) if ns else np.zeros((E, 0), dtype=np.int32)
        sb = np.where(sb >= 0, sb - env_ids * nb, -1)
        if E > 1 and not np.all(sb == sb[0:1]):
            raise NotImplementedError("heterogeneous worlds: shape_body differs between worlds")
        glob_minus1 = -np.ones(self.ng, dtype=np.int32)
        self.shape_body = np.concatenate([sb[0].astype(np.int32), glob_minus1])

        def shape_uniform(a, what):
            a = np.asarray(a)
            loc = a[L0:L0 + E * ns].reshape(E, ns) if ns else np.zeros((E, 0), dtype=a.dtype)
            if E > 1 and not np.all(loc == loc[0:1]):
                raise NotImplementedError(f"heterogeneous worlds: {what} differs between worlds")
            return np.concatenate([loc[0], a[shape_glob]]).astype(np.int32)

        self.shape_type = shape_uniform(m.shape_type, "shape_type")
        # The tile kernels see a triangle mesh as what compute_shape_aabbs makes of it -- a shape with a pre-computed local AABB
        # (collide.py:421-445, the branch MESH and CONVEX_MESH share): their table carries CONVEX_MESH for it (AABB from the vertex
        # bounds below).  A MESH never is a tile PAIR: its pairs go to the SDF / vertex legs or are refused further down.
        self.tile_shape_type = np.where((self.shape_type == int(GeoType.MESH)) | (self.shape_type == int(GeoType.HFIELD)), int(GeoType.CONVEX_MESH),
                                        self.shape_type).astype(np.int32)  # (a heightfield likewise: its bounding box, collide.py:348)
        self.shape_flags = shape_uniform(m.shape_flags, "shape_flags")
        self.shape_group = shape_uniform(m.shape_collision_group, "shape_collision_group")
        # convex-hull vertex slices (shared Mesh assets => identical in every world) + unscaled hull bounds per shape
        n_all = len(np.asarray(m.shape_type))
        mesh_start = np.asarray(getattr(m, "shape_mesh_start", -np.ones(n_all)), dtype=np.int32)
        mesh_count = np.asarray(getattr(m, "shape_mesh_count", np.zeros(n_all)), dtype=np.int32)
        self.shape_mesh_start = shape_uniform(mesh_start, "convex hull mesh")
        self.shape_mesh_count = shape_uniform(mesh_count, "convex hull mesh")
        self.mesh_points = np.asarray(getattr(m, "mesh_points", np.zeros((0, 3))), dtype=np.float32).reshape(-1, 3)
        self.shape_mesh_bounds = np.zeros((ns + self.ng, 6), dtype=np.float32)
        for k in range(ns + self.ng):
            if self.shape_mesh_count[k] > 0:
                v = self.mesh_points[self.shape_mesh_start[k]:self.shape_mesh_start[k] + self.shape_mesh_count[k]]
                self.shape_mesh_bounds[k, :3], self.shape_mesh_bounds[k, 3:] = v.min(axis=0), v.max(axis=0)

        # candidate pairs, per env, in Newton's order
        pairs = np.asarray(m.shape_contact_pairs, dtype=np.int64).reshape(-1, 2)
        glob_rank = -np.ones(len(np.asarray(m.shape_type)) + 1, dtype=np.int64)
        glob_rank[shape_glob] = np.arange(len(shape_glob))

        def is_local(col):
            return (col >= L0) & (col < L0 + E * ns)

        if len(pairs):
            wa = np.where(is_local(pairs[:, 0]), (pairs[:, 0] - L0) // max(ns, 1), -1)
            wb = np.where(is_local(pairs[:, 1]), (pairs[:, 1] - L0) // max(ns, 1), -1)
            pw = np.maximum(wa, wb)
            keep = pw >= 0  # global-vs-global pairs are static-static: no consumer, dropped
            pairs, pw = pairs[keep], pw[keep]
            if np.any((wa[keep] >= 0) & (wb[keep] >= 0) & (wa[keep] != wb[keep])):
                raise ValueError("shape_contact_pairs contains a cross-world pair")
            if len(pairs) % E:
                raise NotImplementedError("heterogeneous worlds: candidate pair count differs between worlds")
            npair = len(pairs) // E
            if not np.array_equal(pw, np.repeat(np.arange(E), npair)):
                raise NotImplementedError("shape_contact_pairs must be ordered world-major")

            def to_local(col):
                w = pw
                return np.where(is_local(col), col - L0 - w * ns, ns + glob_rank[col])

            la = to_local(pairs[:, 0]).reshape(E, npair)
            lb = to_local(pairs[:, 1]).reshape(E, npair)
            if E > 1 and not (np.all(la == la[0:1]) and np.all(lb == lb[0:1])):
                raise NotImplementedError("heterogeneous worlds: candidate pairs differ between worlds")
            self.pair_a, self.pair_b = la[0].astype(np.int32), lb[0].astype(np.int32)
        else:
            self.pair_a = np.zeros(0, dtype=np.int32)
            self.pair_b = np.zeros(0, dtype=np.int32)
        # Pairs whose two shapes carry a texture SDF and collision edges (not box-box) leave the primitive / GJK-MPR path: the
        # reference routes them to its mesh-mesh SDF kernel (narrow_phase.py:620-640).  They are not tile pairs; the pipeline's
        # SDF leg walks them per world (newton_amd/sdf_pipeline.py), in ascending Newton (shape0, shape1) order.
        n_all = len(np.asarray(m.shape_type))
        sdf_idx = np.asarray(getattr(m, "_shape_sdf_index", None) if getattr(m, "_shape_sdf_index", None) is not None
                             else -np.ones(n_all), dtype=np.int32)
        e_rng = np.asarray(getattr(m, "shape_edge_range", None) if getattr(m, "shape_edge_range", None) is not None
                           else np.zeros((n_all, 2)), dtype=np.int32).reshape(n_all, 2)
        self.shape_sdf_index = shape_uniform(sdf_idx, "shape SDF index")
        self.shape_edge_count = shape_uniform(e_rng[:, 1], "shape collision-edge count")
        has_sdf = (self.shape_sdf_index >= 0) & (self.shape_edge_count > 0)
        # both shapes hydroelastic (ShapeFlags.HYDROELASTIC) with SDFs: the SDF-SDF leg, tested first (narrow_phase.py:531-538)
        hydro = ((self.shape_flags & int(ShapeFlags.HYDROELASTIC)) != 0) & (self.shape_sdf_index >= 0)
        is_hydro_pair = np.array([hydro[a] and hydro[b] for a, b in zip(self.pair_a, self.pair_b)], dtype=bool)
        is_sdf_pair = is_hydro_pair | np.array(
            [has_sdf[a] and has_sdf[b] and not (self.shape_type[a] == GeoType.BOX and self.shape_type[b] == GeoType.BOX)
             for a, b in zip(self.pair_a, self.pair_b)], dtype=bool)

        def newton_id0(l):  # Newton id of template shape l in world 0 (the order is the same in every world)
            return L0 + l if l < ns else int(shape_glob[l - ns])

        # a triangle mesh against an INFINITE plane (scale x = y = 0) is tested vertex by vertex (narrow_phase.py:618-631, after the
        # mesh-mesh / SDF-edge rule above): the vertex leg of the pipeline (csrc/nt_mesh_plane.hip), not a tile pair either
        scale_all = np.asarray(m.shape_scale, dtype=np.float32).reshape(-1, 3)

        def infinite_plane(l):
            return int(self.shape_type[l]) == GeoType.PLANE and scale_all[newton_id0(l), 0] == 0.0 and scale_all[newton_id0(l), 1] == 0.0

        def tri_mesh(l):
            return int(self.shape_type[l]) == GeoType.MESH

        def mesh_like(l):  # a triangle mesh or a heightfield (narrow_phase.py:553-583: the same triangle kernels, cell by cell)
            return int(self.shape_type[l]) in (GeoType.MESH, GeoType.HFIELD)

        is_mesh_plane_pair = ~is_sdf_pair & np.array([(infinite_plane(a) and tri_mesh(b)) or (infinite_plane(b) and tri_mesh(a))
                                                      for a, b in zip(self.pair_a, self.pair_b)], dtype=bool)
        is_sdf_pair = is_sdf_pair | is_mesh_plane_pair
        # a triangle mesh against a convex primitive (narrow_phase.py:633-638 `shape_pairs_mesh`, after the rules above): the triangle
        # leg of the pipeline (csrc/nt_mesh_triangle.hip, pair kind 3) -- midphase over the mesh's triangles, GJK / MPR per triangle
        tri_partner_types = (GeoType.SPHERE, GeoType.CAPSULE, GeoType.ELLIPSOID, GeoType.CYLINDER, GeoType.BOX, GeoType.CONE,
                             GeoType.CONVEX_MESH)

        def tri_partner(l):
            return int(self.shape_type[l]) in tri_partner_types

        is_mesh_tri_pair = ~is_sdf_pair & np.array([(mesh_like(a) and tri_partner(b)) or (mesh_like(b) and tri_partner(a))
                                                    for a, b in zip(self.pair_a, self.pair_b)], dtype=bool)
        is_sdf_pair = is_sdf_pair | is_mesh_tri_pair
        sp = [(min(newton_id0(a), newton_id0(b)), max(newton_id0(a), newton_id0(b)), int(a), int(b), bool(h), bool(mp), bool(mt))
              for a, b, h, mp, mt in zip(self.pair_a[is_sdf_pair], self.pair_b[is_sdf_pair], is_hydro_pair[is_sdf_pair],
                                         is_mesh_plane_pair[is_sdf_pair], is_mesh_tri_pair[is_sdf_pair])]
        sp.sort()
        self.sdf_pair = np.asarray([[r[2], r[3]] if newton_id0(r[2]) < newton_id0(r[3]) else [r[3], r[2]] for r in sp], dtype=np.int32).reshape(-1, 2)
        self.sdf_pair_hydro = np.asarray([r[4] for r in sp], dtype=bool)  # hydroelastic when the pipeline enables it
        self.sdf_pair_mesh_plane = np.asarray([r[5] for r in sp], dtype=bool)  # the vertex leg (pair kind 2)
        self.sdf_pair_mesh_tri = np.asarray([r[6] for r in sp], dtype=bool)  # the triangle leg (pair kind 3)
        self.sdf_pair_has_edges = np.asarray([bool(has_sdf[r[2]] and has_sdf[r[3]]) for r in sp], dtype=bool)
        self.tile_pair_index = np.flatnonzero(~is_sdf_pair)  # positions of the tile pairs in one world's shape_contact_pairs slice
        self.pair_a, self.pair_b = self.pair_a[~is_sdf_pair], self.pair_b[~is_sdf_pair]
        self.np = len(self.pair_a)

        # The reference writes analytic-primitive contacts in its first narrow-phase kernel and queues every other
        # pair for the GJK/MPR kernel (narrow_phase.py:642-655,1004-1014), so in append order all analytic contacts
        # precede all convex ones.  Device pairs are stored in that order (stable partition); `pair_order` maps a
        # device pair index back to the per-env position in Model.shape_contact_pairs.
        # Barrel cylinders (scale z = radius of the side arc, builder.py:7050-7089): a plane pair is analytic only while the cylinder
        # rests on an end cap, a sphere pair never (narrow_phase.py:682-686,847) -- no fixed route, so they sit with the convex pairs
        # (manifold slots) and the kernel decides per environment and substep.
        def barrel(l):
            if int(self.shape_type[l]) != GeoType.CYLINDER:
                return False
            z = scale_all[newton_id0(l) + (np.arange(E) * ns if l < ns else 0), 2] != 0.0
            if not np.all(z == z.flat[0]):
                raise NotImplementedError("heterogeneous worlds: a cylinder is a barrel in some worlds and straight in others")
            return bool(z.flat[0])

        is_barrel = np.array([barrel(l) for l in range(ns + self.ng)], dtype=bool)

        def analytic(a, b):
            return pair_types_analytic(int(self.shape_type[a]), int(self.shape_type[b])) and not (is_barrel[a] or is_barrel[b])

        convex_types = (GeoType.SPHERE, GeoType.CAPSULE, GeoType.ELLIPSOID, GeoType.CYLINDER, GeoType.BOX, GeoType.CONE,
                        GeoType.CONVEX_MESH)
        is_analytic = np.array([analytic(a, b) for a, b in zip(self.pair_a, self.pair_b)], dtype=bool)

        def convex_ok(s):
            ty = int(self.shape_type[s])
            # infinite planes become a box proxy under the other shape (collision_core.py:562-625), finite ones are
            # rectangles with their own support map (support_function.py:334-345)
            return ty == GeoType.PLANE or ty in convex_types

        for a, b, ok in zip(self.pair_a, self.pair_b, is_analytic):
            both_planes = int(self.shape_type[a]) == GeoType.PLANE and int(self.shape_type[b]) == GeoType.PLANE
            if not ok and (both_planes or not (convex_ok(a) and convex_ok(b))):
                raise NotImplementedError(
                    f"collision pair ({GeoType(int(self.shape_type[a])).name}, {GeoType(int(self.shape_type[b])).name}) "
                    "has no analytic path and is outside the convex (MPR/GJK) scope of this build")
        self.pair_order = np.concatenate([np.flatnonzero(is_analytic), np.flatnonzero(~is_analytic)]).astype(np.int64)
        self.pair_a, self.pair_b = self.pair_a[self.pair_order], self.pair_b[self.pair_order]
        self.np_analytic = int(is_analytic.sum())
        # contact slots per pair: 4 if every pair has an analytic path, else 5 (manifold of 4 + deepest point)
        self.cpp = 4 if self.np_analytic == self.np else 5

        # ordered incidence lists
        bj = [[] for _ in range(nb)]
        for j in range(nj):
            p, c = int(self.joint_parent[j]), int(self.joint_child[j])
            if p >= 0:
                bj[p].append(2 * j)
            if c >= 0:
                bj[c].append(2 * j + 1)
        self.body_joint_start = np.cumsum([0] + [len(x) for x in bj]).astype(np.int32)
        self.body_joint_list = np.zeros(2 * nj, dtype=np.int32)  # padded to 2*nj (C-ABI contract)
        flat = [c for x in bj for c in x]
        self.body_joint_list[:len(flat)] = flat
        bp = [[] for _ in range(nb)]
        for p in range(self.np):
            ba, bb = int(self.shape_body[self.pair_a[p]]), int(self.shape_body[self.pair_b[p]])
            if ba >= 0:
                bp[ba].append(2 * p)
            if bb >= 0:
                bp[bb].append(2 * p + 1)
        self.body_pair_start = np.cumsum([0] + [len(x) for x in bp]).astype(np.int32)
        self.body_pair_list = np.zeros(2 * self.np, dtype=np.int32)  # padded to 2*np (C-ABI contract)
        flat = [c for x in bp for c in x]
        self.body_pair_list[:len(flat)] = flat


def pack_param_arrays(model, t) -> dict:
    """Per-env parameter tables as env-major SoA numpy arrays ([comp, slots, env_stride]; gravity [3, env_stride]; the
    shared global-shape table in AoS): what DeviceModel uploads and what the nt_model descriptor points at."""
    E, ES = t.env_count, t.env_stride

    def soa(aos, n):  # aos [E*n, comp] -> [comp, n, ES]
        if n == 0:  # e.g. a model without shapes or joints
            return np.zeros((1, 1, ES), dtype=np.float32)
        aos = np.asarray(aos, dtype=np.float32).reshape(E, n, -1)
        out = np.zeros((aos.shape[2], n, ES), dtype=np.float32)
        out[:, :, :E] = aos.transpose(2, 1, 0)
        return out

    m = model
    nb, nj, nd, ns = t.nb, t.nj, t.nd, t.ns
    body = np.concatenate([
        m.body_com.reshape(-1, 3), m.body_inv_mass.reshape(-1, 1), m.body_inertia.reshape(-1, 9),
        m.body_inv_inertia.reshape(-1, 9), m.body_mass.reshape(-1, 1)], axis=1)
    grav = np.zeros((3, ES), dtype=np.float32)
    body_world = np.asarray(m.body_world).reshape(E, nb)[:, 0] if nb else np.zeros(E, dtype=np.int32)
    grav[:, :E] = m.gravity[body_world].T  # gravity[-1] is the global world (model.py:1300-1304)
    joint = np.concatenate([m.joint_X_p, m.joint_X_c], axis=1) if nj else np.zeros((0, 14), dtype=np.float32)
    # joint_armature_effective (solver_featherstone.py:269-282): 1e10 on the dofs of joints whose child body is
    # kinematic; the armature is only read by the Featherstone kernels
    armature = np.array(m.joint_armature, dtype=np.float32)
    if nd and nj:
        kin = (np.asarray(m.body_flags)[np.asarray(m.joint_child)] & int(BodyFlags.KINEMATIC)) != 0
        dof_joint = np.searchsorted(np.asarray(m.joint_qd_start), np.arange(len(armature)), side="right") - 1
        armature[kin[dof_joint]] = 1.0e10
    dof = np.concatenate([
        m.joint_axis.reshape(-1, 3), m.joint_limit_lower[:, None], m.joint_limit_upper[:, None],
        m.joint_target_ke[:, None], m.joint_target_kd[:, None], m.joint_limit_ke[:, None], m.joint_limit_kd[:, None],
        armature[:, None], m.joint_damping[:, None]], axis=1) if nd else np.zeros((0, 11), dtype=np.float32)
    shape_all = np.concatenate([
        m.shape_transform, m.shape_scale, m.shape_margin[:, None], m.shape_gap[:, None], m.shape_material_mu[:, None],
        m.shape_material_mu_torsional[:, None], m.shape_material_mu_rolling[:, None], m.shape_material_ke[:, None],
        m.shape_material_kd[:, None], m.shape_material_kf[:, None], m.shape_material_ka[:, None],
        m.shape_material_restitution[:, None]], axis=1)
    new = {
        "body_param": soa(body, nb), "gravity": grav, "joint_param": soa(joint, nj), "dof_param": soa(dof, nd),
        "shape_param": soa(shape_all[t.shape_local0:t.shape_local0 + E * ns], ns),
        "gshape_param": np.ascontiguousarray(shape_all[t.gshape_id], dtype=np.float32),
    }
    return new


def params_uniform(packed: dict, env_count: int) -> int:
    """1 when the body / joint / dof / shape parameter rows of `pack_param_arrays` are bit-identical in every environment
    (replicated worlds without per-world randomisation): nt_model.params_uniform then lets the XPBD rollout keep one
    block-shared copy in LDS instead of one per environment.  NT_PARAMS_UNIFORM=0 forces the per-environment tile (A/B)."""
    if os.environ.get("NT_PARAMS_UNIFORM", "1") == "0":
        return 0
    for k in ("body_param", "joint_param", "dof_param", "shape_param"):
        a = np.ascontiguousarray(packed[k], dtype=np.float32).view(np.int32)
        a = a.reshape(-1, a.shape[-1])[:, :env_count]
        if a.shape[1] > 1 and not np.array_equal(a, np.broadcast_to(a[:, :1], a.shape)):
            return 0
    return 1


def choose_contact_scratch(lib, desc) -> None:
    """Pair-heavy scenes: when the per-contact solver records do not fit the CU's LDS even with one environment per
    workgroup, keep them in HBM (nt_model.contact_scratch_in_hbm; Contacts then allocates nt_contacts.cw)."""
    desc.contact_scratch_in_hbm = 0
    if lib.nt_pick_envs_per_block(C.byref(desc), 0) == 0:
        desc.contact_scratch_in_hbm = 1
        if lib.nt_pick_envs_per_block(C.byref(desc), 0) == 0:
            desc.contact_scratch_in_hbm = 0  # does not fit either way: the launches report NT_ERR_UNSUPPORTED


def newton_model_struct(model):
    """nt_newton_model (include/newton_hip.h) over the flat arrays of a finalized Model -- the arguments of nt_model_create /
    nt_model_refresh_params, i.e. exactly what a Newton host hands over -- plus the numpy buffers that must stay alive while the C
    side reads them."""
    keep = []

    def i32(a):
        x = np.ascontiguousarray(np.asarray(a), dtype=np.int32)
        keep.append(x)
        return x.ctypes.data

    def f32(a):
        x = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
        keep.append(x)
        return x.ctypes.data

    m, s = model, _lib.nt_newton_model()
    s.world_count = m.world_count
    s.body_count, s.joint_count, s.shape_count = len(m.body_world), len(m.joint_world), len(m.shape_type)
    s.joint_dof_count, s.joint_coord_count, s.joint_target_q_count = m.joint_dof_count, m.joint_coord_count, len(m.joint_target_q)
    s.articulation_count = int(getattr(m, "articulation_count", 0))
    pairs = np.asarray(m.shape_contact_pairs, dtype=np.int32).reshape(-1, 2)
    s.shape_contact_pair_count = len(pairs)
    pts = np.asarray(getattr(m, "mesh_points", np.zeros((0, 3))), dtype=np.float32).reshape(-1, 3)
    s.mesh_point_count = len(pts)
    g = np.asarray(m.gravity, dtype=np.float32).reshape(-1, 3)
    s.gravity_count = len(g)
    for k in ("body_world", "body_flags", "joint_world", "joint_type", "joint_parent", "joint_child", "joint_q_start",
              "joint_qd_start", "joint_target_q_start", "joint_dof_dim", "shape_world", "shape_body", "shape_type", "shape_flags",
              "shape_collision_group"):
        setattr(s, k, i32(getattr(m, k)))
    s.joint_enabled = i32(np.asarray(m.joint_enabled, dtype=np.int32))
    if s.articulation_count:
        s.articulation_start, s.articulation_end = i32(m.articulation_start), i32(m.articulation_end)
    for k in ("body_com", "body_mass", "body_inv_mass", "body_inertia", "body_inv_inertia", "joint_X_p", "joint_X_c", "joint_axis",
              "joint_limit_lower", "joint_limit_upper", "joint_target_ke", "joint_target_kd", "joint_limit_ke", "joint_limit_kd",
              "joint_armature", "joint_damping", "shape_transform", "shape_scale", "shape_margin", "shape_gap", "shape_material_mu",
              "shape_material_mu_torsional", "shape_material_mu_rolling", "shape_material_ke", "shape_material_kd",
              "shape_material_kf", "shape_material_ka", "shape_material_restitution"):
        setattr(s, k, f32(getattr(m, k)))
    s.shape_contact_pairs = i32(pairs)
    if hasattr(m, "shape_mesh_start"):
        s.shape_mesh_start, s.shape_mesh_count = i32(m.shape_mesh_start), i32(m.shape_mesh_count)
    s.mesh_points, s.gravity = f32(pts), f32(g)
    if getattr(m, "_shape_sdf_index", None) is not None:
        s.shape_sdf_index = i32(m._shape_sdf_index)
    if getattr(m, "shape_edge_range", None) is not None:
        s.shape_edge_range = i32(np.asarray(m.shape_edge_range).reshape(-1, 2))
    return s, keep


def c_sdf_pairs(lib, handle):
    """nt_model_sdf_pairs: the pairs of one world the C builder routed out of the tiles -> (pairs [n, 2], kind [n], has_edges [n])."""
    n = C.c_int32()
    _lib.check(lib.nt_model_sdf_pairs(handle, C.byref(n), None, None, None), "nt_model_sdf_pairs")
    pairs, kind, edges = np.zeros((n.value, 2), np.int32), np.zeros(n.value, np.uint8), np.zeros(n.value, np.uint8)
    if n.value:
        _lib.check(lib.nt_model_sdf_pairs(handle, C.byref(n), pairs.ctypes.data, kind.ctypes.data, edges.ctypes.data), "nt_model_sdf_pairs")
    return pairs, kind, edges


class DeviceModel:
    """Device-resident env-major SoA copy of a Model + the nt_model descriptor passed across the C ABI.

    The descriptor the kernels get is the one ``nt_model_create`` builds in C from the model's flat arrays
    (csrc/nt_model_build.hip) -- the entry point a non-Python Newton host binds -- including the routing of SDF / hydroelastic /
    mesh-vertex pairs out of the tiles (``nt_model_sdf_pairs``).  The tables built here in Python (`EnvTemplate`,
    `pack_param_arrays`) stay as the host-side mirror (world slicing, the SDF legs and the tests read them); tests/test_model_build.py
    holds them equal to the C ones table by table, and the constructor compares the pair routing of the two on every model."""

    def __init__(self, model: Model):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.NewtonHipError("no GPU visible: the HIP path needs an MI355X (there is no CPU fallback)")
        self.device = torch.device(model.device)
        t = model.env
        self.t = t
        E, ES = t.env_count, t.env_stride
        self._keep = []

        def dev_i32(a):
            x = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
            if x.numel() == 0:
                x = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._keep.append(x)
            return x

        def dev_f32(a):
            x = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
            if x.numel() == 0:
                x = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._keep.append(x)
            return x

        # The tables the kernels read live in ONE place on the device: the handle nt_model_create builds (below).  The Python-built
        # tables are uploaded only when that builder is switched off (tests compare the two end to end); otherwise they stay on the
        # host as the mirror (world slicing, the SDF legs, tests) -- uploading both kept every per-env parameter table twice.
        self._c_handle = None
        use_c = self._c_builder_eligible(model)
        self.topology, self.mesh_tables, self.params = {}, {}, {}
        if not use_c:
            self.topology = {k: dev_i32(getattr(t, k)) for k in (
                "body_flags", "joint_type", "joint_enabled", "joint_parent", "joint_child", "joint_q_start",
                "joint_qd_start", "joint_tq_start", "joint_lin_count", "joint_ang_count", "shape_body", "shape_type",
                "shape_flags", "shape_group", "pair_a", "pair_b", "body_joint_start", "body_joint_list", "body_pair_start",
                "body_pair_list", "art_start", "shape_mesh_start", "shape_mesh_count", "gshape_id")}
            self.topology["shape_type"] = dev_i32(t.tile_shape_type)  # (triangle meshes as pre-computed-AABB shapes, see EnvTemplate)
            self.mesh_tables = {"mesh_points": dev_f32(t.mesh_points), "shape_mesh_bounds": dev_f32(t.shape_mesh_bounds)}
        self.upload_params(model, to_device=not use_c)
        d = _lib.nt_model()
        d.env_count, d.env_stride = E, ES
        d.nb, d.nj, d.nd, d.nc, d.ntq, d.ns, d.ng, d.np, d.cpp = t.nb, t.nj, t.nd, t.nc, t.ntq, t.ns, t.ng, t.np, t.cpp
        d.np_analytic = t.np_analytic
        d.na, d.max_art_dofs = t.na, t.max_art_dofs
        d.shape_local0 = t.shape_local0
        for k, v in self.topology.items():
            setattr(d, k, v.data_ptr())
        for k, v in self.params.items():
            setattr(d, k, v.data_ptr())
        for k, v in self.mesh_tables.items():
            setattr(d, k, v.data_ptr())
        d.params_uniform = self._params_uniform
        choose_contact_scratch(self.lib, d)
        self.desc = d
        if use_c:
            # the C ABI's own builder: flat Newton arrays in, device descriptor out (the Python tables above stay as the host-side
            # mirror: world slicing, the SDF legs and the tests read them)
            src, keep = newton_model_struct(model)
            h = C.c_void_p()
            with torch.cuda.device(self.device):  # (the handle's tables are allocated on the calling thread's current HIP device)
                rc = self.lib.nt_model_create(C.byref(src), 1, C.byref(h))
            if rc != 0:
                raise _lib.NewtonHipError(f"nt_model_create: {self.lib.nt_model_last_error().decode()}")
            self._c_handle = h
            cd = _lib.nt_model()
            C.memmove(C.byref(cd), self.lib.nt_model_get(h), C.sizeof(cd))
            assert (cd.nb, cd.nj, cd.np, cd.ns, cd.ng, cd.cpp, cd.np_analytic, cd.env_count, cd.env_stride, cd.contact_scratch_in_hbm,
                    cd.params_uniform) == \
                (d.nb, d.nj, d.np, d.ns, d.ng, d.cpp, d.np_analytic, d.env_count, d.env_stride, d.contact_scratch_in_hbm,
                 d.params_uniform), "nt_model_create and the host mirror disagree on the model's sizes / tile mode"
            # pairs routed out of the tiles: the SDF legs read the Python mirror (t.sdf_pair ...), the kernels the C tables
            sp, kind, edges = c_sdf_pairs(self.lib, h)
            want_kind = (np.where(t.sdf_pair_hydro, 1, np.where(t.sdf_pair_mesh_plane, 2, np.where(t.sdf_pair_mesh_tri, 3, 0))).astype(np.uint8)
                         if len(t.sdf_pair) else kind[:0])
            if not (np.array_equal(sp, np.asarray(t.sdf_pair).reshape(-1, 2)) and np.array_equal(kind, want_kind)
                    and np.array_equal(edges.astype(bool), np.asarray(t.sdf_pair_has_edges, dtype=bool))):
                raise _lib.NewtonHipError("nt_model_create routed the SDF / vertex pairs differently from the host mirror")
            self.desc = cd
        # environments per workgroup the collide / XPBD / SemiImplicit kernels will use (0: the working set of one
        # environment does not fit the CU's LDS in either mode)
        self.envs_per_block = int(self.lib.nt_pick_envs_per_block(C.byref(self.desc), 0))  # (of the descriptor the kernels launch with)
        self.lds_bytes_per_env = int(self.lib.nt_lds_bytes_per_env(C.byref(self.desc)))

    USE_C_BUILDER = True  # (tests flip it to compare the two builders end to end)

    def _c_builder_eligible(self, model) -> bool:
        return bool(DeviceModel.USE_C_BUILDER)

    def __del__(self):
        h, self._c_handle = getattr(self, "_c_handle", None), None
        if h is not None and self.lib is not None:
            try:
                self.lib.nt_model_destroy(h)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def require_fit(self, what: str):
        if self.envs_per_block == 0:
            raise NotImplementedError(
                f"{what}: one environment of this model needs {self.lds_bytes_per_env / 1024:.0f} KB of LDS (plus its topology "
                "tables) and the CU has 160 KB; the scene is too large for the LDS-resident kernels of this build "
                f"({self.t.nb} bodies, {self.t.ns} shapes, {self.t.np} candidate pairs per environment)")

    def refresh_flags(self, model: Model):
        """Re-derive the env-uniform flag tables a runtime edit may touch (body_flags, joint_enabled, shape_flags,
        shape_collision_group) and copy them into the resident topology tensors; edits that break the uniformity or change
        the topology itself (parents, types, pairs) raise, because the kernels' tables were built for the old one."""
        t = self.t
        E, nb, nj, ns = t.env_count, t.nb, t.nj, t.ns
        L0 = t.shape_local0

        def uniform(a, n, what):
            a = np.asarray(a).reshape(E, n) if n else np.zeros((E, 0), dtype=np.int32)
            if E > 1 and not np.all(a == a[0:1]):
                raise NotImplementedError(f"heterogeneous worlds: {what} differs between worlds after a runtime edit")
            return a[0].astype(np.int32)

        def shape_tab(a, what):
            a = np.asarray(a)
            return np.concatenate([uniform(a[L0:L0 + E * ns], ns, what), a[t.gshape_id].astype(np.int32)])

        new = {"body_flags": uniform(model.body_flags, nb, "body_flags"),
               "joint_enabled": uniform(np.asarray(model.joint_enabled, dtype=np.int32), nj, "joint_enabled"),
               "shape_flags": shape_tab(model.shape_flags, "shape_flags"),
               "shape_group": shape_tab(model.shape_collision_group, "shape_collision_group")}
        torch = _torch()
        for k, v in new.items():
            if not np.array_equal(v, getattr(t, k)):
                setattr(t, k, v)
                if v.size and k in self.topology:  # (the Python-built device tables, when they exist)
                    self.topology[k][: v.size].copy_(torch.from_numpy(v))
        if nj and not np.array_equal(uniform(model.joint_type, nj, "joint_type"), t.joint_type):
            raise NotImplementedError("joint_type changed after finalize(): rebuild the model (the kernels' topology tables "
                                      "are static)")

    def upload_params(self, model: Model, to_device=None):
        """(Re)build the per-env parameter SoA arrays from the model's AoS numpy arrays.  to_device: also keep them on the device
        (default: only when the Python-built descriptor is the one in use -- with the C handle its own tables are refreshed)."""
        torch = _torch()
        if to_device is None:
            to_device = getattr(self, "_c_handle", None) is None
        self.refresh_flags(model)
        new = pack_param_arrays(model, self.t)
        self._params_uniform = params_uniform(new, self.t.env_count)
        if getattr(self, "desc", None) is not None:
            self.desc.params_uniform = self._params_uniform
        for k, v in (new.items() if to_device else ()):
            if v.size == 0:
                v = np.zeros(1, dtype=np.float32)
            if k in self.params and self.params[k].numel() == v.size:
                self.params[k].copy_(torch.from_numpy(v).reshape(self.params[k].shape))
            else:
                self.params[k] = torch.from_numpy(v).to(self.device)
        if getattr(self, "_c_handle", None) is not None:  # Model.notify_model_changed(): the C-built tables follow
            src, keep = newton_model_struct(model)
            with torch.cuda.device(self.device):
                rc = self.lib.nt_model_refresh_params(self._c_handle, C.byref(src))
            if rc != 0:
                raise _lib.NewtonHipError(f"nt_model_refresh_params: {self.lib.nt_model_last_error().decode()}")
            C.memmove(C.byref(self.desc), self.lib.nt_model_get(self._c_handle), C.sizeof(self.desc))

    def stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)


class Model:
    """Flat-array simulation model (newton/_src/sim/model.py:808-1364, rigid subset)."""

    EXTENDED_STATE_ATTRIBUTES = ("body_parent_f",)  # state.py:73-79, rigid subset
    EXTENDED_CONTACT_ATTRIBUTES = ("force",)  # contacts.py:170-226

    def __init__(self, device=None):
        self.device = device if device is not None else "cpu"
        self.requires_grad = False
        self._requested_state_attributes: set[str] = set()
        self._requested_contact_attributes: set[str] = set()
        self.world_count = 0
        self.particle_count = 0
        self.rigid_contact_max = 0
        self.env: EnvTemplate | None = None
        self._dev: DeviceModel | None = None

    # -- construction helpers -------------------------------------------------------------------
    def _build_env_template(self):
        """One topology for all worlds -> the EnvTemplate the fused kernels run on.  Worlds that differ (model.py:881-900 permits
        them) leave ``env`` unset; the model is then served through its world groups (hetero.py)."""
        self._world_groups = None
        self._hetero_reason = None
        try:
            self.env = EnvTemplate(self)
        except NotImplementedError as e:
            if not str(e).startswith("heterogeneous worlds") or self.world_count <= 1:
                raise
            self.env = None
            self._hetero_reason = str(e)

    @property
    def is_heterogeneous(self) -> bool:
        return getattr(self, "_hetero_reason", None) is not None

    @property
    def world_groups(self):
        """The homogeneous sub-models (maximal runs of worlds with one topology) of a heterogeneous model."""
        if not self.is_heterogeneous:
            raise ValueError("world_groups: all worlds of this model share one topology")
        if self._world_groups is None:
            from .hetero import WorldGroups  # noqa: PLC0415

            self._world_groups = WorldGroups(self)
        return self._world_groups

    @property
    def is_gpu(self):
        return _is_gpu(self.device)

    def device_model(self) -> DeviceModel:
        if not self.is_gpu:
            raise _lib.NewtonHipError(
                f"Model is on device '{self.device}': the solvers / collision pipeline of newton_amd run only on an "
                "MI355X through libnewton_hip.so (no CPU fallback). Finalize with device='cuda:0'.")
        if self.is_heterogeneous:
            raise NotImplementedError(f"{self._hetero_reason}: one device descriptor serves one topology; use the world groups "
                                      "(model.world_groups, or model.state() / the solver classes, which dispatch to them)")
        if self._dev is None:
            self._dev = DeviceModel(self)
        return self._dev

    @staticmethod
    def _check_requested(attributes, known, kind):
        unknown = [a for a in attributes if a not in known]
        if unknown:
            raise ValueError(f"Unknown extended {kind} attribute(s): {unknown}; supported: {list(known)}")

    def request_state_attributes(self, *attributes: str) -> None:
        """model.py:2068-2078: allocate these extended attributes in every State created afterwards."""
        self._check_requested(attributes, self.EXTENDED_STATE_ATTRIBUTES, "state")
        self._requested_state_attributes.update(attributes)

    def request_contact_attributes(self, *attributes: str) -> None:
        """model.py:2080-2088: allocate these extended attributes in every Contacts created afterwards."""
        self._check_requested(attributes, self.EXTENDED_CONTACT_ATTRIBUTES, "contact")
        self._requested_contact_attributes.update(attributes)

    def get_requested_state_attributes(self) -> list[str]:
        return sorted(self._requested_state_attributes)

    def get_requested_contact_attributes(self) -> set[str]:
        return set(self._requested_contact_attributes)

    def set_gravity(self, gravity, world: int | None = None) -> None:
        """Runtime gravity change (model.py:1908-1960): one vector for every world, one per local world, or one per local
        world plus the global entry; ``world`` selects a single world (-1 = global).  Call
        ``solver.notify_model_changed(ModelFlags.MODEL_PROPERTIES)`` afterwards."""
        g = np.asarray(gravity, dtype=np.float32)
        W = self.world_count
        if world is not None:
            if g.shape != (3,):
                raise ValueError("Expected single gravity vector (3,) when world is specified")
            if world < -1 or world >= W:
                raise IndexError(f"world {world} out of range; expected -1 or [0, {W})")
            self.gravity[world if world >= 0 else W] = g
        elif g.shape == (3,):
            self.gravity[:] = g
        elif g.shape == (W, 3):
            self.gravity[:W] = g
        elif g.shape == (W + 1, 3):
            self.gravity[:] = g
        else:
            raise ValueError(f"gravity must have shape (3,), ({W}, 3) or ({W + 1}, 3), got {g.shape}")

    def notify_model_changed(self):
        """Re-upload per-env parameters after the host arrays were edited."""
        if self.is_heterogeneous:
            if self._world_groups is not None:
                self._world_groups.refresh_parameters()
            return
        if self._dev is not None:
            self._dev.upload_params(self)

    # -- factories (model.py:1779-1863) --------------------------------------------------------------
    def state(self) -> State:
        from .state import State  # noqa: PLC0415

        if self.is_heterogeneous:
            from .hetero import _make_state_classes  # noqa: PLC0415

            return _make_state_classes()[0](self)
        return State(self)

    def control(self, clone_variables: bool = True) -> Control:
        from .state import Control  # noqa: PLC0415

        if self.is_heterogeneous:
            from .hetero import _make_state_classes  # noqa: PLC0415

            return _make_state_classes()[1](self)
        return Control(self)

    def contacts(self):
        from .collide import CollisionPipeline  # noqa: PLC0415

        return CollisionPipeline(self).contacts()

    def shape_collision_filter_mask(self, pairs):
        pairs = np.asarray(pairs).reshape(-1, 2)
        return np.asarray([(min(a, b), max(a, b)) in self.shape_collision_filter_pairs for a, b in pairs], dtype=bool)
