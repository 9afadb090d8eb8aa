"""nt_model_create (include/newton_hip.h, newton_amd/csrc/nt_model_build.hip): the C helper that derives the nt_model descriptor
from newton.Model's flat arrays, against the Python host logic that does the same for the Python binding
(newton_amd/model.py: EnvTemplate, pack_param_arrays, params_uniform) -- every table, on several scenes, in host memory
(on_device = 0: no GPU needed).  The reference arrays it consumes are the Model attributes of newton/_src/sim/model.py:808-1364."""
import ctypes as C
import os

import numpy as np
import pytest

from newton_amd import _lib
from newton_amd.model import pack_param_arrays, params_uniform


def _newton_arrays(model):
    """nt_newton_model over the flat arrays of a finalized Model (the product's own marshalling: newton_amd.model.newton_model_struct)."""
    from newton_amd.model import newton_model_struct

    return newton_model_struct(model)


def _arr(ptr, n, ctype=C.c_int32):
    if n == 0:
        return np.zeros(0, dtype=np.int32 if ctype is C.c_int32 else np.float32)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).copy()


def _scenes():
    from scenes import box_stack_scene, hull_bin_scene, mixed_primitive_scene, pendulum_scene, quadruped_scene

    return {
        "quadruped": lambda: quadruped_scene(5, seed=3),
        "quadruped_convex": lambda: quadruped_scene(3, seed=None, colliders="box"),
        "mixed_primitives": lambda: mixed_primitive_scene(4),
        "box_stack": lambda: box_stack_scene(3, n_boxes=4, seed=1),
        "pendulum": lambda: pendulum_scene(6, seed=2),
        "hull_bin": lambda: hull_bin_scene(2, n_hulls=6),
        # pairs routed out of the tiles (narrow_phase.py:531-538,618-640): mesh-SDF edge pairs incl. static walls, hydroelastic
        # pairs, (triangle mesh, infinite plane) vertex pairs with the plane before and after the meshes
        "sdf_pile": lambda: _sdf_scene(3, 4, walls=True),
        "hydro_bin": lambda: hull_bin_scene(2, 5, seed=2, sdf=True, mu=0.5, shape_cfg=dict(gap=0.005), hydroelastic=True),
        "mesh_ground": lambda: _mesh_scene(3, ground_first=False),
        "mesh_ground_first": lambda: _mesh_scene(2, ground_first=True),
        # barrel cylinders: their plane / sphere pairs have no fixed analytic route and sit with the convex pairs
        "barrel_cylinders": lambda: _barrel_scene(),
    }


def _barrel_scene():
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import collide_cases as cc

    return cc.barrel_cases()["barrel_wide"]()[0]


def _sdf_scene(*a, **k):
    from sdf_pipeline_checker import sdf_scene

    return sdf_scene(*a, device=None, **k)


def _mesh_scene(worlds, ground_first):
    from test_gpu_mesh_plane_pipeline import mesh_scene

    return mesh_scene(worlds, kind="box", device=None, ground_first=ground_first)


@pytest.mark.parametrize("name", list(_scenes()))
def test_c_helper_reproduces_the_python_descriptor(name):
    lib = _lib.load()
    model = _scenes()[name]()
    t = model.env
    src, keep = _newton_arrays(model)
    h = C.c_void_p()
    rc = lib.nt_model_create(C.byref(src), 0, C.byref(h))
    assert rc == 0, lib.nt_model_last_error()
    try:
        d = lib.nt_model_get(h).contents
        for k in ("env_count", "env_stride", "nb", "nj", "nd", "nc", "ntq", "ns", "ng", "np", "cpp", "np_analytic", "na",
                  "max_art_dofs", "shape_local0"):
            assert getattr(d, k) == getattr(t, k), k
        sizes = {"body_flags": t.nb, "joint_type": t.nj, "joint_enabled": t.nj, "joint_parent": t.nj, "joint_child": t.nj,
                 "joint_q_start": t.nj, "joint_qd_start": t.nj, "joint_tq_start": t.nj, "joint_lin_count": t.nj,
                 "joint_ang_count": t.nj, "shape_body": t.ns + t.ng, "shape_type": t.ns + t.ng, "shape_flags": t.ns + t.ng,
                 "shape_group": t.ns + t.ng, "pair_a": t.np, "pair_b": t.np, "body_joint_start": t.nb + 1,
                 "body_joint_list": 2 * t.nj, "body_pair_start": t.nb + 1, "body_pair_list": 2 * t.np, "art_start": t.na + 1,
                 "shape_mesh_start": t.ns + t.ng, "shape_mesh_count": t.ns + t.ng, "gshape_id": t.ng}
        for k, n in sizes.items():
            want = t.tile_shape_type if k == "shape_type" else getattr(t, k)  # (a triangle mesh: a pre-computed-AABB shape to the tiles)
            assert np.array_equal(_arr(getattr(d, k), n), np.asarray(want, dtype=np.int32)[:n]), k
        assert np.array_equal(_arr(d.mesh_points, 3 * len(t.mesh_points), C.c_float), t.mesh_points.reshape(-1))
        assert np.array_equal(_arr(d.shape_mesh_bounds, 6 * (t.ns + t.ng), C.c_float), t.shape_mesh_bounds.reshape(-1))
        packed = pack_param_arrays(model, t)
        for k, v in packed.items():
            got = _arr(getattr(d, k), v.size, C.c_float)
            assert np.array_equal(got.view(np.int32), np.ascontiguousarray(v, dtype=np.float32).reshape(-1).view(np.int32)), k
        assert d.params_uniform == params_uniform(packed, t.env_count)
        if name == "barrel_cylinders":  # no pair of the analytic group holds a barrel; (plane, barrel) and (sphere, barrel) pairs exist
            scale = np.asarray(model.shape_scale, dtype=np.float32).reshape(-1, 3)
            ids = [t.shape_local0 + l if l < t.ns else int(t.gshape_id[l - t.ns]) for l in range(t.ns + t.ng)]
            barrel = np.array([int(t.shape_type[l]) == 6 and scale[ids[l], 2] != 0.0 for l in range(t.ns + t.ng)])
            pa, pb = np.asarray(t.pair_a), np.asarray(t.pair_b)
            assert barrel.sum() >= 13 and not (barrel[pa[:t.np_analytic]] | barrel[pb[:t.np_analytic]]).any()
            kinds = {tuple(sorted((int(t.shape_type[a]), int(t.shape_type[b])))) for a, b in zip(pa[t.np_analytic:], pb[t.np_analytic:])
                     if barrel[a] or barrel[b]}
            assert (1, 6) in kinds and (3, 6) in kinds  # (PLANE, CYLINDER), (SPHERE, CYLINDER) in the convex group
        order = np.zeros(t.np, dtype=np.int64)
        assert lib.nt_model_pair_order(h, order.ctypes.data_as(C.POINTER(C.c_int64))) == 0
        assert np.array_equal(order, np.asarray(t.tile_pair_index)[t.pair_order])  # positions in the world's shape_contact_pairs slice
        # the pairs that leave the tiles, their kinds and order: what the pipeline's SDF leg walks per world
        from newton_amd.model import c_sdf_pairs

        sp, kind, edges = c_sdf_pairs(lib, h)
        assert np.array_equal(sp, np.asarray(t.sdf_pair).reshape(-1, 2))
        assert np.array_equal(kind == 1, t.sdf_pair_hydro) and np.array_equal(kind == 2, t.sdf_pair_mesh_plane)
        assert np.array_equal(edges.astype(bool), t.sdf_pair_has_edges)
        if name in ("sdf_pile", "hydro_bin", "mesh_ground", "mesh_ground_first"):
            assert len(sp) > 0 and set(kind.tolist()) == {dict(sdf_pile=0, hydro_bin=1).get(name, 2)}
        # the mode choice for pair-heavy scenes is the library's own
        from newton_amd.model import choose_contact_scratch

        mine = _lib.nt_model()
        C.memmove(C.byref(mine), C.byref(d), C.sizeof(mine))
        choose_contact_scratch(lib, mine)
        assert mine.contact_scratch_in_hbm == d.contact_scratch_in_hbm
        # notify_model_changed: one world's link gets heavier -> the tables follow, the uniform flag drops
        if t.nb and t.env_count > 1:
            model.body_mass = np.array(model.body_mass, copy=True)
            model.body_mass[t.nb] *= 2.0
            src2, keep2 = _newton_arrays(model)
            assert lib.nt_model_refresh_params(h, C.byref(src2)) == 0
            packed = pack_param_arrays(model, t)
            got = _arr(d.body_param, packed["body_param"].size, C.c_float)
            assert np.array_equal(got, packed["body_param"].reshape(-1)) and d.params_uniform == 0
        # ... and the env-uniform flag tables follow a runtime edit too (joint_enabled / body_flags / shape_flags / collision groups)
        if t.nj:
            model.joint_enabled = np.array(model.joint_enabled, copy=True)
            model.joint_enabled[t.nj - 1::t.nj] = False  # the last joint of every world
            src3, keep3 = _newton_arrays(model)
            assert lib.nt_model_refresh_params(h, C.byref(src3)) == 0
            want = np.ones(t.nj, dtype=np.int32)
            want[:] = np.asarray(model.joint_enabled, dtype=np.int32)[: t.nj]
            assert np.array_equal(_arr(d.joint_enabled, t.nj), want) and want[-1] == 0
            model.joint_enabled[t.nj - 1] = True  # one world differs: refused with a reason
            src4, keep4 = _newton_arrays(model)
            if t.env_count > 1:
                assert lib.nt_model_refresh_params(h, C.byref(src4)) != 0 and b"joint_enabled" in lib.nt_model_last_error()
    finally:
        lib.nt_model_destroy(h)


def test_heterogeneous_worlds_are_refused_with_a_reason():
    from scenes import quadruped_scene

    lib = _lib.load()
    model = quadruped_scene(3, seed=None)
    model.joint_type = np.array(model.joint_type, copy=True)
    model.joint_type[model.env.nj + 2] = 3  # FIXED instead of REVOLUTE in world 1
    src, keep = _newton_arrays(model)
    h = C.c_void_p()
    assert lib.nt_model_create(C.byref(src), 0, C.byref(h)) == -3 and not h.value
    assert b"joint_type differs between worlds" in lib.nt_model_last_error()
