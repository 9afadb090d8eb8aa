#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/broadphase_reference_vectors.npz by EXECUTING the reference's own broad-phase
classes in this container (see tests/golden/refshim): precompute_world_map (newton/_src/geometry/broad_phase_common.py:271-388),
BroadPhaseAllPairs / BroadPhaseExplicit (broad_phase_nxn.py:221-535) and BroadPhaseSAP (broad_phase_sap.py:513-849, segmented
sort) on the random configurations of the reference's tests (tests/test_broad_phase_standalone.py::make_case), with excluded
pairs, the immovable-pair filter and per-shape displacements (swept AABBs: check_aabb_overlap_moving, broad_phase_common.py:41-85;
_sap_project_aabb with sort_axis_displacement_limit, broad_phase_sap.py:44-79).  tests/test_reference_vectors.py compares the checker's world map and candidate lists with
the record: N x N and explicit in append order, sort-and-sweep as a set.
Run from the repo root:  python tests/golden/make_broadphase_reference_vectors.py"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "refshim"))
import lazy_ref  # noqa: E402

lazy_ref.install()
import warp as wp  # noqa: E402  (the stand-in)

common = importlib.import_module("newton._src.geometry.broad_phase_common")
nxn = importlib.import_module("newton._src.geometry.broad_phase_nxn")
sap = importlib.import_module("newton._src.geometry.broad_phase_sap")


def arr(a, dtype):
    return wp.to_array(np.asarray(a), dtype)


def variants(name):
    """-> list of (tag, kwargs for the case): plain, with excluded pairs, with the immovable-pair filter."""
    import broadphase_cases as bc

    return bc.variants(name)


def run(kind, lower, upper, gap, group, world, flags, filter_pairs, shape_body, body_flags, include, explicit_pairs=None,
        displacement=None, limit=None):
    n = lower.shape[0]
    cap = n * (n - 1) // 2 + 1
    cand, cnt = wp.zeros(cap, dtype=wp.vec2i), wp.zeros(1, dtype=wp.int32)
    kw = dict(filter_pairs=arr(filter_pairs, wp.vec2i) if filter_pairs is not None and len(filter_pairs) else None,
              shape_body=arr(shape_body, wp.int32) if shape_body is not None else None,
              body_flags=arr(body_flags, wp.int32) if body_flags is not None else None, include_static_kinematic_pairs=include)
    lo, up, g = arr(lower, wp.vec3), arr(upper, wp.vec3), arr(gap, wp.float32)
    if displacement is not None:  # swept AABBs (check_aabb_overlap_moving, _sap_project_aabb)
        kw["shape_displacement"] = arr(displacement, wp.vec3)
        if kind == "sap":
            kw["sort_axis_displacement_limit"] = limit
    if kind == "explicit":
        bp = nxn.BroadPhaseExplicit()
        kw.pop("filter_pairs")
        bp.launch(lo, up, g, arr(explicit_pairs, wp.vec2i), len(explicit_pairs), cand, cnt, **kw)
    else:
        cls = nxn.BroadPhaseAllPairs if kind == "nxn" else sap.BroadPhaseSAP
        bp = cls(world, flags, device="cpu")
        bp.launch(lo, up, g, arr(group, wp.int32), arr(world, wp.int32), n, cand, cnt, **kw)
    c = int(cnt.numpy()[0])
    return c, cand.numpy().reshape(-1, 2)[:c].astype(np.int32)


def main():
    import broadphase_cases as bc

    out = {}
    for name in bc.CASES:
        for tag, v in bc.variants(name).items():
            key = f"{name}/{tag}"
            index_map, ends = common.precompute_world_map(v["world"], v["flags"])
            out[f"{key}/index_map"], out[f"{key}/slice_ends"] = np.asarray(index_map, np.int32), np.asarray(ends, np.int32)
            for kind in ("nxn", "sap", "explicit"):
                c, pairs = run(kind, v["lower"], v["upper"], v["gap"], v["group"], v["world"], v["flags"], v["filter_pairs"],
                               v["shape_body"], v["body_flags"], v["include"], explicit_pairs=v["explicit_pairs"],
                               displacement=v.get("displacement"), limit=v.get("limit"))
                out[f"{key}/{kind}_pairs"] = pairs
                print(key, kind, c)
    path = os.path.join(HERE, "broadphase_reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", len(out), "arrays to", path)


if __name__ == "__main__":
    main()
