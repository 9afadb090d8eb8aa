#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/match_reference_vectors.npz by EXECUTING the reference's ContactMatcher
(newton/_src/geometry/contact_match.py:602-1055, non-sticky) on the Warp stand-in (tests/golden/refshim): key-sorted contacts
of consecutive frames of small scenes (contacts from the in-repo checker's collide, sorted with the reference's contact key),
save_sorted_state on frame k, match on frame k + 1.  tests/test_reference_vectors.py holds oracle/oracle_match.py against it."""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), HERE, os.path.join(HERE, "refshim")):
    sys.path.insert(0, p)
import lazy_ref  # noqa: E402

lazy_ref.install(dummies={"newton._src.sim": ("Contacts", "Control", "Model", "State", "ModelBuilder"),
                          "newton._src.geometry": ("ContactSorter",)},
                 dummy_modules=("newton._src.geometry.contact_sort",))
import warp as wp  # noqa: E402

cm = importlib.import_module("newton._src.geometry.contact_match")


def arr(a, dtype):
    return wp.to_array(np.asarray(a), dtype)


def sorted_frame(model, orc, body_q):
    """The checker's contacts of one frame in the reference's deterministic order (contact_data.py:60-90)."""
    import oracle_match as om

    ct = orc.contacts()
    orc.collide(body_q, ct)
    n = int(ct.count[0])
    sub = np.zeros(n, np.int64)
    for i in range(1, n):
        same = ct.shape0[i] == ct.shape0[i - 1] and ct.shape1[i] == ct.shape1[i - 1]
        sub[i] = sub[i - 1] + 1 if same else 0
    keys = np.array([om.sort_key(a, b, k) for a, b, k in zip(ct.shape0[:n], ct.shape1[:n], sub)], np.int64)
    order = np.argsort(keys, kind="stable")
    return {"keys": keys[order], "shape0": ct.shape0[:n][order], "shape1": ct.shape1[:n][order], "point0": ct.point0[:n][order],
            "point1": ct.point1[:n][order], "normal": ct.normal[:n][order], "offset0": ct.offset0[:n][order],
            "offset1": ct.offset1[:n][order], "margin0": ct.margin0[:n][order], "margin1": ct.margin1[:n][order]}


def main():
    import oracle_bridge as ob
    from scenes import box_stack_scene, mixed_primitive_scene

    blob = {}
    scenes = {"box_stack": (box_stack_scene(1, n_boxes=4, seed=2, jitter=5e-3), 1.0 / 240.0, 4, 6),
              "mixed_primitives": (mixed_primitive_scene(1, seed=4), 2e-3, 2, 8)}
    for name, (model, dt, iters, frames) in scenes.items():
        orc = ob.Oracle(model)
        a, b = ob.OracleState(model), ob.OracleState(model)
        cap = 256
        sorter = types.SimpleNamespace(scratch_pos_world=wp.zeros(cap, dtype=wp.vec3), scratch_normal=wp.zeros(cap, dtype=wp.vec3))
        matcher = cm.ContactMatcher(cap, sorter=sorter, shape_world=arr(model.shape_world, int), world_count=model.world_count)
        sorter2 = types.SimpleNamespace(scratch_pos_world=wp.zeros(cap, dtype=wp.vec3), scratch_normal=wp.zeros(cap, dtype=wp.vec3))
        sticky = cm.ContactMatcher(cap, sorter=sorter2, shape_world=arr(model.shape_world, int), world_count=model.world_count,
                                   sticky=True)
        shape_body = arr(model.shape_body, int)
        for k in range(frames):
            fr = sorted_frame(model, orc, a.body_q)
            n = len(fr["keys"])
            pad = lambda x, d: wp.to_array(np.concatenate([x, np.zeros((cap - n, *x.shape[1:]), x.dtype)]), d)  # noqa: E731
            bq = arr(a.body_q, wp.transform)
            out = wp.full(cap, -5, dtype=int)
            args = (pad(fr["keys"], int), arr(np.array([n]), int), pad(fr["point0"], wp.vec3), pad(fr["point1"], wp.vec3),
                    pad(fr["shape0"], int), pad(fr["shape1"], int), pad(fr["normal"], wp.vec3), bq, shape_body)
            matcher.match(*args, out)
            matcher.save_sorted_state(*args)
            # sticky mode on the same fresh contacts: match -> (already sorted) -> replay_matched -> save_sorted_state
            sargs = (pad(fr["keys"], int), arr(np.array([n]), int), pad(fr["point0"], wp.vec3), pad(fr["point1"], wp.vec3),
                     pad(fr["shape0"], int), pad(fr["shape1"], int), pad(fr["normal"], wp.vec3), bq, shape_body)
            sout = wp.full(cap, -5, dtype=int)
            sticky.match(*sargs, sout)
            off0, off1 = pad(fr["offset0"], wp.vec3), pad(fr["offset1"], wp.vec3)
            sticky.replay_matched(sargs[1], sout, point0=sargs[2], point1=sargs[3], offset0=off0, offset1=off1, normal=sargs[6],
                                  shape0=sargs[4], shape1=sargs[5], margin0=pad(fr["margin0"], float), margin1=pad(fr["margin1"], float),
                                  body_q=bq, shape_body=shape_body)
            sticky.save_sorted_state(*sargs, sorted_offset0=off0, sorted_offset1=off1)
            v3 = lambda a_: np.array([[float(c) for c in x] for x in a_[:n]], np.float32).reshape(n, 3)  # noqa: E731
            blob[f"{name}/{k}/sticky_match"] = np.array(sout[:n], np.int32)
            for f, a_ in (("point0", sargs[2]), ("point1", sargs[3]), ("offset0", off0), ("offset1", off1), ("normal", sargs[6])):
                blob[f"{name}/{k}/sticky_{f}"] = v3(a_)
            for f, v in fr.items():
                blob[f"{name}/{k}/{f}"] = v
            blob[f"{name}/{k}/body_q"] = np.array(a.body_q, np.float32)
            blob[f"{name}/{k}/match"] = np.array(out[:n], np.int32)
            print(name, "frame", k, "contacts", n, "matched", int((np.array(out[:n]) >= 0).sum()), "broken", int((np.array(out[:n]) == -2).sum()),
                  flush=True)
            ct = orc.contacts()
            orc.collide(a.body_q, ct)
            orc.xpbd_step(a, b, orc.control(), ct if ct.count[0] else None, dt, iterations=iters)
            a, b = b, a
    np.savez_compressed(os.path.join(HERE, "match_reference_vectors.npz"), **blob)
    print("wrote", len(blob), "arrays")


if __name__ == "__main__":
    main()
