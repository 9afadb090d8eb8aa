#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- bitwise regression check of a kernel refactor on the CPU: builds the emulated kernel library
(tests/emu/build.py) twice, from the kernel sources of a git revision (default HEAD) and from the working tree, runs the same
seeded scenes through both (collide, XPBD step / fused rollout in several launch shapes, SemiImplicit, Featherstone) and
compares every output array bit for bit.

usage: python tools/emu_bitcheck.py [git-rev]          exit code 0 = identical
"""
import ctypes as C
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def build_from(csrc_dir, out_dir, include_dir):
    import build

    build.CSRC, build.OUT = csrc_dir, out_dir
    build.LIB = os.path.join(out_dir, "libnewton_emu.so")
    if include_dir:
        build.ROOT_INCLUDE = include_dir
    return build.build(force=True)


def outputs(libpath):
    import harness as H

    H._emu = None
    lib = C.CDLL(libpath)
    from newton_amd import _lib as L

    for name, (restype, argtypes) in L.SYMBOLS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
    H._emu = lib
    import newton_amd as nt
    from scenes import box_stack_scene, hull_bin_scene, joint_zoo_scene, mixed_primitive_scene, pendulum_scene, quadruped_scene

    res = {}

    def lower(model, dz):
        E = model.world_count
        model.joint_q.reshape(E, -1)[:, 2] -= dz
        model.body_q, model.body_qd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)

    # quadruped XPBD: rollout in the default shape + every tile, restitution on/off
    for epb in (0, 8, 16, 1):
        m = quadruped_scene(19, seed=1)
        lower(m, 0.23)
        em = H.EmuModel(m)
        ct = H.EmuContacts(em)
        out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), H.EmuControl(em), ct, 1e-3, 6, epb=epb)
        res[f"quad_rollout_epb{epb}"] = [out.body_q.copy(), out.body_qd.copy(), ct.shape0.copy(), ct.data.copy(), ct.env_count.copy()]
    m = quadruped_scene(10, seed=2)
    lower(m, 0.24)
    em = H.EmuModel(m)
    a, b, ct = H.EmuState(em), H.EmuState(em), H.EmuContacts(em)
    H.collide(em, a, ct)
    H.xpbd_step(em, a, b, H.EmuControl(em), ct, 1e-3, enable_restitution=True)
    res["quad_step_restitution"] = [b.body_q.copy(), b.body_qd.copy()]
    for name, fn, n, kw in (("boxes", box_stack_scene, 5, dict(iterations=4)), ("mixed", mixed_primitive_scene, 6, {})):
        m = fn(n)
        em = H.EmuModel(m)
        ct = H.EmuContacts(em)
        out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), H.EmuControl(em), ct, 1.0 / 240.0, 4, **kw)
        res[name] = [out.body_q.copy(), out.body_qd.copy(), ct.data.copy()]
    # fused rollout with restitution; pair-heavy scene (contact records in HBM, one environment per workgroup)
    m = quadruped_scene(9, seed=5)
    lower(m, 0.24)
    em = H.EmuModel(m)
    ct = H.EmuContacts(em)
    out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), H.EmuControl(em), ct, 1e-3, 4, enable_restitution=True)
    res["quad_rollout_restitution"] = [out.body_q.copy(), out.body_qd.copy(), ct.env_count.copy()]
    m = hull_bin_scene(2, 40)
    em = H.EmuModel(m)
    ct = H.EmuContacts(em)
    out = H.xpbd_rollout(em, H.EmuState(em), H.EmuState(em), H.EmuControl(em), ct, 1.0 / 600.0, 3)
    res["hull_bin_rollout"] = [out.body_q.copy(), out.body_qd.copy(), ct.shape0.copy(), ct.data.copy(), ct.env_count.copy()]
    m = pendulum_scene(5, seed=4)
    em = H.EmuModel(m)
    a, b = H.EmuState(em), H.EmuState(em)
    for _ in range(3):
        H.semi_implicit_step(em, a, b, H.EmuControl(em), None, 1e-3)
        a, b = b, a
    res["semi_pendulum"] = [a.body_q.copy(), a.body_qd.copy()]
    m = mixed_primitive_scene(4)
    em = H.EmuModel(m)
    a, b, ct = H.EmuState(em), H.EmuState(em), H.EmuContacts(em)
    H.collide(em, a, ct)
    H.semi_implicit_step(em, a, b, H.EmuControl(em), ct, 1e-3)
    res["semi_contacts"] = [b.body_q.copy(), b.body_qd.copy()]
    m = quadruped_scene(6, seed=3)
    lower(m, 0.24)
    em = H.EmuModel(m)
    ct = H.EmuContacts(em)
    out = H.featherstone_rollout(em, H.EmuState(em), H.EmuState(em), H.EmuControl(em), ct, 1e-3, 3)
    res["fs_rollout"] = [out.body_q.copy(), out.joint_q.copy(), out.joint_qd.copy()]
    m = joint_zoo_scene(4)
    em = H.EmuModel(m)
    a, b = H.EmuState(em), H.EmuState(em)
    H.xpbd_step(em, a, b, H.EmuControl(em), None, 1e-3)
    res["zoo_xpbd"] = [b.body_q.copy(), b.body_qd.copy()]
    return res


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
    tmp = tempfile.mkdtemp(prefix="emu_bitcheck_")
    old_src = os.path.join(tmp, "old_csrc")
    os.makedirs(old_src)
    files = subprocess.run(["git", "ls-tree", "--name-only", rev, "newton_amd/csrc/"], cwd=ROOT, capture_output=True, text=True,
                           check=True).stdout.split()
    for f in files:
        blob = subprocess.run(["git", "show", f"{rev}:{f}"], cwd=ROOT, capture_output=True, check=True).stdout
        open(os.path.join(old_src, os.path.basename(f)), "wb").write(blob)
    code = ("import sys, pickle; sys.argv=['x']; sys.path.insert(0, %r); import emu_bitcheck as E; "
            "lib = E.build_from(%%r, %%r, None); pickle.dump(E.outputs(lib), open(%%r, 'wb'))" % os.path.join(ROOT, "tools"))
    import pickle

    outs = []
    for tag, src in (("old", old_src), ("new", os.path.join(ROOT, "newton_amd", "csrc"))):
        out_dir, pk = os.path.join(tmp, tag), os.path.join(tmp, tag + ".pkl")
        subprocess.run([sys.executable, "-c", code % (src, out_dir, pk)], check=True, cwd=ROOT)
        outs.append(pickle.load(open(pk, "rb")))
    bad = 0
    for k in outs[0]:
        same = all(np.array_equal(a, b) for a, b in zip(outs[0][k], outs[1][k]))
        print(f"{k:28s} {'identical' if same else 'DIFFERENT'}")
        bad += 0 if same else 1
    shutil.rmtree(tmp, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
