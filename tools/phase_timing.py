#!/usr/bin/env python
"""Per-phase cycle accounting of xpbd_rollout_kernel (run ON the GPU box).

Builds a throw-away debug library with -DNT_PHASE_TIMING (workgroup 0 / thread 0 accumulates cycle deltas at every
phase barrier), runs the bench workload through it (newton_amd._lib.LIB_PATH assigned before the first load) and prints the share of each phase.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dbg = os.environ.get("VARIANT_LIB") or os.path.join(ROOT, "build_ab", "libnewton_timing.so")  # prebuilt off the GPU box (tools/build_variant.py ... -DNT_PHASE_TIMING)
if not os.path.exists(dbg):
    dbg = "/tmp/libnewton_hip_timing.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-Os", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-fPIC", "-shared", "-DNT_PHASE_TIMING", os.path.join(ROOT, "newton_amd/csrc/nt_kernels.hip"),
                    os.path.join(ROOT, "newton_amd/csrc/nt_broadphase.hip"), "-o", dbg], check=True)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import newton_amd as nt  # noqa: E402
from newton_amd import _lib  # noqa: E402

_lib.LIB_PATH = dbg
for _name in ("nt_bandwidth_probe",):  # (a variant built from older sources may lack entry points added since)
    if _name.encode() not in open(dbg, "rb").read():
        _lib.SYMBOLS.pop(_name, None)
from scenes import quadruped_scene  # noqa: E402

lib = _lib.load()
BOX = len(sys.argv) > 1 and sys.argv[1] == "box_stack"
HULL = len(sys.argv) > 1 and sys.argv[1] == "hull_bin"  # config C5's geometry on the pair-heavy tile (bench.py --workload hull_bin)
if HULL:
    from scenes import hull_bin_scene  # noqa: E402

    model = hull_bin_scene(int(sys.argv[2]) if len(sys.argv) > 2 else 512, 64, seed=2, device="cuda:0")
elif BOX:
    from scenes import box_stack_scene  # noqa: E402

    model = box_stack_scene(int(sys.argv[2]) if len(sys.argv) > 2 else 256, device="cuda:0", seed=1)
else:
    NENV = int(os.environ.get("NT_TIMING_ENVS", "4096"))
    if len(sys.argv) > 1 and sys.argv[1] == "quadruped_convex":  # config C4's convex-convex variant (box links on a box slab)
        from scenes import quadruped_convex_scene  # noqa: E402

        model = quadruped_convex_scene(NENV, device="cuda:0", seed=1)
    else:
        model = quadruped_scene(NENV, device="cuda:0", seed=1)
    model.joint_q.reshape(NENV, -1)[:, 2] -= 0.24  # feet on the ground: the standing regime the bench measures
    model.body_q, model.body_qd = nt.articulation.eval_fk_numpy(model, model.joint_q, model.joint_qd)
s0, s1 = model.state(), model.state()
pipe = nt.CollisionPipeline(model)
contacts = pipe.contacts()
FS = len(sys.argv) > 1 and sys.argv[1] == "featherstone"
solver = nt.solvers.SolverFeatherstone(model, mass_matrix=(sys.argv[2] if len(sys.argv) > 2 else "tree")) if FS else nt.solvers.SolverXPBD(model, iterations=4 if BOX else 2)
DT = 1.0 / 1200.0 if HULL else 1e-3
for _ in range(30 if HULL else 100):
    solver.rollout(s0, s1, None, contacts, DT, 10)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
raw = C.CDLL(dbg)
clocks = raw.nt_debug_phase_clocks_fs if FS and hasattr(raw, "nt_debug_phase_clocks_fs") else raw.nt_debug_phase_clocks  # (per translation unit)
clocks(buf)
N = 10 if HULL else 50
for _ in range(N):
    solver.rollout(s0, s1, None, contacts, DT, 10)
torch.cuda.synchronize()
clocks(buf)
if FS:
    names = {10: "collide (+ previous tail)", 11: "FK", 12: "to internal qd", 13: "RNEA forward (pre + levels)", 14: "contacts + f_ext",
             15: "RNEA backward (tau)", 16: "P = I S (tree: I^c, I^c S)", 17: "H", 18: "factorise + solve", 19: "integrate", 20: "FK + velocities",
             21: "to public qd"}
    tot = sum(buf[i] for i in names)
    for i in sorted(names):
        print(f"{names[i]:32s} {buf[i] / N:12.0f} cycles/launch  {100.0 * buf[i] / tot:5.1f} %")
    print(f"{'total':32s} {tot / N:12.0f} cycles/launch")
    sys.exit(0)
names = {0: "prologue (load + derived)", 1: "shapes/AABB || joint forces", 2: "pair evaluation (1 lane / pair)",
         3: "contact records || live prefix", 4: "integrate",
         5: "contacts", 6: "apply (contacts)", 7: "joints", 8: "apply (joints)", 9: "epilogue (count + store)"}
tot = sum(buf[i] for i in range(10))
for i in range(10):
    print(f"{names[i]:32s} {buf[i] / N:12.0f} cycles/launch  {100.0 * buf[i] / tot:5.1f} %")
print(f"{'total':32s} {tot / N:12.0f} cycles/launch (s_memtime ticks of workgroup 0)")
