#!/usr/bin/env python
"""Measure the HBM traffic of the dominant kernel of a bench.py workload with rocprofv3 PMC counters (run ON the GPU box).

Two separate passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
each over the same command: a known-byte-count calibration copy (4 B/lane coalesced, our access pattern) followed by
bench.py rollouts.  Counters are in KiB; the calibration ratio (known bytes / reported bytes) corrects the gfx950
FETCH_SIZE under-count.  Each record carries the library's source hash (nt_build_info), so bench.py only trusts a record
taken on the build it is running.  Appends to gpurun_out/r06_pmc_traffic.json (NT_PMC_OUT overrides the name; copy it to profiles/ to commit it).

usage (from the repo root on the GPU box):  python tools/pmc_traffic.py [workload[@envs] ...]      default: quadruped@4096
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
CAL_FLOATS = 64 * 1024 * 1024  # 256 MiB read + 256 MiB written: past the 256 MiB Infinity Cache
KERNEL = {"quadruped": "xpbd_rollout", "quadruped_convex": "xpbd_rollout", "box_stack": "xpbd_rollout", "hull_bin": "xpbd_rollout",
          "quadruped_featherstone": "featherstone_rollout",
          # the staged legs of collide(): one launch of the named kernel per substep (the 10 averaged launches = the last timed frame)
          "hydro_bin": "hydro_stage_faces", "sdf_bin": "sdf_reduce", "terrain": "mesh_triangle_pairs", "mesh_ground": "mesh_plane_pairs"}

WORKLOAD = r'''
import sys, ctypes as C
sys.path.insert(0, "%(root)s"); sys.path.insert(0, "%(root)s/tests")
import torch
from newton_amd import _lib
lib = _lib.load()
a = torch.rand(%(n)d, device="cuda"); b = torch.empty_like(a)
for _ in range(3):
    lib.nt_calibration_copy(a.data_ptr(), b.data_ptr(), %(n)d, C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
del a, b
sys.argv = ["bench.py", "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--workload", "%(workload)s",
            "--envs-per-gpu", "%(envs)d"]
import runpy
runpy.run_path("%(root)s/bench.py", run_name="__main__")
'''


def run_pass(counter, workload, envs):
    tag = f"{workload}_{envs}_{counter.lower()}"
    d = os.path.join(OUT, f"pmc_{tag}")
    env = dict(os.environ, TMPDIR="/tmp")
    code = WORKLOAD % {"root": ROOT, "n": CAL_FLOATS, "workload": workload, "envs": envs}
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                    sys.executable, "-c", code], check=True, cwd="/tmp", env=env,
                   stdout=open(os.path.join(OUT, f"pmc_{tag}.log"), "w"), stderr=subprocess.STDOUT)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    cal, roll = [], []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        if "calibration_copy" in r["Kernel_Name"]:
            cal.append(float(r["Counter_Value"]))
        elif KERNEL[workload] in r["Kernel_Name"]:
            roll.append(float(r["Counter_Value"]))
    return cal, roll


def build_id():
    sys.path.insert(0, ROOT)
    from newton_amd import _lib

    return _lib.load().nt_build_info().decode()


def main():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, os.environ.get("NT_PMC_OUT", "r06_pmc_traffic.json"))
    res = json.load(open(path)) if os.path.exists(path) else {}
    known = CAL_FLOATS * 4
    for spec in sys.argv[1:] or ["quadruped@4096"]:
        workload, _, envs = spec.partition("@")
        envs = int(envs or 4096)
        import __graft_entry__ as g  # noqa: PLC0415  (ROOT is on sys.path after build_id())

        rec = {"build_id": build_id(), "step_unit": g.step_unit_id(workload), "workload": workload, "envs": envs, "kernel": KERNEL[workload],
               "substeps_per_launch": 1 if workload in ("hydro_bin", "sdf_bin") else 10}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cal, roll = run_pass(counter, workload, envs)
            cal_kib = sum(cal[-2:]) / 2          # steady-state calibration launches
            roll_kib = sum(roll[-10:]) / 10      # the 10 timed steady-state rollout launches
            factor = known / (cal_kib * 1024.0)
            rec[counter] = {"calibration_reported_bytes": cal_kib * 1024, "calibration_known_bytes": known,
                            "correction_factor": factor, "reported_bytes": roll_kib * 1024,
                            "corrected_bytes": roll_kib * 1024 * factor}
        rec["bytes_per_launch"] = rec["FETCH_SIZE"]["corrected_bytes"] + rec["WRITE_SIZE"]["corrected_bytes"]
        rec["note"] = ("HBM bytes per launch of the workload's rollout kernel, FETCH_SIZE + WRITE_SIZE in separate rocprofv3 "
                       "--pmc passes, corrected by a known-byte 4 B/lane coalesced copy in the same pass")
        res[f"{workload}@{envs}"] = rec
        json.dump(res, open(path, "w"), indent=1)
        print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
