#!/usr/bin/env python
"""Measure HBM traffic of xpbd_rollout_kernel with rocprofv3 PMC counters (run ON the GPU box).

Two separate passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
each over the same command: a known-byte-count calibration copy (4 B/lane coalesced, our access pattern) followed by
bench.py rollouts.  Counters are in KiB; the calibration ratio (known bytes / reported bytes) corrects the gfx950
FETCH_SIZE under-count.  Writes gpurun_out/pmc_traffic.json.

usage (from the repo root on the GPU box):  python tools/pmc_traffic.py
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
sys.argv = ["bench.py", "--steps", "10", "--warmup", "60", "--no-cpu-baseline"]
import runpy
runpy.run_path("%(root)s/bench.py", run_name="__main__")
''' % {"root": ROOT, "n": CAL_FLOATS}


def run_pass(counter):
    d = os.path.join(OUT, f"pmc_{counter.lower()}")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                    sys.executable, "-c", WORKLOAD], check=True, cwd="/tmp", env=env, stdout=open(os.path.join(OUT, f"pmc_{counter.lower()}.log"), "w"), stderr=subprocess.STDOUT)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    cal, roll = [], []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        if "calibration_copy" in r["Kernel_Name"]:
            cal.append(float(r["Counter_Value"]))
        elif "xpbd_rollout" in r["Kernel_Name"]:
            roll.append(float(r["Counter_Value"]))
    return cal, roll


def main():
    os.makedirs(OUT, exist_ok=True)
    res = {}
    known = CAL_FLOATS * 4
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        cal, roll = run_pass(counter)
        cal_kib = sum(cal[-2:]) / 2          # steady-state calibration launches
        roll_kib = sum(roll[-10:]) / 10      # the 10 timed steady-state rollout launches
        factor = known / (cal_kib * 1024.0)
        res[counter] = {"calibration_reported_bytes": cal_kib * 1024, "calibration_known_bytes": known,
                        "correction_factor": factor, "rollout_reported_bytes": roll_kib * 1024,
                        "rollout_corrected_bytes": roll_kib * 1024 * factor}
    res["xpbd_rollout_kernel_bytes_per_launch"] = (res["FETCH_SIZE"]["rollout_corrected_bytes"] +
                                                   res["WRITE_SIZE"]["rollout_corrected_bytes"])
    res["note"] = ("HBM bytes per xpbd_rollout_kernel launch (4096 envs x 10 substeps), FETCH_SIZE + WRITE_SIZE in separate "
                   "rocprofv3 --pmc passes, corrected by a known-byte 4 B/lane coalesced copy in the same pass")
    json.dump(res, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
