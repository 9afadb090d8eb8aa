#!/usr/bin/env python
"""Issue / stall breakdown of the dominant kernel from one rocprofv3 SQ counter pass (run ON the GPU box).

SQ has 8 counter slots per pass on gfx950 (MI355X_MICROARCH.md "rocprofv3 PMC slots").  WAIT_ANY (wave parked on
s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES, all in quad-cycles summed over waves.
Writes gpurun_out/pmc_sq_<workload>.json.

usage:  python tools/pmc_sq.py [workload[@envs]] [COUNTER,COUNTER,...|set2]
        set2 = the instruction-mix set (scalar / branch / memory instruction counts), filtered by what `rocprofv3 -L` offers
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
            "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS"]
SET2 = ["SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
        "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_MISC", "SQ_WAVES", "SQ_BUSY_CYCLES"]


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "quadruped"
    workload, _, envs = spec.partition("@")
    counters = sys.argv[2].split(",") if len(sys.argv) > 2 else COUNTERS
    tag = spec.replace("@", "_")
    if counters == ["set2"]:
        listing = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120, cwd="/tmp",
                                 env=dict(os.environ, TMPDIR="/tmp")).stdout
        counters = [c for c in SET2 if c in listing][:8]
        tag += "_set2"
    os.makedirs(OUT, exist_ok=True)
    d = os.path.join(OUT, f"pmc_sq_{tag}")
    env = dict(os.environ, TMPDIR="/tmp")
    log = open(os.path.join(OUT, f"pmc_sq_{tag}.log"), "w")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                    sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "10", "--warmup", "10",
                    "--no-cpu-baseline", *(["--envs-per-gpu", envs] if envs else [])], check=True, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=240)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rollout" not in k:
            continue
        k = k[k.index("rollout") - 5:].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    res = {}
    for k, c in acc.items():
        n = max(len(launches[k]), 1)
        per = {name: v / n for name, v in c.items()}
        wc = per.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            per["frac_wait_any"] = per.get("SQ_WAIT_ANY", 0.0) / wc
            per["frac_wait_inst_any"] = per.get("SQ_WAIT_INST_ANY", 0.0) / wc
            per["frac_active_any"] = per.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            per["frac_active_valu"] = per.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
        res[k] = {"launches": n, "per_launch": per}
    # the build the counters belong to (bench.py attaches them only to this build, or to one with a byte-identical stepping unit)
    sys.path.insert(0, ROOT)
    import ctypes

    import __graft_entry__ as g

    lib = ctypes.CDLL(g.LIB)
    lib.nt_build_info.restype = ctypes.c_char_p
    res["_build"] = {"build_id": lib.nt_build_info().decode(), "step_unit": g.step_unit_id(workload), "workload": workload, "envs": envs or "default"}
    json.dump(res, open(os.path.join(OUT, f"pmc_sq_{tag}.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
