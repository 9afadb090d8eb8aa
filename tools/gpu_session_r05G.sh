#!/bin/bash
# Round-5 GPU session G: PMC traffic of the shipped stepping unit, the device suite with DeviceModel on the C descriptor builder
# (nt_model_create), and the steady-state kernel split of sdf_bin (settled state saved by one run, profiled run loads it: no settle frames
# in the kernel-stats averages).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05G}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -20 ) > $O/${T}_pmc_traffic.log
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/${T}_gputests.log
b timeout 400 python bench.py --no-cpu-baseline > $O/${T}_bench_default.json
b timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 2 --save-state /tmp/sdf_state.npz > $O/${T}_bench_sdf_bin.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_s -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 20 --warmup 2 --load-state /tmp/sdf_state.npz > $O/${T}_prof_s.log 2>&1
f=$(find $O/${T}_prof_s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" > $O/${T}_kernel_stats_sdf_bin_steady.csv; rm -rf $O/${T}_prof_s
tail -1 $O/${T}_prof_s.log > $O/${T}_bench_sdf_bin_profiled.json
echo done > $O/${T}_done
