#!/bin/bash
# Round-3 GPU session B: config C5 through the SDF leg with the 5 mm contact gap: bench line + rocprofv3 kernel stats.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/r03b_bench_sdf_bin.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03b_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 --settle-frames 40 > $O/r03b_prof.log 2>&1
f=$(find $O/r03b_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" > $O/r03b_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03b_prof
echo done > $O/r03b_done
