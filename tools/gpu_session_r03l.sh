#!/bin/bash
# Round-3 GPU session L (shipped build): env-count sweep, heterogeneous worlds, rocprofv3 kernel stats of hydro_bin.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python bench.py --no-cpu-baseline --sweep 4096,8192,16384,65536,262144,1048576 2>&1 | grep -v amdgpu.ids | tail -8 ) > $O/r03l_env_sweep.json
( timeout 300 python tools/hetero_bench.py 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/r03l_hetero_bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03l_prof -o h --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/r03l_prof.log 2>&1
f=$(find $O/r03l_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > $O/r03l_kernel_stats_hydro_bin_256.csv
rm -rf $O/r03l_prof
echo done > $O/r03l_done
