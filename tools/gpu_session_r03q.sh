#!/bin/bash
# Round-3 GPU session Q: hydroelastic reduction walking the pair's faces by rank.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests -m gpu -q -k "hydro" 2>&1 | tail -6 ) > $O/r03q_gputests_hydro.log
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 5 --warmup 2 > $O/r03q_bench_hydro_bin.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03q_prof -o h --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/r03q_prof.log 2>&1
f=$(find $O/r03q_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > $O/r03q_kernel_stats_hydro_bin_256.csv
rm -rf $O/r03q_prof
echo done > $O/r03q_done
