#!/bin/bash
# Round-4 GPU session I: per-call API kernels on the uniform-parameter tile (quadruped_api A/B against 36.6 M), GPU suite.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04i
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/${T}_gputests.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload quadruped_api --steps 100 --warmup 20 > $O/${T}_bench_quadruped_api.json 2>$O/${T}_prof.log
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -6 "$f" > $O/${T}_kernel_stats_quadruped_api.csv
rm -rf $O/${T}_prof
cd $R
b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_api --steps 100 --warmup 20 --graph > $O/${T}_bench_quadruped_api_graph.json
echo done > $O/${T}_done
