#!/bin/bash
# Round-4 GPU session H: SolverFeatherstone's phases with contraction (C3 A/B against session B: 26.65 M), hydroelastic reduce without
# the rebase pass + batched block-stage atomics, GPU suite.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04h
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/${T}_gputests.log
b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_featherstone --steps 100 --warmup 20 > $O/${T}_bench_quadruped_featherstone.json
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/${T}_bench_hydro_bin.json 2>$O/${T}_prof.log
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -10 "$f" > $O/${T}_kernel_stats_hydro_bin.csv
rm -rf $O/${T}_prof
echo done > $O/${T}_done
