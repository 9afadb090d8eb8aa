#!/bin/bash
# Round-6 GPU session Z: the triangle leg's bench line (bench.py --workload terrain) and the kernel averages of the same command.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r06Z3}
timeout 900 python bench.py --workload terrain --no-cpu-baseline --steps 5 --warmup 2 > $O/${T}_bench_terrain.json 2> $O/${T}_bench_terrain.err
tail -3 $O/${T}_bench_terrain.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload terrain --steps 3 --warmup 1 > $O/${T}_prof_terrain.log 2>&1
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -d, -f1-4 | cut -c1-170 > $O/${T}_kernel_stats_terrain.csv; rm -rf $O/${T}_prof
cd $R
cat $O/${T}_bench_terrain.json | cut -c1-600
cat $O/${T}_kernel_stats_terrain.csv
echo done > $O/${T}_done
