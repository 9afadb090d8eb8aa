#!/bin/bash
# Round-6 GPU session Z5: the triangle leg, one wave per pair (default) against 256 lanes per pair (NT_MESH_TRIANGLE_THREADS=256),
# both with the block pass: device tests in both launch shapes, same-box ABAB of bench.py --workload terrain, kernel averages.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r06Z5}
VAR=${2:-NT_MESH_TRIANGLE_THREADS}; A=${3:-256}; B=${4:-64}
( timeout 600 python -m pytest tests/test_mesh_triangle.py tests/test_gpu_mesh_triangle_pipeline.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) > $O/${T}_gputests.log
( env $VAR=$A timeout 600 python -m pytest tests/test_mesh_triangle.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) >> $O/${T}_gputests.log
rm -f $O/${T}_ab.txt
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v timeout 600 python bench.py --workload terrain --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('terrain $VAR=$v', round(d['value'] / 1e6, 4), 'M env-steps/s', round(d['ms_per_step'], 3), 'ms/step valid', d.get('valid_state'))" >> $O/${T}_ab.txt
  done
done
cd /tmp
for v in $A $B; do
  env $VAR=$v timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof$v -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload terrain --steps 3 --warmup 1 > $O/${T}_prof_terrain_$v.log 2>&1
  f=$(find $O/${T}_prof$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -d, -f1-4 | cut -c1-170 > $O/${T}_kernel_stats_terrain_$v.csv; rm -rf $O/${T}_prof$v
done
cd $R
timeout 600 python bench.py --workload terrain --no-cpu-baseline --steps 10 --warmup 3 > $O/${T}_bench_terrain.json 2>/dev/null
cat $O/${T}_gputests.log $O/${T}_ab.txt; head -3 $O/${T}_kernel_stats_terrain_$A.csv; head -3 $O/${T}_kernel_stats_terrain_$B.csv
echo done > $O/${T}_done
