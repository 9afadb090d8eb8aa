#!/bin/bash
# Round-6 GPU session Z4: the triangle leg with block bounds -- device tests, same-box A/B of bench.py --workload terrain with the
# plain scan (NT_TRIANGLE_BLOCKS=0) and the block pass, kernel averages of both.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r06Z4}
( timeout 600 python -m pytest tests/test_mesh_triangle.py tests/test_gpu_mesh_triangle_pipeline.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) > $O/${T}_gputests.log
for rep in 1 2; do
  for blocks in 0 1; do
    NT_TRIANGLE_BLOCKS=$blocks timeout 600 python bench.py --workload terrain --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('terrain blocks=$blocks', round(d['value'] / 1e6, 4), 'M env-steps/s', round(d['ms_per_step'], 3), 'ms/step valid', d.get('valid_state'))" >> $O/${T}_ab.txt
  done
done
cd /tmp
for blocks in 0 1; do
  NT_TRIANGLE_BLOCKS=$blocks timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof$blocks -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload terrain --steps 3 --warmup 1 > $O/${T}_prof_terrain_blocks$blocks.log 2>&1
  f=$(find $O/${T}_prof$blocks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -d, -f1-4 | cut -c1-170 > $O/${T}_kernel_stats_terrain_blocks$blocks.csv; rm -rf $O/${T}_prof$blocks
done
cd $R
NT_TRIANGLE_BLOCKS=1 timeout 600 python bench.py --workload terrain --no-cpu-baseline --steps 10 --warmup 3 > $O/${T}_bench_terrain.json 2>/dev/null
cat $O/${T}_gputests.log $O/${T}_ab.txt; head -3 $O/${T}_kernel_stats_terrain_blocks0.csv; head -3 $O/${T}_kernel_stats_terrain_blocks1.csv
echo done > $O/${T}_done
