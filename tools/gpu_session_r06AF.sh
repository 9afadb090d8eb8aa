#!/bin/bash
# Round-6 GPU session AF: the heightfield workload (bench.py --workload terrain_hfield) beside the mesh terrain, kernel averages.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r06AF}
rm -f $O/${T}_ab.txt
for rep in 1 2; do
  for wl in terrain terrain_hfield; do
    timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tee $O/${T}_bench_$wl.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', round(d['value'] / 1e6, 4), 'M env-steps/s', round(d['ms_per_step'], 3), 'ms/step valid', d.get('valid_state'))" >> $O/${T}_ab.txt
  done
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload terrain_hfield --steps 3 --warmup 1 > $O/${T}_prof.log 2>&1
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -d, -f1-4 | cut -c1-170 > $O/${T}_kernel_stats_terrain_hfield.csv; rm -rf $O/${T}_prof
cd $R
cat $O/${T}_ab.txt; head -3 $O/${T}_kernel_stats_terrain_hfield.csv
