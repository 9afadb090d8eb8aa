#!/bin/bash
# same-box ABAB of bench.py workloads under one environment variable:  gpu_ab_env.sh TAG VAR A B "workload:steps ..."
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
T=$1; VAR=$2; A=$3; B=$4; WL=$5
rm -f $O/${T}_ab.txt
for w in $WL; do
  IFS=: read wl steps <<< "$w"
  for rep in 1 2; do
    for v in $A $B; do
      env $VAR=$v timeout 900 python bench.py --workload $wl --no-cpu-baseline --steps $steps --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $VAR=$v', round(d['value'] / 1e6, 4), 'M env-steps/s', round(d['ms_per_step'], 3), 'ms/step valid', d.get('valid_state'))" >> $O/${T}_ab.txt
    done
  done
done
cat $O/${T}_ab.txt
