#!/bin/bash
# Round-6 GPU session Q: rocprofv3 kernel averages of hydro_bin (one timed frame) per library.  usage: tools/gpu_session_r06Q.sh TAG "lib ..."
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; T=$1
for l in $2; do
  cd /tmp
  if [ $l = product ]; then cmd="python $R/bench.py"; else cmd="python $R/tools/with_lib.py $R/variants/$l $R/bench.py"; fi
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_$l -o p --output-format csv -- $cmd --no-cpu-baseline --workload hydro_bin --steps 1 --warmup 1 > $O/${T}_prof_$l.log 2>&1
  cd $R
  f=$(find $O/${T}_prof_$l -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -d, -f1-4 | cut -c1-150 > $O/${T}_kernel_stats_hydro_$l.csv; rm -rf $O/${T}_prof_$l
  tail -1 $O/${T}_prof_$l.log | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$l', round(d['value']),'env-steps/s', round(d['ms_per_step'],1),'ms valid',d['valid_state'], 'faces', d['sdf_leg'].get('hydro_faces'), 'rows', d['sdf_leg'].get('rows'))
except Exception as e: print('$l FAILED', e)" >> $O/${T}_bench.txt
done
