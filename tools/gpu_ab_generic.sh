#!/bin/bash
# Same-box A/B of whole-library variants on bench workloads.  usage: tools/gpu_ab_generic.sh TAG "workload:steps ..." "lib lib ..." (product | file under variants/)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1
for e in $2; do
  IFS=: read w steps <<< "$e"
  for lib in $3; do
    if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
    echo -n "$w $lib " >> $O/${T}_ab.txt
    ( timeout 900 $cmd --no-cpu-baseline --workload $w --steps $steps --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab.txt
  done
done
echo done > $O/${T}_done
