#!/bin/bash
# Same-box A/B of variant libraries on one bench workload, ABBA order.  usage: tools/gpu_ab_libs.sh TAG "workload:steps ..." libA libB
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; shift
W=$1; shift
A=$1; B=$2
for e in $W; do
  IFS=: read w steps <<< "$e"
  for lib in $A $B $B $A; do
    echo -n "$w $lib " >> $O/${T}_ab_libs.txt
    if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
    ( timeout 400 $cmd --no-cpu-baseline --workload $w --steps $steps --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab_libs.txt
  done
done
echo done > $O/${T}_done
