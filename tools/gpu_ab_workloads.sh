#!/bin/bash
# Same-box A/B of a variant library against the in-tree one on the secondary workloads.  usage: tools/gpu_ab_workloads.sh TAG variant.so "workload:steps ..."
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; LIB=$2; shift; shift
for e in $1; do
  IFS=: read w steps <<< "$e"
  for lib in product $LIB product $LIB; do
    echo -n "$w $lib " >> $O/${T}_ab_workloads.txt
    if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
    ( timeout 400 $cmd --no-cpu-baseline --workload $w --steps $steps --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'], d['roofline']['kernel'][:60])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab_workloads.txt
  done
done
echo done > $O/${T}_done
