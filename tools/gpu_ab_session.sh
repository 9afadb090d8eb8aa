#!/bin/bash
# Same-box A/B of library variants on bench workloads (the shape of this round's sessions I-M; results: profiles/r05[I-M]_ab.txt).
# usage: tools/gpu_ab_session.sh TAG "workload:steps:warmup ..." lib [lib ...]     lib = product | a file under variants/
# Each workload runs every library in the order given (repeat a name for ABBA).  Variants are built off the box with
# tools/build_variant.py; the previous commit's library is kept as variants/libbase.so by copying newton_amd/libnewton_hip.so before a rebuild.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; shift
W=$1; shift
for e in $W; do
  IFS=: read w steps warm <<< "$e"
  for lib in "$@"; do
    if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
    echo -n "$w $lib " >> $O/${T}_ab.txt
    ( timeout 400 $cmd --no-cpu-baseline --workload $w --steps $steps --warmup ${warm:-5} 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab.txt
  done
done
echo done > $O/${T}_done
