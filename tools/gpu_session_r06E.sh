#!/bin/bash
# Round-6 GPU session E: the device suite on the in-tree build + same-box A/B of whole-library builds on every fused workload.
# usage: tools/gpu_session_r06E.sh TAG "libA libB" [notests]
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; LIBS=$2
if [ "${3:-}" != notests ]; then
  ( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) > $O/${T}_gputests.log
fi
for w in quadruped:300 quadruped_convex:100 box_stack:100 quadruped_featherstone:100 quadruped_api:40 hull_bin:10; do
  IFS=: read wl steps <<< "$w"
  for lib in $LIBS; do
    if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
    echo -n "$wl $lib " >> $O/${T}_ab_workloads.txt
    ( timeout 400 $cmd --no-cpu-baseline --workload $wl --steps $steps --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab_workloads.txt
  done
done
echo done > $O/${T}_done
