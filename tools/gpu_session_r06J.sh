#!/bin/bash
# Round-6 GPU session J: SolverFeatherstone variants -- per-phase cycles + bench A/B (product vs variants).  usage: TAG "timing libs" "bench libs"
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1
for l in $2; do
  echo "== $l" >> $O/${T}_phase_timing_featherstone.txt
  ( VARIANT_LIB=variants/$l timeout 600 python tools/phase_timing.py featherstone 2>&1 | tail -13 ) >> $O/${T}_phase_timing_featherstone.txt
done
for rep in 1 2; do for lib in $3; do
  if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
  echo -n "quadruped_featherstone $lib " >> $O/${T}_ab_featherstone.txt
  ( timeout 600 $cmd --no-cpu-baseline --workload quadruped_featherstone --steps 200 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab_featherstone.txt
done; done
echo done > $O/${T}_done
