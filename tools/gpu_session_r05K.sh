#!/bin/bash
# Round-5 GPU session K: same-box A/B -- barrel support map inlined (product of this commit) vs as a call (libnoinl.so) on the convex
# lines; hull vertices in LDS vs global (libnolds.so) on config C5's geometry; the Featherstone rollout with its merged barrier
# intervals against variants/libbase.so.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05K}
line() { python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)"; }
run() {  # lib workload steps warmup
  if [ "$1" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$1 bench.py"; fi
  echo -n "$2 $1 " >> $O/${T}_ab.txt
  ( timeout 400 $cmd --no-cpu-baseline --workload $2 --steps $3 --warmup $4 2>&1 | grep -v amdgpu.ids | tail -1 ) | line >> $O/${T}_ab.txt
}
for lib in libbase.so product libnoinl.so libbase.so; do run $lib quadruped_convex 150 10; done
for lib in libbase.so libnoinl.so; do run $lib box_stack 100 5; done
for lib in libbase.so libnolds.so product libnoinl.so; do run $lib hull_bin 10 2; done
for lib in libbase.so product product libbase.so; do run $lib quadruped_featherstone 100 10; done
( timeout 300 python -m pytest tests/test_gpu_parity_featherstone.py tests/test_gpu_full_size.py -m gpu -q -x -k "not c4 and not c2 and not c5" 2>&1 | tail -4 ) > $O/${T}_gputests.log
echo done > $O/${T}_done
