#!/bin/bash
# Round-6 GPU session B: A/B of dev variants on the headline (4096 / 65536 envs) + per-phase cycles of timing variants.
# usage: tools/gpu_session_r06B.sh TAG "lib ..." "timing_lib ..."
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; LIBS=$2; TLIBS=${3:-}
run() {  # lib envs steps
  echo -n "$1 envs=$2 " >> $O/${T}_ab.txt
  ( timeout 400 python tools/with_lib.py variants/$1 bench.py --no-cpu-baseline --envs-per-gpu $2 --steps $3 --warmup 20 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M env-steps/s', round(d['ms_per_step'],4),'ms valid',d['valid_state'], d['roofline'].get('kernel',''))
except Exception as e: print('FAILED', e)" >> $O/${T}_ab.txt
}
for rep in 1 2; do for lib in $LIBS; do run $lib 4096 400; done; done
for lib in $LIBS; do run $lib 65536 60; done
for lib in $TLIBS; do
  echo "== $lib envs=4096" >> $O/${T}_phase_timing.txt
  ( NT_TIMING_ENVS=4096 VARIANT_LIB=variants/$lib timeout 300 python tools/phase_timing.py 2>&1 | tail -11 ) >> $O/${T}_phase_timing.txt
done
echo done > $O/${T}_done
