#!/bin/bash
# Round-5 GPU session M: Featherstone rollout with eval_rigid_tau split (wrench sweep + one lane per dof) against variants/libbase.so,
# per-phase cycles of the Featherstone rollout and of the pair-heavy tile on this build, device tests of the new paths.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05M}
line() { python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)"; }
run() {  # lib workload steps warmup
  if [ "$1" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$1 bench.py"; fi
  echo -n "$2 $1 " >> $O/${T}_ab.txt
  ( timeout 400 $cmd --no-cpu-baseline --workload $2 --steps $3 --warmup $4 2>&1 | grep -v amdgpu.ids | tail -1 ) | line >> $O/${T}_ab.txt
}
for lib in libbase.so product product libbase.so; do run $lib quadruped_featherstone 100 10; done
( VARIANT_LIB=variants/libtiming.so timeout 200 python tools/phase_timing.py featherstone tree 2>&1 | tail -14 ) > $O/${T}_phase_timing_featherstone.txt
( timeout 400 python -m pytest tests/test_gpu_parity_featherstone.py tests/test_gpu_full_size.py tests/test_zx_round2_gpu.py tests/test_gpu_sdf_pipeline.py -m gpu -q -x -k "(featherstone or barrel or c3) and not c5" 2>&1 | tail -4 ) > $O/${T}_gputests.log
for lib in $(ls variants | grep big); do run $lib hull_bin 10 2; done
run product hull_bin 10 2
echo done > $O/${T}_done
