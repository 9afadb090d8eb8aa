#!/bin/bash
# Round-5 GPU session J: the pair-heavy tile with the hull vertices staged in LDS and contiguous (environment, slot) correction records
# (nt_contacts.cw), barrel cylinders (device tests against the executed reference), same-box A/B against variants/libbase.so on config
# C5's geometry and on the two convex lines (the barrel branch sits in their support map), per-phase cycles of the pair-heavy tile.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05J}
line() { python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)"; }
run() {  # lib workload steps warmup
  if [ "$1" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$1 bench.py"; fi
  echo -n "$2 $1 " >> $O/${T}_ab.txt
  ( timeout 400 $cmd --no-cpu-baseline --workload $2 --steps $3 --warmup $4 2>&1 | grep -v amdgpu.ids | tail -1 ) | line >> $O/${T}_ab.txt
}
( timeout 500 python -m pytest tests/test_zz_pair_heavy_gpu.py tests/test_gpu_parity_convex.py tests/test_gpu_convex_known_answers.py tests/test_zx_round2_gpu.py -m gpu -q -x -k "pair_heavy or convex or hull or barrel or reference_collision or known" 2>&1 | tail -6 ) > $O/${T}_gputests.log
for lib in libbase.so product; do run $lib hull_bin 10 2; done
for lib in libbase.so product product libbase.so; do run $lib quadruped_convex 150 10; done
for lib in libbase.so product; do run $lib box_stack 100 5; done
if [ -f variants/libtiming.so ]; then
  ( VARIANT_LIB=variants/libtiming.so timeout 300 python tools/phase_timing.py hull_bin 512 2>&1 | tail -12 ) > $O/${T}_phase_timing_hull_bin.txt
fi
echo done > $O/${T}_done
