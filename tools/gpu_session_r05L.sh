#!/bin/bash
# Round-5 GPU session L: support map with the hull scan / the curved primitives as calls (variants) against the product (both inlined,
# barrel as a call) on the two convex lines and on config C5's geometry; device tests of the collide paths on the product.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05L}
line() { python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)"; }
run() {  # lib workload steps warmup
  if [ "$1" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$1 bench.py"; fi
  echo -n "$2 $1 " >> $O/${T}_ab.txt
  ( timeout 400 $cmd --no-cpu-baseline --workload $2 --steps $3 --warmup $4 2>&1 | grep -v amdgpu.ids | tail -1 ) | line >> $O/${T}_ab.txt
}
for lib in product libhullcall.so liballcall.so libbase.so product; do run $lib quadruped_convex 150 10; done
for lib in product libhullcall.so liballcall.so libbase.so; do run $lib box_stack 100 5; done
for lib in product libhullcall.so liballcall.so; do run $lib hull_bin 10 2; done
( timeout 300 python -m pytest tests/test_zz_pair_heavy_gpu.py tests/test_gpu_parity_convex.py tests/test_zx_round2_gpu.py -m gpu -q -x -k "pair_heavy or convex or hull or barrel or reference_collision" 2>&1 | tail -4 ) > $O/${T}_gputests.log
echo done > $O/${T}_done
