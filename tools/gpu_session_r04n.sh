#!/bin/bash
# Round-4 GPU session N: convex tile A/B — (16 envs, 256 threads, 1 workgroup per CU: 512-register budget, no spills) against the
# shipped (16, 512, 1) / (8, 256, 2) shapes, on the all-shapes variant library.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
for w in box_stack quadruped_convex; do
  b timeout 200 python tools/with_lib.py variants/libv_shapes.so bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 5 > $O/r04n_${w}_default.json
  for cfg in 16,256,1,1,1 16,256,2,1,1 16,512,1,1,1 8,256,2,1,1; do
    NT_XPBD_CFG=$cfg b timeout 200 python tools/with_lib.py variants/libv_shapes.so bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 5 > $O/r04n_${w}_${cfg//,/_}.json
  done
done
echo done > $O/r04n_done
