#!/bin/bash
# Round-4 GPU session O: (1) the new row-matching tests + the Featherstone suite on the device, (2) SolverFeatherstone A/B: tree-structured
# mass matrix (default) against the reference's dense operation order on C3, (3) launch-shape A/B of the analytic XPBD rollout at
# 4 096 / 65 536 envs on the all-shapes variant library (same kernels as the shipped build).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04o
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 600 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_gpu_parity_featherstone.py tests/test_gpu_full_size.py tests/test_zx_round2_gpu.py -m gpu -q -k "contact_matching or feather or c3 or Feather" 2>&1 | tail -15 ) > $O/${T}_tests.log
for mm in tree dense; do
  b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_featherstone --fs-mass-matrix $mm --steps 100 --warmup 5 > $O/${T}_bench_featherstone_$mm.json
done
b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_featherstone --envs-per-gpu 65536 --steps 20 --warmup 3 > $O/${T}_bench_featherstone_tree_65536.json
for cfg in 16,256,2,1 16,512,2,1 16,512,4,1 32,512,1,1 8,128,4,1 8,256,4,1 16,256,1,0 16,512,1,0; do
  NT_XPBD_CFG=$cfg b timeout 200 python tools/with_lib.py variants/libv_shapes.so bench.py --no-cpu-baseline --steps 300 --warmup 30 > $O/${T}_shape_4096_${cfg//,/_}.json
done
for cfg in 32,512,1,1 16,512,2,1 16,512,4,1 8,256,4,1; do
  NT_XPBD_CFG=$cfg b timeout 200 python tools/with_lib.py variants/libv_shapes.so bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/${T}_shape_65536_${cfg//,/_}.json
done
echo done > $O/${T}_done
