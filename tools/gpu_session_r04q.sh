#!/bin/bash
# Round-4 GPU session Q: SolverFeatherstone tree-structured mass matrix, third version (v2 = bit masks + entry list, factor and L^-T
# fused per level, last substitution on wave 0; v3 = lane-owned entries decoded once per step, paired updates, unrolled bit loops): device tests, A/B against the dense order, per-phase cycles of both.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04q
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 600 python -m pytest tests/test_gpu_parity_featherstone.py tests/test_gpu_full_size.py tests/test_zx_round2_gpu.py tests/test_gpu_parity_joint_zoo.py -m gpu -q -s -k "feather or c3 or Feather or zoo" 2>&1 | grep -E "c3 live contacts.*step': 9|passed|failed|Error" | tail -15 ) > $O/${T}_tests.log
for mm in tree dense; do
  b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_featherstone --fs-mass-matrix $mm --steps 100 --warmup 5 > $O/${T}_bench_featherstone_$mm.json
done
b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_featherstone --envs-per-gpu 65536 --steps 20 --warmup 3 > $O/${T}_bench_featherstone_tree_65536.json
for mm in tree; do
  ( VARIANT_LIB=$R/variants/libv_timing.so timeout 300 python tools/phase_timing.py featherstone $mm 2>&1 | grep -v amdgpu.ids | tail -14 ) > $O/${T}_phase_timing_featherstone_$mm.txt
done
echo done > $O/${T}_done
