#!/bin/bash
# Round-5 GPU session B: the body phases on linear / angular lanes (world COM as the linear state inside a step).
#  1. launch-shape A/B of the round-4 kernels at 4096 envs (is the uniform-parameter tile of 16 as fast as the per-environment one?)
#  2. per-phase cycles of the new kernels (NT_PHASE_TIMING variant)
#  3. product build: XPBD parity tests, headline, saturated size
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r05b
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
for s in 16,512,1,0 16,512,1,1 16,256,1,0 8,256,2,0; do
  echo -n "base shape=$s " >> $O/${T}_shape_ab.txt
  NT_XPBD_CFG=$s b timeout 300 python tools/with_lib.py variants/libnewton_base_allshapes.so bench.py --no-cpu-baseline --steps 300 --warmup 20 | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M env-steps/s', round(d['roofline']['kernel_ms'],4),'ms valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_shape_ab.txt
done
( VARIANT_LIB=variants/libnewton_timing.so timeout 300 python tools/phase_timing.py 2>&1 | tail -30 ) > $O/${T}_phase_timing_quadruped.txt
( timeout 900 python -m pytest tests/test_gpu_parity_xpbd.py tests/test_gpu_full_size.py tests/test_zx_round2_gpu.py tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -15 ) > $O/${T}_gputests.log
b timeout 400 python bench.py --no-cpu-baseline > $O/${T}_bench_default.json
b timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/${T}_bench_driver_shape.json
b timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 100 --warmup 10 > $O/${T}_bench_65536.json
echo done > $O/${T}_done
