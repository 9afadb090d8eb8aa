#!/bin/bash
# Round-4 GPU session D: live-contact table build (headline A/B against session B), per-phase cycles of the fused-namespace
# kernel (variants/libv_timing.so, built before the table), re-gated full-size tests, full suite.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04d
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default.json
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default_again.json
b timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/${T}_bench_65536.json
( VARIANT_LIB=$R/variants/libv_timing.so timeout 300 python tools/phase_timing.py 2>&1 | grep -v amdgpu.ids | tail -14 ) > $O/${T}_phase_timing_quadruped.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/${T}_gputests.log
echo done > $O/${T}_done
