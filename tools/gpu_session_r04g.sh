#!/bin/bash
# Round-4 GPU session G: register-cached incidence lists in the body phases (headline A/B against session D), where the staged
# hydroelastic reduce stage spends its cycles (variants/libv_hytime.so), GPU suite.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04g
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default.json
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default_again.json
b timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/${T}_bench_65536.json
( timeout 600 python tools/with_lib.py $R/variants/libv_hytime.so tools/hydro_timing.py 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/${T}_hydro_timing.json
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/${T}_gputests.log
echo done > $O/${T}_done
