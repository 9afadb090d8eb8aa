#!/bin/bash
# Round-3 GPU session E: SDF bin after the reduction rewrite; headline A/B of the fast-math variant (v_rcp / v_sqrt in the XPBD
# projection phases) incl. the XPBD parity tests under the variant; per-phase cycles of the headline; RCCL single-rank test.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03e_bench_sdf_bin.json
( timeout 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 200 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03e_bench_default_a.json
( NEWTON_HIP_LIB=$R/build_ab/libnewton_fastmath.so timeout 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 200 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03e_bench_fastmath.json
( timeout 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 200 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03e_bench_default_b.json
( NEWTON_HIP_LIB=$R/build_ab/libnewton_fastmath.so timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/r03e_gputests_fastmath.log
( timeout 300 python -m pytest tests/test_gpu_nccl_single.py -m gpu -q 2>&1 | tail -8 ) > $O/r03e_nccl_single.log
( NEWTON_HIP_LIB=$R/build_ab/libnewton_timing2.so timeout 300 python tools/phase_timing.py 2>&1 | grep -v amdgpu.ids | tail -14 ) > $O/r03e_phase_timing_quadruped.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03e_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 --settle-frames 40 > $O/r03e_prof.log 2>&1
f=$(find $O/r03e_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -20 "$f" > $O/r03e_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03e_prof
echo done > $O/r03e_done
