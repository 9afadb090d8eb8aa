#!/bin/bash
# Round-3 GPU session C: the restructured SDF narrow phase (cull -> compaction -> dense search, records in LDS, OBB pre-test):
# bench line + rocprofv3 kernel stats, and the register-cap variants (3 / 4 waves per SIMD) through NEWTON_HIP_LIB.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03c_bench_sdf_bin.json
for n in 3 4; do
  ( NEWTON_HIP_LIB=$R/build_ab/libnewton_sdfw$n.so timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03c_bench_sdf_bin_w$n.json
done
( timeout 900 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_sdf_reference_vectors.py tests/test_gpu_sdf.py -m gpu -x -q 2>&1 | tail -5 ) > $O/r03c_gputests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03c_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 --settle-frames 40 > $O/r03c_prof.log 2>&1
f=$(find $O/r03c_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" > $O/r03c_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03c_prof
echo done > $O/r03c_done
