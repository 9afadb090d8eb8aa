#!/bin/bash
# Round-3 GPU session J: index-based reduction, occupancy variants of the cull / resolve kernels, hipGraph replay of the call-by-call frames.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -k "sdf or hydro or graph" 2>&1 | tail -15 ) > $O/r03j_gputests_sdf.log
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
b timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 > $O/r03j_bench_sdf_bin.json
for v in cull5 cull6 cull8 res4; do
  NEWTON_HIP_LIB=$R/build_ab/libnewton_$v.so b timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 6 --warmup 2 > $O/r03j_bench_sdf_bin_$v.json
done
b timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 --graph > $O/r03j_bench_sdf_bin_graph.json
b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_api --steps 200 --warmup 50 > $O/r03j_bench_quadruped_api.json
b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_api --steps 200 --warmup 50 --graph > $O/r03j_bench_quadruped_api_graph.json
b timeout 600 python tools/sdf_leg_stats.py 2048 40 > $O/r03j_sdf_leg_stats.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03j_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 > $O/r03j_prof.log 2>&1
f=$(find $O/r03j_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" > $O/r03j_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03j_prof
echo done > $O/r03j_done
