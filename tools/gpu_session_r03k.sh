#!/bin/bash
# Round-3 GPU session K: the build that ships -- full GPU suite, smoke, headline bench (default + driver shape) with rocprofv3 kernel
# stats, PMC traffic and SQ passes (4 096 and 65 536 envs), secondary workloads, the SDF legs.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/r03k_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/r03k_smoke.log
b timeout 400 python bench.py > $O/r03k_bench_default.json
b timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r03k_bench_driver_shape.json
for w in quadruped_api quadruped_convex box_stack quadruped_featherstone hull_bin; do
  b timeout 300 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 20 > $O/r03k_bench_$w.json
done
b timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 > $O/r03k_bench_sdf_bin.json
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 5 --warmup 2 > $O/r03k_bench_hydro_bin.json
( timeout 400 python tools/pmc_traffic.py quadruped@4096 2>&1 | tail -30 ) > $O/r03k_pmc_traffic.log
( timeout 300 python tools/pmc_sq.py quadruped 2>&1 | tail -30 ) > $O/r03k_pmc_sq.log
( timeout 300 python tools/pmc_sq.py quadruped@65536 2>&1 | tail -30 ) > $O/r03k_pmc_sq_65536.log
rm -rf $O/pmc_sq_*/ $O/pmc_quadruped_*/ 2>/dev/null
cp $O/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json 2>/dev/null
b timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/r03k_bench_with_traffic.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r03k_prof_q -o q --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/r03k_prof_q.log 2>&1
f=$(find $O/r03k_prof_q -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -10 "$f" > $O/r03k_kernel_stats_quadruped.csv
rm -rf $O/r03k_prof_q
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03k_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 > $O/r03k_prof.log 2>&1
f=$(find $O/r03k_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" > $O/r03k_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03k_prof
echo done > $O/r03k_done
