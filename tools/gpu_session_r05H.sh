#!/bin/bash
# Round-5 GPU session H: the shipped stepping unit after the last two kernel edits (joint rows fetch each body's W tile once; lane split a
# compile-time fact per kernel): stepping-kernel device tests, smoke, headline (+ cpu_baseline, driver shape), kernel stats, PMC traffic,
# SQ passes, env sweep, the two convex lines.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05H}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests/test_gpu_parity_xpbd.py tests/test_gpu_full_size.py tests/test_zx_round2_gpu.py tests/test_zy_recent_gpu.py tests/test_gpu_graph.py tests/test_gpu_parity_convex.py tests/test_gpu_parity_joint_zoo.py tests/test_gpu_viewer_recorder.py tests/test_zz_pair_heavy_gpu.py -m gpu -q 2>&1 | tail -6 ) > $O/${T}_gputests_stepping.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -4 ) > $O/${T}_pmc_traffic.log
b timeout 400 python bench.py > $O/${T}_bench_default.json
b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
( timeout 400 python tools/pmc_sq.py quadruped@4096 2>&1 | tail -3 ) > $O/${T}_pmc_sq_4096.log; cp $O/pmc_sq_quadruped_4096.json $O/${T}_pmc_sq_quadruped_4096.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
cd $R
b timeout 600 python bench.py --no-cpu-baseline --sweep 4096,8192,65536 --sweep-out $O/${T}_env_sweep.json > /dev/null
for e in quadruped_convex:150 box_stack:100; do
  IFS=: read w steps <<< "$e"
  b timeout 600 python bench.py --no-cpu-baseline --workload $w --steps $steps --warmup 5 > $O/${T}_bench_$w.json
done
rm -rf $O/pmc_quadruped_* $O/pmc_sq_quadruped_*/
echo done > $O/${T}_done
