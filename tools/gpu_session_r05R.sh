#!/bin/bash
# Round-5 GPU session R (two calls): "ab" = pair-heavy rollout with its contact records in nt_contacts.cr against the previous build
# (variants/libship.so) on config C5's geometry + the device tests of that tile; "ship" = the measurements of the shipped build
# (PMC traffic first, headline, kernel stats, stepping-kernel device tests, hull_bin line + its traffic).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${2:-r05R}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
if [ "${1:-ab}" = ab ]; then
  bash tools/gpu_ab_session.sh $T "hull_bin:10:2" libship.so product product libship.so
  ( timeout 300 python -m pytest tests/test_zz_pair_heavy_gpu.py tests/test_gpu_full_size.py -m gpu -q -x -k "pair_heavy or hull or c5" 2>&1 | tail -4 ) > $O/${T}_gputests.log
else
  ( timeout 400 python tools/pmc_traffic.py quadruped@4096 2>&1 | tail -20 ) > $O/${T}_pmc_traffic.log
  b timeout 400 python bench.py > $O/${T}_bench_default.json
  b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
  f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
  cd $R
  ( timeout 600 python -m pytest tests/test_gpu_parity_xpbd.py tests/test_gpu_full_size.py tests/test_zx_round2_gpu.py tests/test_zy_recent_gpu.py tests/test_gpu_graph.py tests/test_gpu_parity_convex.py tests/test_gpu_parity_joint_zoo.py tests/test_zz_pair_heavy_gpu.py tests/test_gpu_sdf_pipeline.py -m gpu -q 2>&1 | tail -6 ) > $O/${T}_gputests_stepping.log
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
  b timeout 400 python bench.py --no-cpu-baseline --workload hull_bin --steps 20 --warmup 5 > $O/${T}_bench_hull_bin.json
  ( timeout 300 python tools/pmc_traffic.py hull_bin@2048 2>&1 | tail -12 ) > $O/${T}_pmc_traffic_hull_bin.log
  rm -rf $O/pmc_quadruped_* $O/pmc_hull_bin_*
  echo done > $O/${T}_ship_done
fi
