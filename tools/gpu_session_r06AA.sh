#!/bin/bash
# Round-6 GPU session AA ("ship", after the triangle leg): every measurement of the build in the tree.  Parts (arg 2, default all): pmc | bench | prof | tests | secondary
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r06AA}; P=${2:-all}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
want() { [ "$P" = all ] || [ "$P" = "$1" ]; }
if want pmc; then
  ( NT_PMC_OUT=${T}_pmc_traffic.json timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped_featherstone@4096 2>&1 | tail -30 ) > $O/${T}_pmc_traffic.log
  ( timeout 400 python tools/pmc_sq.py quadruped@4096 2>&1 | tail -30 ) > $O/${T}_pmc_sq.log
  [ -f $O/pmc_sq_quadruped_4096.json ] && cp $O/pmc_sq_quadruped_4096.json $O/${T}_pmc_sq_quadruped_4096.json
  rm -rf $O/pmc_quadruped_* $O/pmc_sq_quadruped_4096 $O/pmc_quadruped_featherstone_*
fi
if want bench; then
  b timeout 600 python bench.py > $O/${T}_bench_default.json
  b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
  b timeout 900 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --sweep 4096,8192,65536 --sweep-out $O/${T}_env_sweep.json > $O/${T}_env_sweep.log
fi
if want prof; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
  f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_f -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload quadruped_featherstone --steps 400 --warmup 40 > $O/${T}_prof_f.log 2>&1
  f=$(find $O/${T}_prof_f -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_featherstone.csv; rm -rf $O/${T}_prof_f
  cd $R
fi
if want tests; then
  ( timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 ) > $O/${T}_gputests.log
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
fi
if want secondary; then
  for w in quadruped_convex:100 box_stack:100 quadruped_featherstone:100 quadruped_api:40 hull_bin:10 sdf_bin:6 mesh_ground:20 terrain:10 hydro_bin:2; do
    IFS=: read wl steps <<< "$w"
    b timeout 900 python bench.py --no-cpu-baseline --workload $wl --steps $steps --warmup 4 > $O/${T}_bench_$wl.json
  done
fi
echo done > $O/${T}_done_$P
