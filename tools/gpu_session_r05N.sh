#!/bin/bash
# Round-5 GPU session N (two calls: N1 = "suite", N2 = "lines"): the shipped build -- full device suite, smoke, headline (+ cpu_baseline,
# driver shape), rocprofv3 kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ pass on this build id; then the env sweep and
# every secondary bench line.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${2:-r05N}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
if [ "${1:-suite}" = suite ]; then
  ( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/${T}_gputests.log
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
  ( timeout 600 python tools/pmc_traffic.py quadruped@4096 2>&1 | tail -20 ) > $O/${T}_pmc_traffic.log
  b timeout 400 python bench.py > $O/${T}_bench_default.json
  b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
  ( timeout 400 python tools/pmc_sq.py quadruped@4096 2>&1 | tail -3 ) > $O/${T}_pmc_sq_4096.log; cp $O/pmc_sq_quadruped_4096.json $O/${T}_pmc_sq_quadruped_4096.json 2>/dev/null
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
  f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
  cd $R
  rm -rf $O/pmc_quadruped_* $O/pmc_sq_quadruped_*/
  echo done > $O/${T}_suite_done
else
  for e in hull_bin:20 quadruped_featherstone:100 quadruped_convex:150 box_stack:100 quadruped_api:150 sdf_bin:20 hydro_bin:4 mesh_ground:20; do
    IFS=: read w steps <<< "$e"
    b timeout 600 python bench.py --no-cpu-baseline --workload $w --steps $steps --warmup 5 > $O/${T}_bench_$w.json
  done
  b timeout 600 python bench.py --no-cpu-baseline --sweep 4096,8192,65536 --sweep-out $O/${T}_env_sweep.json > /dev/null
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_h -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hull_bin --steps 10 --warmup 2 > $O/${T}_prof_h.log 2>&1
  f=$(find $O/${T}_prof_h -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" > $O/${T}_kernel_stats_hull_bin.csv; rm -rf $O/${T}_prof_h
  cd $R
  echo done > $O/${T}_lines_done
fi
