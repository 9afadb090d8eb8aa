#!/bin/bash
# Round-4 GPU session Z (shipped build): GPU suite, smoke, PMC traffic + SQ passes at 4 096 / 65 536 envs, headline line with the
# cpu baseline, env sweep, kernel stats of the headline and of the SDF workloads, every secondary workload.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r04z}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -20 ) > $O/${T}_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
rm -f $O/r04_pmc_traffic.json
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -40 ) > $O/${T}_pmc_traffic.log
( timeout 300 python tools/pmc_sq.py quadruped 2>&1 | tail -20 ) > $O/${T}_pmc_sq.log
( timeout 300 python tools/pmc_sq.py quadruped@65536 2>&1 | tail -20 ) > $O/${T}_pmc_sq_65536.log
cp $O/pmc_sq_quadruped.json $O/${T}_pmc_sq_quadruped_4096.json 2>/dev/null
cp $O/pmc_sq_quadruped_65536.json $O/${T}_pmc_sq_quadruped_65536.json 2>/dev/null
rm -rf $O/pmc_sq_*/ $O/pmc_quadruped_*/ 2>/dev/null
cp $O/r04_pmc_traffic.json $R/profiles/r04_pmc_traffic.json 2>/dev/null
b timeout 400 python bench.py > $O/${T}_bench_default.json
b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
( timeout 600 python bench.py --no-cpu-baseline --sweep 4096,8192,65536,262144 --sweep-out $O/${T}_env_sweep.json 2>&1 | grep -v amdgpu.ids | tail -4 ) > $O/${T}_env_sweep.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
for w in sdf_bin hydro_bin quadruped_api; do
  st=3; [ $w = sdf_bin ] && st=10; [ $w = quadruped_api ] && st=100
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_$w -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload $w --steps $st --warmup 2 > $O/${T}_bench_$w.json 2>$O/${T}_prof_$w.log
  f=$(find $O/${T}_prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -14 "$f" > $O/${T}_kernel_stats_$w.csv
  rm -rf $O/${T}_prof_$w
done
cd $R
for w in quadruped_convex box_stack quadruped_featherstone hull_bin; do
  st=100; [ $w = hull_bin ] && st=5
  b timeout 400 python bench.py --no-cpu-baseline --workload $w --steps $st --warmup 5 > $O/${T}_bench_$w.json
done
echo done > $O/${T}_done
