#!/bin/bash
# Round-5 GPU session F: the shipped build -- full device suite, smoke, headline (+ cpu_baseline, driver shape), rocprofv3 kernel stats,
# PMC traffic (FETCH_SIZE / WRITE_SIZE passes) and SQ passes on this build id, env sweep, every secondary bench line.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05F}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/${T}_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -20 ) > $O/${T}_pmc_traffic.log
b timeout 400 python bench.py > $O/${T}_bench_default.json
b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
( timeout 400 python tools/pmc_sq.py quadruped@4096 2>&1 | tail -3 ) > $O/${T}_pmc_sq_4096.log; cp $O/pmc_sq_quadruped_4096.json $O/${T}_pmc_sq_quadruped_4096.json 2>/dev/null
( timeout 400 python tools/pmc_sq.py quadruped@65536 2>&1 | tail -3 ) > $O/${T}_pmc_sq_65536.log; cp $O/pmc_sq_quadruped_65536.json $O/${T}_pmc_sq_quadruped_65536.json 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
cd $R
b timeout 600 python bench.py --no-cpu-baseline --sweep 4096,8192,65536,262144 --sweep-out $O/${T}_env_sweep.json > /dev/null
for e in quadruped_convex:150 box_stack:100 quadruped_featherstone:100 quadruped_api:150 hull_bin:20 sdf_bin:20 mesh_ground:20 hydro_bin:4; do
  IFS=: read w steps <<< "$e"
  b timeout 600 python bench.py --no-cpu-baseline --workload $w --steps $steps --warmup 5 > $O/${T}_bench_$w.json
done
echo done > $O/${T}_done
