#!/bin/bash
# Round-2 GPU session A (run through gpurun from the repo root): full GPU suite, headline bench (pre-settled), secondary
# bench lines, the BASELINE.md section 5 env-count sweep, rocprofv3 kernel stats + PMC traffic of the SHIPPED build.
# Everything lands in gpurun_out/r02a_*; summaries that are to be judged get copied into profiles/ afterwards.
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/parity_numbers.jsonl
python -c "import __graft_entry__ as g; g.build()" > $O/r02a_build.log 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/r02a_gputests.log
( timeout 300 python bench.py 2>&1 | tail -2 ) > $O/r02a_bench_default.log
( timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02a_bench_driver_shape.log
for w in quadruped_convex box_stack quadruped_featherstone; do
  ( timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02a_bench_$w.log
done
( timeout 600 python bench.py --workload hull_bin --envs-per-gpu 2048 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02a_bench_hull_bin_2048.log
( timeout 900 python bench.py --sweep 4096,16384,65536,262144,1048576 --steps 500 --warmup 20 --sweep-out $O/r02a_env_sweep.json 2>&1 | tail -8 ) > $O/r02a_sweep.log
# kernel stats of the shipped build, one rocprofv3 run per workload (kernel-trace + stats only)
cd /tmp
prof() { # name, bench args...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02a_prof_$n -o $n --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $O/r02a_prof_$n.log 2>&1
  f=$(find $O/r02a_prof_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f" > $O/r02a_kernel_stats_$n.csv
}
prof quadruped --steps 200 --warmup 50
prof quadruped_featherstone --workload quadruped_featherstone --steps 100 --warmup 10
prof box_stack --workload box_stack --steps 100 --warmup 10
prof quadruped_convex --workload quadruped_convex --steps 100 --warmup 10
prof hull_bin_2048 --workload hull_bin --envs-per-gpu 2048 --steps 10 --warmup 2
cd $R
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@262144 2>&1 | tail -40 ) > $O/r02a_pmc_traffic.log
( timeout 300 python tools/pmc_sq.py 2>&1 | tail -30 ) > $O/r02a_pmc_sq.log
echo done > $O/r02a_done
