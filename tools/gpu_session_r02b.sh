#!/bin/bash
# Round-2 GPU session B (run through gpurun from the repo root, on the build that ships): GPU suite, headline + secondary bench
# lines, env-count sweep, rocprofv3 kernel stats per workload, PMC traffic (FETCH / WRITE, separate passes) and the SQ passes.
# Everything lands in gpurun_out/r02b_*; the summaries to be judged are copied into profiles/ afterwards.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
rm -f $O/parity_numbers.jsonl
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/r02b_gputests.log
( timeout 400 python bench.py 2>&1 | tail -1 ) > $O/r02b_bench_default.json
( timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02b_bench_driver_shape.json
for w in quadruped_convex box_stack quadruped_featherstone; do
  ( timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02b_bench_$w.json
done
( timeout 600 python bench.py --workload hull_bin --envs-per-gpu 2048 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02b_bench_hull_bin_2048.json
( timeout 900 python bench.py --sweep 4096,8192,16384,65536,262144,1048576 --steps 300 --warmup 20 --sweep-out $O/r02b_env_sweep.json 2>&1 | tail -8 ) > $O/r02b_sweep.log
cd /tmp
prof() { # name, bench args...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02b_prof_$n -o $n --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $O/r02b_prof_$n.log 2>&1
  f=$(find $O/r02b_prof_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f" > $O/r02b_kernel_stats_$n.csv
  rm -rf $O/r02b_prof_$n
}
prof quadruped --steps 200 --warmup 50
prof quadruped_65536 --envs-per-gpu 65536 --steps 40 --warmup 10
prof quadruped_featherstone --workload quadruped_featherstone --steps 100 --warmup 10
prof box_stack --workload box_stack --steps 100 --warmup 10
prof quadruped_convex --workload quadruped_convex --steps 100 --warmup 10
prof hull_bin_2048 --workload hull_bin --envs-per-gpu 2048 --steps 10 --warmup 2
cd $R
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 quadruped@262144 2>&1 | tail -60 ) > $O/r02b_pmc_traffic.log
( timeout 300 python tools/pmc_sq.py quadruped 2>&1 | tail -30 ) > $O/r02b_pmc_sq.log
( timeout 300 python tools/pmc_sq.py quadruped set2 2>&1 | tail -30 ) > $O/r02b_pmc_sq_set2.log
( timeout 300 python tools/pmc_sq.py quadruped@65536 2>&1 | tail -30 ) > $O/r02b_pmc_sq_65536.log
( timeout 300 python tools/pmc_sq.py quadruped@65536 set2 2>&1 | tail -30 ) > $O/r02b_pmc_sq_65536_set2.log
rm -rf $O/pmc_sq_*/ $O/pmc_quadruped_*/ 2>/dev/null
echo done > $O/r02b_done
