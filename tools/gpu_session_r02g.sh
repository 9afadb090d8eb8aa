#!/bin/bash
# Round-2 GPU session G (re-run of session F's essentials on the build that ships, after the container was re-created):
# GPU suite, headline bench (default + driver shape), C5 contact-model lines (collide stage and stepped), rocprofv3 kernel stats
# of the C5 collide stage, PMC traffic of the headline kernel.  Output: gpurun_out/r02g_*.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/r02g_gputests.log
( timeout 300 python bench.py 2>&1 | tail -1 ) > $O/r02g_bench_default.json
( timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02g_bench_driver_shape.json
( timeout 200 python tools/sdf_bin_bench.py --envs 2048 --settle-frames 120 --threads 64 --unreduced 2>&1 | tail -2 ) > $O/r02g_sdf_bin_2048_t64.json
( timeout 200 python tools/sdf_bin_bench.py --envs 2048 --settle-frames 120 --threads 128 2>&1 | tail -2 ) > $O/r02g_sdf_bin_2048_t128.json
( timeout 300 python tools/sdf_bin_bench.py --step --envs 2048 --settle-frames 60 --steps 10 2>&1 | tail -2 ) > $O/r02g_sdf_step_2048.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02g_prof_sdf -o sdf --output-format csv -- python $R/tools/sdf_bin_bench.py --envs 2048 --settle-frames 60 --threads 64 --steps 20 > $O/r02g_prof_sdf.log 2>&1
f=$(find $O/r02g_prof_sdf -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > $O/r02g_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r02g_prof_sdf
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02g_prof_q -o q --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/r02g_prof_q.log 2>&1
f=$(find $O/r02g_prof_q -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -10 "$f" > $O/r02g_kernel_stats_quadruped.csv
rm -rf $O/r02g_prof_q
cd $R
( timeout 300 python tools/pmc_traffic.py quadruped@4096 2>&1 | tail -30 ) > $O/r02g_pmc_traffic.log
rm -rf $O/pmc_quadruped_*/ 2>/dev/null
echo done > $O/r02g_done
