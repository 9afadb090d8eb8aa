#!/bin/bash
# Round-2 GPU session F (through gpurun from the repo root, on the build that ships): GPU suite, headline bench (default and
# driver shape), PMC traffic of the headline kernel for this build id, the C5 contact-model lines (collide only: SAP broad phase +
# mesh-SDF contacts + global contact reduction; stepped: + write_contact rows + eval_body_contact + SolverSemiImplicit) and the
# rocprofv3 kernel stats of both.  Output: gpurun_out/r02f_*.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/r02f_gputests.log
( timeout 400 python bench.py 2>&1 | tail -1 ) > $O/r02f_bench_default.json
( timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02f_bench_driver_shape.json
( timeout 300 python tools/sdf_bin_bench.py --envs 256 --settle-frames 120 --threads 64 --unreduced 2>&1 | tail -2 ) > $O/r02f_sdf_bin_256_t64.json
( timeout 300 python tools/sdf_bin_bench.py --envs 256 --settle-frames 120 --threads 256 2>&1 | tail -2 ) > $O/r02f_sdf_bin_256_t256.json
( timeout 600 python tools/sdf_bin_bench.py --envs 2048 --settle-frames 120 --threads 64 --unreduced 2>&1 | tail -2 ) > $O/r02f_sdf_bin_2048_t64.json
( timeout 600 python tools/sdf_bin_bench.py --envs 2048 --settle-frames 120 --threads 128 2>&1 | tail -2 ) > $O/r02f_sdf_bin_2048_t128.json
( timeout 600 python tools/sdf_bin_bench.py --step --envs 256 --settle-frames 100 --steps 20 2>&1 | tail -2 ) > $O/r02f_sdf_step_256.json
( timeout 900 python tools/sdf_bin_bench.py --step --envs 2048 --settle-frames 60 --steps 10 2>&1 | tail -2 ) > $O/r02f_sdf_step_2048.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02f_prof_sdf -o sdf --output-format csv -- python $R/tools/sdf_bin_bench.py --envs 2048 --settle-frames 60 --threads 64 --steps 20 > $O/r02f_prof_sdf.log 2>&1
f=$(find $O/r02f_prof_sdf -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > $O/r02f_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r02f_prof_sdf
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02f_prof_step -o step --output-format csv -- python $R/tools/sdf_bin_bench.py --step --envs 2048 --settle-frames 10 --steps 10 > $O/r02f_prof_step.log 2>&1
f=$(find $O/r02f_prof_step -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" > $O/r02f_kernel_stats_sdf_step_2048.csv
rm -rf $O/r02f_prof_step
cd $R
( timeout 600 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -40 ) > $O/r02f_pmc_traffic.log
rm -rf $O/pmc_quadruped_*/ 2>/dev/null
echo done > $O/r02f_done
