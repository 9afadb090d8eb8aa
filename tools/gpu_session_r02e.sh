#!/bin/bash
# Round-2 GPU session E (through gpurun from the repo root, on the build that ships): GPU suite, headline bench (default and
# driver shape), PMC traffic of the headline kernel for this build id, and the C5 contact-model collide line (SAP broad phase +
# mesh-SDF contacts + global contact reduction) with its rocprofv3 kernel stats.  Output: gpurun_out/r02e_*.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/r02e_gputests.log
( timeout 400 python bench.py 2>&1 | tail -1 ) > $O/r02e_bench_default.json
( timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > $O/r02e_bench_driver_shape.json
for t in 64 256; do
  ( timeout 300 python tools/sdf_bin_bench.py --envs 256 --settle-frames 120 --threads $t --unreduced 2>&1 | tail -1 ) > $O/r02e_sdf_bin_256_t$t.json
done
( timeout 600 python tools/sdf_bin_bench.py --envs 2048 --settle-frames 120 --threads 64 --unreduced 2>&1 | tail -3 ) > $O/r02e_sdf_bin_2048_t64.json
( timeout 600 python tools/sdf_bin_bench.py --envs 2048 --settle-frames 120 --threads 128 2>&1 | tail -3 ) > $O/r02e_sdf_bin_2048_t128.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02e_prof_sdf -o sdf --output-format csv -- python $R/tools/sdf_bin_bench.py --envs 2048 --settle-frames 120 --threads 64 --steps 20 > $O/r02e_prof_sdf.log 2>&1
f=$(find $O/r02e_prof_sdf -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > $O/r02e_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r02e_prof_sdf
cd $R
( timeout 600 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -40 ) > $O/r02e_pmc_traffic.log
rm -rf $O/pmc_quadruped_*/ 2>/dev/null
echo done > $O/r02e_done
