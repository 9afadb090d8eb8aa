#!/bin/bash
# Round-3 GPU session D: full GPU suite on the build with the hydroelastic leg, the SDF bin lines (edge contacts / hydroelastic),
# rocprofv3 kernel stats of both.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/r03d_gputests.log
( timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03d_bench_sdf_bin.json
( timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/r03d_bench_hydro_bin.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03d_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 --settle-frames 20 > $O/r03d_prof.log 2>&1
f=$(find $O/r03d_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -20 "$f" > $O/r03d_kernel_stats_hydro_bin_256.csv
rm -rf $O/r03d_prof
echo done > $O/r03d_done
