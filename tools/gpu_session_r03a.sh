#!/bin/bash
# Round-3 GPU session A: the driver's gate (pytest -m gpu -x, smoke), the headline bench line, the per-call API line, config C5
# through the SDF leg inside CollisionPipeline.collide + SolverXPBD (bench line + rocprofv3 kernel stats).  Output: gpurun_out/r03a_*.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/r03a_gputests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/r03a_smoke.log
( timeout 300 python bench.py --steps 1000 --warmup 200 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03a_bench_default.json
( timeout 300 python bench.py --no-cpu-baseline --workload quadruped_api --steps 100 --warmup 20 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03a_bench_quadruped_api.json
( timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/r03a_bench_sdf_bin.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03a_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 --settle-frames 40 > $O/r03a_prof.log 2>&1
f=$(find $O/r03a_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" > $O/r03a_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03a_prof
echo done > $O/r03a_done
