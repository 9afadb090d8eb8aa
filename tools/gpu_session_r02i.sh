#!/bin/bash
# Round-2 GPU session I: the driver's GPU gate (pytest -m gpu -x, smoke), the stepped C5 line with its SDF contact model on the
# stabilised scene + its rocprofv3 kernel stats, a mixed-world (heterogeneous) timing.  Output: gpurun_out/r02i_*.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/r02i_gputests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/r02i_smoke.log
( timeout 300 python tools/sdf_bin_bench.py --step --envs 256 --settle-frames 100 --steps 20 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r02i_sdf_step_256.json
( timeout 400 python tools/sdf_bin_bench.py --step --envs 2048 --settle-frames 60 --steps 10 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r02i_sdf_step_2048.json
( timeout 300 python tools/hetero_bench.py 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/r02i_hetero_bench.jsonl
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/r02i_prof_step -o step --output-format csv -- python $R/tools/sdf_bin_bench.py --step --envs 2048 --settle-frames 10 --steps 10 > $O/r02i_prof_step.log 2>&1
f=$(find $O/r02i_prof_step -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" > $O/r02i_kernel_stats_sdf_step_2048.csv
rm -rf $O/r02i_prof_step
echo done > $O/r02i_done
