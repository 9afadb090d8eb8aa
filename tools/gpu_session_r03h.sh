#!/bin/bash
# Round-3 GPU session H: staged SDF narrow phase (cull -> resolve -> reduce) against the one-workgroup-per-pair kernel.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -k "sdf or hydro" 2>&1 | tail -15 ) > $O/r03h_gputests_sdf.log
( timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03h_bench_sdf_bin_staged.json
( NT_SDF_STAGED=0 timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/r03h_bench_sdf_bin_single.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03h_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 > $O/r03h_prof.log 2>&1
f=$(find $O/r03h_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" > $O/r03h_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03h_prof
echo done > $O/r03h_done
