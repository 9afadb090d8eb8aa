#!/bin/bash
# Round-4 GPU session M: hydroelastic face stage with contiguous item runs per wave (pair descriptors once per pair).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04m
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/${T}_bench_hydro_bin.json 2>$O/${T}_prof.log
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_hydro_bin.csv
rm -rf $O/${T}_prof
cd $R
( timeout 900 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_gpu_hydro_forces.py tests/test_gpu_hydro_bands.py tests/test_gpu_hydro_stack.py -m gpu -q 2>&1 | tail -4 ) > $O/${T}_gputests_hydro.log
echo done > $O/${T}_done
