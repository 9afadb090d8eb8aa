#!/bin/bash
# Round-4 GPU session C: the staged hydroelastic pipeline on the device (its tests, the full-size C5 test, A/B against the single
# kernel at 2 048 worlds with kernel stats), the re-gated full-size tests, then the whole GPU suite.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04c
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_gpu_hydro_forces.py tests/test_gpu_hydro_bands.py tests/test_gpu_hydro_stack.py tests/test_gpu_full_size.py -m gpu -q --durations=8 2>&1 | tail -60 ) > $O/${T}_gputests_hydro.log
b timeout 900 python bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/${T}_bench_hydro_bin_staged.json
b timeout 900 python bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 --hydro-single-kernel > $O/${T}_bench_hydro_bin_single.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o hydro --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 2 --warmup 1 > $O/${T}_prof.log 2>&1
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" > $O/${T}_kernel_stats_hydro_bin_2048.csv
rm -rf $O/${T}_prof
cd $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/${T}_gputests.log
echo done > $O/${T}_done
