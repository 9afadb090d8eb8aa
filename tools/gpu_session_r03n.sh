#!/bin/bash
# Round-3 GPU session N: hydroelastic reduction (reduce_contacts=True) on the device: parity tests, bench with and without reduction.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests -m gpu -q -k "hydro or sdf" 2>&1 | tail -15 ) > $O/r03n_gputests_hydro.log
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 5 --warmup 2 > $O/r03n_bench_hydro_bin.json
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin_faces --steps 5 --warmup 2 > $O/r03n_bench_hydro_bin_faces.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03n_prof -o h --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/r03n_prof.log 2>&1
f=$(find $O/r03n_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > $O/r03n_kernel_stats_hydro_bin_256.csv
rm -rf $O/r03n_prof
echo done > $O/r03n_done
