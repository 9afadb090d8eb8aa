#!/bin/bash
# Round-4 GPU session E: hydroelastic face stage with 16-lane groups + two-pass marching cubes; register caps (3 / 4 waves per SIMD)
# as variants; kernel stats of the product.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04e
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_gpu_hydro_forces.py tests/test_gpu_hydro_bands.py tests/test_gpu_hydro_stack.py -m gpu -q 2>&1 | tail -12 ) > $O/${T}_gputests_hydro.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o hydro --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/${T}_bench_hydro_bin.json 2>$O/${T}_prof.log
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" > $O/${T}_kernel_stats_hydro_bin_2048.csv
rm -rf $O/${T}_prof
cd $R
for w in 3 4; do
  b timeout 900 python tools/with_lib.py $R/variants/libv_hyw$w.so bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/${T}_bench_hydro_bin_w$w.json
done
echo done > $O/${T}_done
