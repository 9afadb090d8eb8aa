#!/bin/bash
# Round-4 GPU session F: texel-pair sampler + corner-parallel hydroelastic face stage + reduce diet: SDF / hydro tests, hydro_bin and
# sdf_bin with kernel stats, the per-call API workload's kernel split.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04f
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1200 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_gpu_sdf.py tests/test_gpu_sdf_headon.py tests/test_gpu_hydro_forces.py tests/test_gpu_hydro_bands.py tests/test_gpu_hydro_stack.py tests/test_gpu_viewer_recorder.py -m gpu -q 2>&1 | tail -12 ) > $O/${T}_gputests_sdf.log
cd /tmp
for w in hydro_bin sdf_bin quadruped_api; do
  st=3; [ $w = sdf_bin ] && st=10; [ $w = quadruped_api ] && st=100
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_$w -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload $w --steps $st --warmup 2 > $O/${T}_bench_$w.json 2>$O/${T}_prof_$w.log
  f=$(find $O/${T}_prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -14 "$f" > $O/${T}_kernel_stats_$w.csv
  rm -rf $O/${T}_prof_$w
done
echo done > $O/${T}_done
