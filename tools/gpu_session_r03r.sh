#!/bin/bash
# Round-3 GPU session R: hydroelastic reduction cut short after each phase (measurement builds).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
for v in 1 2 3 4; do
  NEWTON_HIP_LIB=$R/build_ab/libnewton_hstop$v.so b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 --settle-frames 10 > $O/r03r_bench_hydro_bin_stop$v.json
done
echo done > $O/r03r_done
