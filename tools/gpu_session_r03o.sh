#!/bin/bash
# Round-3 GPU session O: where the hydroelastic reduction's time goes (ablation builds; measurement only).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
for v in hskipred hskipagg; do
  NEWTON_HIP_LIB=$R/build_ab/libnewton_$v.so b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 --settle-frames 10 > $O/r03o_bench_hydro_bin_$v.json
done
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 --settle-frames 10 > $O/r03o_bench_hydro_bin.json
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin_faces --steps 3 --warmup 1 --settle-frames 10 > $O/r03o_bench_hydro_bin_faces.json
echo done > $O/r03o_done
