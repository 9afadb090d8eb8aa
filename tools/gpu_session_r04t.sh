#!/bin/bash
# Round-4 GPU session T (the shipped build, src e2dd20f1a794): the secondary workload lines and the env sweep on this build id.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04t
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
for w in quadruped_convex box_stack quadruped_api; do
  b timeout 200 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 5 > $O/${T}_bench_$w.json
done
( timeout 300 python bench.py --no-cpu-baseline --sweep 4096,8192,65536,262144 --sweep-out $O/${T}_env_sweep.json 2>&1 | grep -v amdgpu.ids | tail -4 ) > $O/${T}_env_sweep.log
b timeout 200 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 2 > $O/${T}_bench_sdf_bin.json
b timeout 200 python bench.py --no-cpu-baseline --workload hull_bin --steps 5 --warmup 2 > $O/${T}_bench_hull_bin.json
b timeout 300 python bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 2 > $O/${T}_bench_hydro_bin.json
echo done > $O/${T}_done
