#!/bin/bash
# Round-4 GPU session K: whole contact records fetched up front (one memory round trip per contact phase instead of three).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r04k}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default.json
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default_again.json
b timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/${T}_bench_65536.json
for w in quadruped_api quadruped_convex box_stack; do
  b timeout 300 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 20 > $O/${T}_bench_$w.json
done
echo done > $O/${T}_done
