#!/bin/bash
# Round-4 GPU session J: batched staging loads (per-call API, headline), the live-contact list only in analytic XPBD layouts (C2 / C4
# convex / C3 back on their tiles), register caps on the hydroelastic face stage (latency-bound?).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04j
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default.json
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default_again.json
for w in quadruped_api quadruped_convex box_stack quadruped_featherstone; do
  b timeout 300 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 20 > $O/${T}_bench_$w.json
done
b timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/${T}_bench_65536.json
for w in 3 4; do
  b timeout 900 python tools/with_lib.py $R/variants/libv_hyw$w.so bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/${T}_bench_hydro_bin_w$w.json
done
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/${T}_gputests.log
echo done > $O/${T}_done
