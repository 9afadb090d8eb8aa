#!/bin/bash
# Round-4 GPU session B: the XPBD phases in the `fused` arithmetic namespace (contraction + v_rcp / v_sqrt), collide phases IEEE:
# whole GPU suite (every failure listed), smoke, headline at 4 096 and 65 536 envs, secondary XPBD workloads.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r04b}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -80 ) > $O/${T}_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/${T}_bench_default.json
b timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/${T}_bench_65536.json
for w in quadruped_convex box_stack quadruped_api quadruped_featherstone; do
  b timeout 300 python bench.py --no-cpu-baseline --workload $w --steps 100 --warmup 20 > $O/${T}_bench_$w.json
done
echo done > $O/${T}_done
