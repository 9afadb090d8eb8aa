#!/bin/bash
# Round-5 GPU session M: the full device suite + every bench line on the product build after the body-lane / LDS-record restructure.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r05m}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/${T}_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
b timeout 400 python bench.py > $O/${T}_bench_default.json
b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
for w in quadruped_convex box_stack quadruped_featherstone quadruped_api hull_bin sdf_bin; do
  b timeout 400 python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 5 > $O/${T}_bench_$w.json
done
b timeout 600 python bench.py --no-cpu-baseline --sweep 4096,8192,65536,262144 --sweep-out $O/${T}_env_sweep.json > /dev/null
echo done > $O/${T}_done
