#!/bin/bash
# Round-4 GPU session S (the shipped build): full GPU suite, smoke, PMC traffic of this build id, headline line with the cpu baseline,
# driver-shape line, kernel stats of the headline, C3 in both mass-matrix modes.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04s
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/${T}_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
rm -f $O/r04_pmc_traffic.json
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -20 ) > $O/${T}_pmc_traffic.log
cp $O/r04_pmc_traffic.json $R/profiles/r04_pmc_traffic.json 2>/dev/null
b timeout 400 python bench.py > $O/${T}_bench_default.json
b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
for mm in tree dense; do
  b timeout 300 python bench.py --no-cpu-baseline --workload quadruped_featherstone --fs-mass-matrix $mm --steps 100 --warmup 5 > $O/${T}_bench_featherstone_$mm.json
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
echo done > $O/${T}_done
