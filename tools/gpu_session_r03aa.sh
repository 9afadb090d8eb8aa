#!/bin/bash
# Round-3 GPU session AA: the final build -- full GPU suite, smoke, PMC traffic and SQ passes at 4 096 and 65 536 envs, headline bench and
# env sweep with the traffic of this build attached.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/r03aa_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/r03aa_smoke.log
rm -f $O/r03_pmc_traffic.json
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -40 ) > $O/r03aa_pmc_traffic.log
( timeout 300 python tools/pmc_sq.py quadruped 2>&1 | tail -20 ) > $O/r03aa_pmc_sq.log
( timeout 300 python tools/pmc_sq.py quadruped@65536 2>&1 | tail -20 ) > $O/r03aa_pmc_sq_65536.log
cp $O/pmc_sq_quadruped.json $O/r03aa_pmc_sq_quadruped_4096.json 2>/dev/null
cp $O/pmc_sq_quadruped_65536.json $O/r03aa_pmc_sq_quadruped_65536.json 2>/dev/null
rm -rf $O/pmc_sq_*/ $O/pmc_quadruped_*/ 2>/dev/null
cp $O/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json 2>/dev/null
b timeout 400 python bench.py > $O/r03aa_bench_default.json
( timeout 600 python bench.py --no-cpu-baseline --sweep 4096,65536 2>&1 | grep -v amdgpu.ids | tail -3 ) > $O/r03aa_env_sweep.json
b timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 > $O/r03aa_bench_sdf_bin.json
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 5 --warmup 2 > $O/r03aa_bench_hydro_bin.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03aa_prof -o sdf --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 5 --warmup 2 > $O/r03aa_prof.log 2>&1
f=$(find $O/r03aa_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" > $O/r03aa_kernel_stats_sdf_bin_2048.csv
rm -rf $O/r03aa_prof
echo done > $O/r03aa_done
