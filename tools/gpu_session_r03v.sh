#!/bin/bash
# Round-3 GPU session V: the build that ships (after the hydroelastic reduction) -- full GPU suite, smoke, headline bench with PMC
# traffic of this build, rocprofv3 kernel stats, the SDF / hydroelastic workloads.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/r03v_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/r03v_smoke.log
b timeout 400 python bench.py > $O/r03v_bench_default.json
b timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r03v_bench_driver_shape.json
b timeout 600 python bench.py --no-cpu-baseline --workload sdf_bin --steps 10 --warmup 3 > $O/r03v_bench_sdf_bin.json
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 5 --warmup 2 > $O/r03v_bench_hydro_bin.json
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin_faces --steps 5 --warmup 2 > $O/r03v_bench_hydro_bin_faces.json
( timeout 400 python tools/pmc_traffic.py quadruped@4096 2>&1 | tail -30 ) > $O/r03v_pmc_traffic.log
rm -rf $O/pmc_quadruped_*/ 2>/dev/null
cp $O/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json 2>/dev/null
b timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/r03v_bench_with_traffic.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r03v_prof_q -o q --output-format csv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/r03v_prof_q.log 2>&1
f=$(find $O/r03v_prof_q -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -10 "$f" > $O/r03v_kernel_stats_quadruped.csv
rm -rf $O/r03v_prof_q
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03v_prof -o h --output-format csv -- python $R/bench.py --no-cpu-baseline --workload hydro_bin --steps 3 --warmup 1 > $O/r03v_prof.log 2>&1
f=$(find $O/r03v_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" > $O/r03v_kernel_stats_hydro_bin_256.csv
rm -rf $O/r03v_prof
echo done > $O/r03v_done
