#!/bin/bash
# Round-5 GPU session A (prepared at the end of round 4, not yet run): the first call of the next round.  Full GPU suite on the build
# of the tree (incl. the device tests that were only pre-flighted on the emulator: test_gpu_mesh_plane_pipeline.py's reference mirror,
# matching and heterogeneous-world tests), smoke, PMC traffic of this build id, headline with cpu_baseline + driver shape + kernel
# stats, and the first bench line + kernel split of the vertex leg (mesh_ground).  ~9 minutes of box time.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r05a
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/${T}_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
( timeout 900 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -20 ) > $O/${T}_pmc_traffic.log
b timeout 400 python bench.py > $O/${T}_bench_default.json
b timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
b timeout 300 python bench.py --no-cpu-baseline --workload mesh_ground --steps 20 --warmup 3 > $O/${T}_bench_mesh_ground.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 1500 --warmup 100 > $O/${T}_prof_q.log 2>&1
f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_m -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload mesh_ground --steps 20 --warmup 3 > $O/${T}_prof_m.log 2>&1
f=$(find $O/${T}_prof_m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" > $O/${T}_kernel_stats_mesh_ground.csv; rm -rf $O/${T}_prof_m
echo done > $O/${T}_done
