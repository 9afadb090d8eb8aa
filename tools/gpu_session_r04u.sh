#!/bin/bash
# Round-4 GPU session U (build src 2161716a98a2 = session S's stepping unit byte for byte (e9e5f0b2fc6f), nt_sdf.hip with identical
# device code (its reducer helpers moved to nt_contact_reduce.hpp) + the swept broad-phase entry points + the mesh vertex leg): smoke, headline line with the cpu baseline (PMC traffic attached through the stepping-unit
# hash e9e5f0b2fc6f), driver-shape line, kernel stats of the headline.  Bounded to ~90 s of box time (what is left of the round).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r04u
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/${T}_smoke.log
b timeout 90 python bench.py > $O/${T}_bench_default.json
b timeout 40 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_driver_shape.json
cd /tmp
timeout 60 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_q -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 600 --warmup 100 > $O/${T}_prof_q.log 2>&1
f=$(find $O/${T}_prof_q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/${T}_kernel_stats_quadruped.csv; rm -rf $O/${T}_prof_q
echo done > $O/${T}_done
