#!/bin/bash
# Short GPU check after a kernel-side change: GPU suite, default bench line, PMC traffic of the headline at two env counts
# (so profiles/r02_pmc_traffic.json carries the id of the build that ships).  Output: gpurun_out/quick_*.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/quick_gputests.log
( timeout 400 python bench.py 2>&1 | tail -1 ) > $O/quick_bench_default.json
( timeout 600 python tools/pmc_traffic.py quadruped@4096 quadruped@65536 2>&1 | tail -40 ) > $O/quick_pmc_traffic.log
rm -rf $O/pmc_quadruped_*/ 2>/dev/null
echo done > $O/quick_done
