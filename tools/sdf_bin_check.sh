set -u
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for i in 1 2; do timeout 600 python bench.py --workload sdf_bin --no-cpu-baseline --steps 6 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sdf_bin', round(d['value']/1e6,4), round(d['ms_per_step'],3))"; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/chk_prof -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 2 --warmup 2 > /dev/null 2>&1
f=$(find $O/chk_prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -d, -f1-4 | cut -c1-150; rm -rf $O/chk_prof
