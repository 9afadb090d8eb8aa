set -u
O=gpurun_out; T=r06M; export TMPDIR=/tmp; R=$PWD
( timeout 900 python -m pytest tests/test_gpu_parity_featherstone.py tests/test_gpu_full_size.py tests/test_zx_round2_gpu.py tests/test_zy_recent_gpu.py -m gpu -q -p no:cacheprovider -k "featherstone or c3 or fs" 2>&1 | tail -4 ) > $O/${T}_gputests.log
( python bench.py --no-cpu-baseline --workload quadruped_featherstone --steps 200 --warmup 10 2>&1 | grep -v amdgpu.ids | tail -1 ) > $O/${T}_bench_quadruped_featherstone_nocounters.json
( NT_PMC_OUT=${T}_pmc_traffic.json timeout 600 python tools/pmc_traffic.py quadruped_featherstone@4096 2>&1 | tail -5 ) > $O/${T}_pmc_traffic.log
rm -rf $O/pmc_quadruped_featherstone_*
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/${T}_prof_f -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload quadruped_featherstone --steps 400 --warmup 40 > $R/$O/${T}_prof_f.log 2>&1
cd $R
f=$(find $O/${T}_prof_f -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" > $O/${T}_kernel_stats_featherstone.csv; rm -rf $O/${T}_prof_f
