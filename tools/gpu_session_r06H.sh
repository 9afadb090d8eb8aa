#!/bin/bash
# Round-6 GPU session H: HBM copy shapes; SolverFeatherstone's uniform 16-environment tile (device tests + A/B); sdf_bin with the compacting
# pair-heavy step kernel.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1
( ./variants/hbm_copy.bin 2>&1 | tail -8 ) > $O/${T}_hbm_copy.jsonl
( timeout 900 python -m pytest tests/test_gpu_parity_featherstone.py tests/test_gpu_full_size.py tests/test_gpu_sdf_pipeline.py tests/test_zz_pair_heavy_gpu.py -m gpu -q -p no:cacheprovider -k "not c5_geometry" 2>&1 | tail -8 ) > $O/${T}_gputests.log
for w in quadruped_featherstone:100 sdf_bin:6; do
  IFS=: read wl steps <<< "$w"
  for lib in libr05ship.so product libr05ship.so product; do
    if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
    echo -n "$wl $lib " >> $O/${T}_ab_workloads.txt
    ( timeout 600 $cmd --no-cpu-baseline --workload $wl --steps $steps --warmup 4 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab_workloads.txt
  done
done
for n in 2048 8192 16384; do
  echo -n "quadruped_featherstone product envs=$n " >> $O/${T}_ab_workloads.txt
  ( timeout 600 python bench.py --no-cpu-baseline --workload quadruped_featherstone --envs-per-gpu $n --steps 60 --warmup 4 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab_workloads.txt
done
echo done > $O/${T}_done
