#!/bin/bash
# Round-6 GPU session G: the whole device suite (no -x) on the in-tree build; the C4 frame's outliers with IEEE vs fast XPBD arithmetic;
# sdf_bin (step-kernel compaction with rows) and the headline against the round-5 library.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1
( timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) > $O/${T}_gputests.log
for l in libdev_fast.so libdev_ieee.so; do
  echo "== $l" >> $O/${T}_ieee_outliers.txt
  ( timeout 600 python tools/with_lib.py variants/$l -m pytest tests/test_gpu_full_size.py -m gpu -q -s -p no:cacheprovider -k "one_frame_vs_oracle and True" 2>&1 | grep "outliers\]\|passed\|failed\|parity.*lowered=True" | cut -c1-700 ) >> $O/${T}_ieee_outliers.txt
done
for w in quadruped:300 sdf_bin:6; do
  IFS=: read wl steps <<< "$w"
  for lib in libr05ship.so product libr05ship.so product; do
    if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
    echo -n "$wl $lib " >> $O/${T}_ab_workloads.txt
    ( timeout 600 $cmd --no-cpu-baseline --workload $wl --steps $steps --warmup 4 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,4),'M env-steps/s', round(d['ms_per_step'],4),'ms/step valid',d['valid_state'], d['roofline'].get('hbm_peak_measured'))
except Exception as e: print('FAILED', e)" >> $O/${T}_ab_workloads.txt
  done
done
echo done > $O/${T}_done
