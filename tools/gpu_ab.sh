#!/bin/bash
# Same-box A/B of kernel variants on the headline (measurement tool).  usage: tools/gpu_ab.sh TAG "lib[:NT_XPBD_CFG[:envs]] ..." [timing-lib]
# Each entry runs bench.py (300 steps at 4096 envs unless stated) through tools/with_lib.py; `product` = the in-tree library.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; shift
ENTRIES=$1; shift
for rep in 1 2; do
for e in $ENTRIES; do
  IFS=: read lib cfg envs <<< "$e"
  envs=${envs:-4096}
  steps=$(( 4096 * 300 / envs )); [ $steps -lt 30 ] && steps=30
  echo -n "$lib cfg=${cfg:-default} envs=$envs " >> $O/${T}_ab.txt
  if [ "$lib" = product ]; then cmd="python bench.py"; else cmd="python tools/with_lib.py variants/$lib bench.py"; fi
  ( if [ -n "${cfg:-}" ]; then export NT_XPBD_CFG=$cfg; fi; timeout 300 $cmd --no-cpu-baseline --envs-per-gpu $envs --steps $steps --warmup 20 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M env-steps/s', round(d['roofline']['kernel_ms'],4),'ms valid',d['valid_state'], d['roofline']['kernel'])
except Exception as e: print('FAILED', e)" >> $O/${T}_ab.txt
done
done
if [ $# -ge 1 ]; then
  ( VARIANT_LIB=variants/$1 timeout 300 python tools/phase_timing.py 2>&1 | tail -12 ) > $O/${T}_phase_timing_quadruped.txt
fi
echo done > $O/${T}_done
