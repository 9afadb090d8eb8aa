#!/bin/bash
# Per-phase cycles of several NT_PHASE_TIMING variants on ONE box (measurement tool).  usage: tools/gpu_timing.sh TAG "envs..." lib[:cfg] ...
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; shift
ENVS=$1; shift
for n in $ENVS; do
for e in "$@"; do
  IFS=: read lib cfg <<< "$e"
  echo "== $lib cfg=${cfg:-default} envs=$n" >> $O/${T}_phase_timing.txt
  ( if [ -n "${cfg:-}" ]; then export NT_XPBD_CFG=$cfg; fi; NT_TIMING_ENVS=$n VARIANT_LIB=variants/$lib timeout 300 python tools/phase_timing.py 2>&1 | tail -10 ) >> $O/${T}_phase_timing.txt
done
done
echo done > $O/${T}_done
