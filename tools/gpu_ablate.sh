#!/bin/bash
# Phase cycles with whole phases skipped (NT_ABLATION variant): what an empty barrier interval costs.  usage: tools/gpu_ablate.sh TAG lib masks...
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; lib=$2; shift; shift
for m in "$@"; do
  echo "== skip mask $m" >> $O/${T}_ablation.txt
  ( NT_DEBUG_SKIP=$m VARIANT_LIB=variants/$lib timeout 300 python tools/phase_timing.py 2>&1 | tail -11 | grep -v prologue ) >> $O/${T}_ablation.txt
done
echo done > $O/${T}_done
