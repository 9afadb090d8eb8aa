#!/bin/bash
# Round-6 GPU session F: device suite on the in-tree build, workload A/B against the round-5 library, and the outlier count of the C4 frame
# with the XPBD phases in IEEE arithmetic (variants/libdev_ieee.so) next to the fast arithmetic (variants/libdev_fast.so).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1
for l in libdev_fast.so libdev_ieee.so; do
  echo "== $l" >> $O/${T}_ieee_outliers.txt
  ( timeout 600 python tools/with_lib.py variants/$l -m pytest tests/test_gpu_full_size.py -m gpu -q -s -p no:cacheprovider -k "one_frame_vs_oracle and True" 2>&1 | grep "outliers\]\|passed\|failed\|parity.*lowered=True" | cut -c1-600 ) >> $O/${T}_ieee_outliers.txt
done
bash tools/gpu_session_r06E.sh $T "libr05ship.so product libr05ship.so product"
