#!/bin/bash
# Round-2 GPU session H: stability sweep of the stepped C5 contact model, heterogeneous-world tests on the device, whole GPU
# suite without -x.  Output: gpurun_out/r02h_*.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python tools/sdf_stage_debug.py 2>&1 | grep -v amdgpu.ids ) > $O/r02h_sdf_stage_sweep.jsonl
( timeout 300 python -m pytest tests/test_heterogeneous_worlds.py -m gpu -q 2>&1 | tail -15 ) > $O/r02h_hetero.log
( timeout 700 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/r02h_gputests.log
echo done > $O/r02h_done
