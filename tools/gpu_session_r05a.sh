#!/bin/bash
# Round-5 GPU session A: VALU issue microbenchmark, the device tests that had only seen the emulator at the end
# of round 4, the headline on this box (baseline of the round) and the first bench line of the vertex leg.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=r05a
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 120 variants/valu_issue.bin 2>&1 ) > $O/${T}_valu_issue.jsonl
( timeout 600 python -m pytest tests/test_gpu_mesh_plane_pipeline.py -m gpu -q 2>&1 | tail -8 ) > $O/${T}_gputests_mesh_plane.log
b timeout 400 python bench.py --no-cpu-baseline > $O/${T}_bench_default.json
b timeout 300 python bench.py --no-cpu-baseline --workload mesh_ground --steps 20 --warmup 3 > $O/${T}_bench_mesh_ground.json
echo done > $O/${T}_done
