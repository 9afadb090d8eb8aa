#!/bin/bash
# Round-6 GPU session U: the hydroelastic face stage with a lane per face -- its GPU tests, the kernel averages of one hydro_bin frame
# (rocprofv3) and the bench lines that moved.  usage: tools/gpu_session_r06U.sh [TAG]
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r06U}
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_gpu_hydro_bands.py tests/test_gpu_hydro_forces.py tests/test_gpu_hydro_stack.py tests/test_gpu_sdf.py tests/test_gpu_parity_featherstone.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12 ) > $O/${T}_gputests.log
bash tools/gpu_session_r06Q.sh $T product
b timeout 600 python bench.py --no-cpu-baseline --workload hydro_bin --steps 2 --warmup 4 > $O/${T}_bench_hydro_bin.json
b timeout 600 python bench.py --no-cpu-baseline --workload quadruped_featherstone --steps 100 --warmup 4 > $O/${T}_bench_quadruped_featherstone.json
echo done > $O/${T}_done
