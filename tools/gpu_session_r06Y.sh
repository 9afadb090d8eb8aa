#!/bin/bash
# Round-6 GPU session Y: the SDF samplers with global loads / 32-bit offsets -- device tests of every SDF leg, same-box A/B against the
# previous build of nt_sdf.hip (variants/libsdf_prev.so) on hydro_bin and sdf_bin, kernel averages of both workloads.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=${1:-r06Y}
( timeout 1500 python -m pytest tests/test_gpu_sdf_pipeline.py tests/test_gpu_hydro_bands.py tests/test_gpu_hydro_forces.py tests/test_gpu_hydro_stack.py tests/test_gpu_sdf.py tests/test_gpu_sdf_headon.py tests/test_gpu_mesh_plane_pipeline.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12 ) > $O/${T}_gputests.log
bash tools/gpu_ab_generic.sh $T "hydro_bin:2 sdf_bin:6" "product libsdf_prev.so product libsdf_prev.so"
bash tools/gpu_session_r06Q.sh $T product
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_s -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --workload sdf_bin --steps 2 --warmup 2 > $O/${T}_prof_sdf.log 2>&1
f=$(find $O/${T}_prof_s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -d, -f1-4 | cut -c1-150 > $O/${T}_kernel_stats_sdf_bin.csv; rm -rf $O/${T}_prof_s
cd $R
echo done > $O/${T}_done
