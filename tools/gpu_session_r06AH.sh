bash tools/gpu_ab_env.sh r06AH NT_HYDRO_REDUCE_THREADS 256 64 "hydro_bin:2" | tail -4
bash tools/gpu_ab_env.sh r06AH2 NT_HYDRO_REDUCE_THREADS 256 128 "hydro_bin:2" | tail -4
NT_HYDRO_REDUCE_THREADS=64 timeout 900 python -m pytest tests/test_gpu_hydro_bands.py tests/test_gpu_hydro_forces.py tests/test_gpu_hydro_stack.py tests/test_gpu_sdf_pipeline.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
