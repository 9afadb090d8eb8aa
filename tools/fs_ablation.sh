# timing ablation of featherstone_rollout_kernel; needs  python tools/build_variant.py build_ab/libnewton_ablation.so -DNT_ABLATION
VARIANT_LIB=${VARIANT_LIB:-build_ab/libnewton_ablation.so}
for s in ${FS_SKIPS:-0 1 2 4 8 16 32 64 128 255}; do
  echo -n "skip=$s "; NT_DEBUG_SKIP=$s python tools/with_lib.py $VARIANT_LIB bench.py --no-cpu-baseline --workload quadruped_featherstone --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
