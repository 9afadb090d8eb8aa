# timing ablation of xpbd_rollout_kernel (bits: 1 collide, 2 forces+integrate, 4 contacts, 8 joints, 16 apply).  The product
# library has no work-skipping switch: this needs the throw-away build  python tools/build_variant.py build_ab/libnewton_ablation.so -DNT_ABLATION
VARIANT_LIB=${VARIANT_LIB:-build_ab/libnewton_ablation.so}
for s in ${SKIPS:-0 1 2 4 8 16 31}; do
  echo -n "skip=$s "; NT_DEBUG_SKIP=$s python tools/with_lib.py $VARIANT_LIB bench.py --no-cpu-baseline --steps 100 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done
