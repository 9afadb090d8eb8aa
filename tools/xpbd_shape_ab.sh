#!/bin/bash
# A/B of the launch shapes of the analytic fused XPBD rollout (NT_XPBD_CFG = envs per workgroup, workgroup size, min waves per
# SIMD, uniform-parameter tile) on the headline workload; prints env-steps/s and kernel ms per shape.  usage: tools/xpbd_shape_ab.sh [envs] [shapes...]
ENVS=${1:-4096}; shift
SHAPES=${@:-"16,512,1,0 32,512,1,1 16,256,2,1"}  # the shapes of the product build; the full A/B list needs a library built with -DNT_ALL_SHAPES
STEPS=$(( 4096 * 300 / ENVS )); [ $STEPS -lt 20 ] && STEPS=20
for s in $SHAPES; do
  echo -n "shape=$s envs=$ENVS "
  NT_XPBD_CFG=$s timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $ENVS --steps $STEPS --warmup 20 2>&1 | tail -1 | \
    python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M env-steps/s', round(d['roofline']['kernel_ms'],4),'ms valid',d['valid_state'])
except Exception as e: print('FAILED', e)"
done
