#!/bin/bash
# A/B of whole-library build variants on the headline workload (NEWTON_HIP_LIB override of the loader; measurement only)
for lib in "$@"; do
  echo -n "lib=$lib "
  NEWTON_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>&1 | tail -1 | \
    python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M env-steps/s', round(d['roofline']['kernel_ms'],4),'ms valid',d['valid_state'], d['validity_gate']['root_height_min'])
except Exception as e: print('FAILED', e)"
done
