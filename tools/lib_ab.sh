#!/bin/bash
# A/B of whole-library build variants on the headline workload (tools/with_lib.py; measurement only)
for lib in "$@"; do
  echo -n "lib=$lib "
  timeout 300 python tools/with_lib.py $lib bench.py --no-cpu-baseline --steps 300 --warmup 20 2>&1 | tail -1 | \
    python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M env-steps/s', round(d['roofline']['kernel_ms'],4),'ms valid',d['valid_state'], d['validity_gate']['root_height_min'])
except Exception as e: print('FAILED', e)"
done
