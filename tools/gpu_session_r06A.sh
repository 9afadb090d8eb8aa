#!/bin/bash
# Round-6 GPU session A: same-box A/B of dev variants on the headline (scheduler strategies of the compiler; cheaper rotations + the
# static-side skip of contact_solve), ABBA on the base.  usage: tools/gpu_session_r06A.sh TAG lib [lib ...]   (files under variants/)
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
T=$1; shift
run() {  # lib envs steps
  echo -n "$1 envs=$2 " >> $O/${T}_ab.txt
  ( timeout 400 python tools/with_lib.py variants/$1 bench.py --no-cpu-baseline --envs-per-gpu $2 --steps $3 --warmup 20 2>&1 | grep -v amdgpu.ids | tail -1 ) | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M env-steps/s', round(d['ms_per_step'],4),'ms valid',d['valid_state'], d['roofline'].get('kernel',''))
except Exception as e: print('FAILED', e)" >> $O/${T}_ab.txt
}
for lib in "$@"; do run $lib 4096 400; done
for lib in "$@"; do run $lib 4096 400; done
run $1 65536 60
run ${@: -1} 65536 60
echo done > $O/${T}_done
