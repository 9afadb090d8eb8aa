#!/bin/bash
# Round-4 GPU session A: the parity closures on the device (GPU suite with the reference-asset quadruped, C3 with contacts, the
# hydroelastic half of C5 at 2 048 worlds, SDF row overflow), smoke, headline bench, and the A/B of compiler-level variants of the
# headline kernel (fast division / contraction / ILP scheduling) through tools/with_lib.py.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
b() { ( "$@" 2>&1 | grep -v amdgpu.ids | tail -1 ); }
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 2>&1 | tail -40 ) > $O/r04a_gputests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/r04a_smoke.log
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/r04a_bench_default.json
for v in fast contract ilp fc all; do
  [ -f $R/variants/libv_$v.so ] && b timeout 300 python tools/with_lib.py $R/variants/libv_$v.so bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/r04a_bench_variant_$v.json
done
b timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $O/r04a_bench_default_again.json
b timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/r04a_bench_65536.json
for v in fc all; do
  [ -f $R/variants/libv_$v.so ] && b timeout 300 python tools/with_lib.py $R/variants/libv_$v.so bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 60 --warmup 10 > $O/r04a_bench_65536_variant_$v.json
done
b timeout 900 python bench.py --no-cpu-baseline --workload hydro_bin --steps 2 --warmup 1 > $O/r04a_bench_hydro_bin_2048.json
echo done > $O/r04a_done
