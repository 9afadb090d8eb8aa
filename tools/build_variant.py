#!/usr/bin/env python
"""Build a throw-away measurement variant of libnewton_hip.so with extra -D flags (never the product library):
    python tools/build_variant.py build_ab/libnewton_ablation.so -DNT_ABLATION
    python tools/build_variant.py build_ab/libnewton_timing.so -DNT_PHASE_TIMING
Use it through the loader's NEWTON_HIP_LIB override (announced on stderr).  Run here (hipcc cross-compiles) so that no GPU-minutes
go into compiling."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

out = os.path.abspath(sys.argv[1])
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = [f for f in g.HIP_FLAGS] + sys.argv[2:]
srcs = [os.path.join(g.CSRC, u) for u in g.UNITS]
cmd = [g.HIPCC, *flags, f'-DNT_BUILD_ID="variant{"".join(sys.argv[2:])}"', *srcs, "-o", out]
print(" ".join(cmd), flush=True)
subprocess.run(cmd, check=True)
