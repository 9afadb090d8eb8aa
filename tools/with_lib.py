#!/usr/bin/env python
"""MEASUREMENT TOOL -- run a script against a variant build of the kernel library (tools/build_variant.py):

    python tools/with_lib.py build_ab/libnewton_fastmath.so bench.py --no-cpu-baseline --steps 300
    python tools/with_lib.py build_ab/libnewton_fastmath.so -m pytest tests -m gpu -q

The product loader (newton_amd/_lib.py) has no environment override; this wrapper assigns its LIB_PATH before anything loads the
library and then runs the script (or module, with -m) in this process.  The loader announces the reassignment on stderr."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    lib = os.path.abspath(sys.argv[1])
    if not os.path.exists(lib):
        sys.exit(f"no such library: {lib}")
    from newton_amd import _lib

    assert _lib._lib is None
    _lib.LIB_PATH = lib
    # a variant built from older sources may lack entry points added since (measurement aids only): drop them from the loader's table
    # (looked up in the file, not by loading it: the library must be loaded AFTER torch so that both share one HIP runtime)
    blob = open(lib, "rb").read()
    for name in ("nt_bandwidth_probe",):
        if name.encode() not in blob:
            _lib.SYMBOLS.pop(name, None)
    if sys.argv[2] == "-m":
        sys.argv = sys.argv[3:]
        runpy.run_module(sys.argv[0], run_name="__main__", alter_sys=True)
    else:
        sys.argv = sys.argv[2:]
        sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
        runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
