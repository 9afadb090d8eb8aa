#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- dry-run the `-m gpu` test files on a machine without a GPU: product Python path unchanged,
"cuda" tensors redirected to host memory, libnewton_hip.so replaced by the emulated kernel library (emu_plugin.py).

    python tests/emu/run_gpu_tests_emulated.py [test files ...]      # default: every GPU file except the full-size ones

Slow (one OS thread per GPU thread): the default set takes on the order of half an hour on 8 cores, so it is a pre-flight
check before spending GPU time, not part of the regular `-m "not gpu"` suite."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
SKIP = {"test_gpu_full_size.py"}  # 4096 environments: hours in emulation


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(TESTS) if f.startswith("test_") and f.endswith(".py") and f not in SKIP
                                   and "pytest.mark.gpu" in open(os.path.join(TESTS, f)).read())
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([HERE, ROOT, os.environ.get("PYTHONPATH", "")]))
    failed = []
    for f in files:
        r = subprocess.run([sys.executable, "-m", "pytest", "-p", "emu_plugin", "-m", "gpu", "-q", f], cwd=TESTS, env=env)
        if r.returncode not in (0, 5):  # 5: no tests collected
            failed.append(f)
    print("emulated GPU suite:", "all green" if not failed else f"FAILED {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
