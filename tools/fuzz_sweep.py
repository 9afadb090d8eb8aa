"""Wide fuzz sweep of the emulated kernels against the oracle (tests/test_emu_fuzz.py logic over a seed range, every mode).
usage: python tools/fuzz_sweep.py SEED_LO SEED_HI   -- test infrastructure, CPU only."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R, R+"/tests", R+"/tests/emu"]
os.chdir(R+"/tests")
import pytest
import harness
harness.lib()
import test_emu_fuzz as F
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    for mode, extras in (("xpbd", False), ("free", False), ("semi_implicit", False), ("featherstone", False), ("xpbd_jitter", False), ("xpbd", True), ("featherstone", True)):
        try:
            F._run(harness, mode, seed, extras)
        except pytest.skip.Exception:
            pass
        except AssertionError as e:
            bad.append((mode, extras, seed, str(e)[:200])); print("FAIL", mode, extras, seed, str(e)[:200], flush=True)
        except Exception as e:
            bad.append((mode, extras, seed, repr(e)[:200])); print("ERR", mode, extras, seed, repr(e)[:200], flush=True)
    if seed % 10 == 0: print("seed", seed, "fails", len(bad), flush=True)
print("DONE", len(bad))
