#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/fuzz_reference_vectors.npz: the reference's own SolverXPBD / SolverSemiImplicit /
SolverFeatherstone source (unmodified, under /root/reference) EXECUTED on tests/golden/refshim for the seeded random cases of
fuzz_reference_cases.py, teacher-forced step by step exactly like make_xpbd_reference_vectors.py (whose driver this script reuses).
Cases the reference or the host rejects (e.g. a D6 joint with several angular axes the host FK refuses) are recorded as skipped, with
the reason.  Run from the repo root:  python tests/golden/make_fuzz_reference_vectors.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_xpbd_reference_vectors as gen  # noqa: E402  (installs the shim, imports the reference modules)


def main():
    import fuzz_reference_cases as fc

    blob, skipped = {}, []
    for name, case in fc.cases().items():
        try:
            out = gen.run_case(name, case)
        except NotImplementedError as e:
            skipped.append(f"{name}: {e}")
            print("SKIP", name, e, flush=True)
            continue
        for k, v in out.items():
            blob[f"{name}/{k}"] = v
    blob["skipped"] = np.array(skipped, dtype=object).astype(str) if skipped else np.array([], dtype=str)
    np.savez_compressed(os.path.join(HERE, "fuzz_reference_vectors.npz"), **blob)
    print("wrote", len(blob) - 1, "arrays;", len(skipped), "cases skipped")


if __name__ == "__main__":
    main()
