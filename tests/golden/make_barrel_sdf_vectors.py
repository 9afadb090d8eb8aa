#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/barrel_sdf_reference_vectors.npz by EXECUTING the reference's barrel-cylinder SDF
(`sdf_cylinder(point, radius, half_height, up_axis, barrel_radius)`, newton/_src/geometry/kernels.py:347-447) on the Warp stand-in
(tests/golden/refshim) for seeded points around three barrels; tests/test_sdf_texture.py compares newton_amd.sdf.primitive_sdf.
Run from the repo root:  python tests/golden/make_barrel_sdf_vectors.py"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "refshim"))
import lazy_ref  # noqa: E402

lazy_ref.install(dummies={"newton._src.sim": ("Contacts", "Control", "Model", "State", "ModelBuilder")},
                 dummy_modules=("newton._src.sim.builder",), f32_literals=("newton._src.geometry.kernels",))
import warp as wp  # noqa: E402  (the stand-in)

k = importlib.import_module("newton._src.geometry.kernels")
CASES = {"wide": (0.05, 0.08, 0.13), "tight": (0.05, 0.08, 0.08), "flat": (0.2, 0.05, 0.4)}


def main():
    rng = np.random.default_rng(17)
    blob = {}
    for name, (r, hh, br) in CASES.items():
        ext = r + br - (br * br - hh * hh) ** 0.5  # radial extent at the equator
        pts = rng.uniform(-1.0, 1.0, size=(400, 3)) * np.array([1.4 * ext, 1.4 * ext, 1.6 * hh])
        pts[:8] = [[0, 0, 0], [0, 0, hh], [r, 0, hh], [r + br - (br * br - hh * hh) ** 0.5, 0, 0], [0, 0, 2 * hh], [3 * r, 0, 0], [r, 0, -hh], [0, r, 0.5 * hh]]
        pts = pts.astype(np.float32)
        d = np.array([float(k.sdf_cylinder(wp.vec3(*[float(x) for x in p]), wp.float32(r), wp.float32(hh), 2, wp.float32(-1.0), wp.float32(br))) for p in pts], np.float32)
        blob[f"{name}/scale"], blob[f"{name}/points"], blob[f"{name}/distance"] = np.array([r, hh, br], np.float32), pts, d
        print(name, "inside", int((d < 0).sum()), "of", len(d))
    np.savez_compressed(os.path.join(HERE, "barrel_sdf_reference_vectors.npz"), **blob)


if __name__ == "__main__":
    main()
