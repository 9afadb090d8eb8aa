"""Shape-pair scenes for the hydroelastic pipeline: shared by make_hydro_reference_vectors.py (executes the reference on them) and
tests/test_hydro_reference_vectors.py (checker + kernels against the record)."""
import numpy as np

from newton_amd import sdf as S
from newton_amd.enums import GeoType


def scenes():
    """name -> dict(pairs, X [S,7], data [S,4] (scale, margin), gap [S], kh [S], sdfs [S] (TextureSDF per shape))."""
    out = {}
    sph = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.5, 0.5, 0.5), max_resolution=16, margin=0.05, scale_baked=True)
    sph_fine = S.create_texture_sdf_from_primitive(GeoType.SPHERE, (0.3, 0.3, 0.3), max_resolution=24, margin=0.04, scale_baked=True)
    box = S.create_texture_sdf_from_primitive(GeoType.BOX, (0.5, 0.4, 0.3), max_resolution=16, margin=0.05, scale_baked=True)
    q = np.array([0.1, -0.2, 0.15, 0.0])
    q[3] = np.sqrt(1.0 - np.sum(q[:3] ** 2))
    out["two_spheres"] = dict(pairs=np.array([[0, 1]], np.int32),
                              X=np.array([[0, 0, 0, 0, 0, 0, 1], [0.05, -0.04, 0.93, *q]], np.float32),
                              data=np.array([[1, 1, 1, 0.0]] * 2, np.float32), gap=np.array([0.01, 0.01], np.float32),
                              kh=np.array([1.0e8, 1.0e8], np.float32), sdfs=[sph, sph])
    out["sphere_in_box_margins"] = dict(pairs=np.array([[0, 1], [0, 2]], np.int32),
                                        X=np.array([[0, 0, 0, 0, 0, 0, 1], [0.1, 0.05, 0.55, *q], [3.0, 0, 0, 0, 0, 0, 1]], np.float32),
                                        data=np.array([[1, 1, 1, 0.004], [1, 1, 1, 0.002], [1, 1, 1, 0.0]], np.float32),
                                        gap=np.array([0.02, 0.01, 0.01], np.float32), kh=np.array([2.0e7, 1.0e8, 1.0e8], np.float32),
                                        sdfs=[box, sph_fine, sph_fine])
    for sc in out.values():  # tables the hydroelastic reduction reads: local AABBs (the SDF boxes here) + voxel grids
        sc["aabb_lo"] = np.array([t.box_lower for t in sc["sdfs"]], np.float32)
        sc["aabb_hi"] = np.array([t.box_upper for t in sc["sdfs"]], np.float32)
        sc["res"] = np.array([[5, 4, 3]] * len(sc["sdfs"]), np.int32)
    return out
