"""Where hydro_pairs_kernel<true> spends its cycles (measurement tool, GPU box): run through tools/with_lib.py on a library built
with -DNT_HYDRO_TIMING (tools/build_variant.py); same physics as the product (the build only adds timestamps)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402

import newton_amd as nt  # noqa: E402
from newton_amd import _lib  # noqa: E402


def main():
    envs, settle, frames = 256, 40, 3
    model = bench.build_shard("hydro_bin", envs, 0, 1, "cuda:0")
    faces_only = len(sys.argv) > 1 and sys.argv[1] == "faces"
    pipe = nt.CollisionPipeline(model, broad_phase="sap",
                                sdf_hydroelastic_config=nt.geometry.HydroelasticSDF.Config(reduce_contacts=not faces_only),
                                sdf_contacts_per_shape=400, sdf_hydro_faces_per_shape=1000)
    contacts = pipe.contacts()
    solver = nt.solvers.SolverXPBD(model, iterations=2)
    s0, s1, ctrl = model.state(), model.state(), model.control()
    dt = bench.WORKLOADS["hydro_bin"]["dt"]
    lib = _lib.load()
    lib.nt_hydro_timing_read.restype = C.c_int32
    lib.nt_hydro_timing_read.argtypes = [C.POINTER(C.c_ulonglong)]

    def read():
        torch.cuda.synchronize()
        out = (C.c_ulonglong * 16)()
        assert lib.nt_hydro_timing_read(out) == 0
        return [int(x) for x in out]

    def frame():
        nonlocal s0, s1
        for _ in range(bench.SUBSTEPS):
            s0.clear_forces()
            pipe.collide(s0, contacts)
            solver.step(s0, s1, ctrl, contacts, dt)
            s0, s1 = s1, s0

    for _ in range(settle):
        frame()
    a = read()
    for _ in range(frames):
        frame()
    b = read()
    d = [y - x for x, y in zip(a, b)]
    n = frames * bench.SUBSTEPS
    names = ["face pass (single kernel only: SAT, octree, marching cubes, records)", "reduce: aggregates", "reduce: table passes",
             "reduce: winners", "reduce: order + depth sums", "reduce: export", "staged reduce: chunk records of the pair",
             "staged reduce: rebase + fence + pair descriptors"]
    out = {"collides": n, "pairs_per_collide": d[8] / n, "active_pairs_per_collide": d[9] / n, "face_blocks_per_collide": d[10] / n,
           "cycles_per_collide": {names[i]: d[i] / n for i in range(8)}, "info": pipe._sdf_leg.overflow(contacts._flat)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
