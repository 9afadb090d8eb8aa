#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- records what the reference's own `create_mesh_box` (newton/_src/utils/mesh.py:2034-2131, behind
newton.Mesh.create_box) returns for duplicate_vertices=True / False into tests/golden/mesh_box_tables.json: vertex positions and
triangle indices.  The vertex ORDER is part of the behaviour since triangle meshes collide vertex by vertex (the vertex index is
the contact fingerprint).  Run from the repo root:  python tests/golden/make_mesh_box_tables.py"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "refshim"))
import lazy_ref  # noqa: E402

lazy_ref.install()
mesh = importlib.import_module("newton._src.utils.mesh")

out = {}
for dup in (True, False):
    p, i, _, _ = mesh.create_mesh_box(0.5, 0.25, 0.125, duplicate_vertices=dup, compute_normals=False, compute_uvs=False)
    out["duplicated" if dup else "shared"] = {"positions": [[float(x) for x in v] for v in p], "indices": [int(x) for x in i]}
# ... and create_mesh_sphere (utils/mesh.py:1026-1075, behind newton.Mesh.create_sphere): the (lat + 1) x (lon + 1) grid, y up,
# seam and poles repeated
p, i, _, _ = mesh.create_mesh_sphere(0.5, num_latitudes=4, num_longitudes=6, compute_normals=False, compute_uvs=False)
out["sphere_4x6"] = {"positions": [[float(x) for x in v] for v in p], "indices": [int(x) for x in i]}
json.dump(out, open(os.path.join(HERE, "mesh_box_tables.json"), "w"), indent=1)
print({k: (len(v["positions"]), len(v["indices"])) for k, v in out.items()})
