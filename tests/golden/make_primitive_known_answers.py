#!/usr/bin/env python
"""Extract the known-answer tables of the reference's own primitive-collision tests into a JSON fixture.

Source (read-only, only available in the build container): /root/reference/newton/tests/test_collision_primitives.py
(test_plane_sphere :429 ... test_plane_cylinder :1430).  Only the literal `test_cases` tables (inputs + expected
distances / contact counts hand-computed by the reference authors) are extracted -- no code is copied.
The reference cannot be *run* here (warp-lang is not installable), so these tables are the golden vectors that pin
the oracle (tests/test_oracle_known_answers.py).

usage: python tests/golden/make_primitive_known_answers.py   (rewrites tests/golden/primitive_known_answers.json)
"""
import ast
import json
import os

import numpy as np

SRC = "/root/reference/newton/tests/test_collision_primitives.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "primitive_known_answers.json")
TESTS = ["test_plane_sphere", "test_sphere_sphere", "test_sphere_capsule", "test_capsule_capsule", "test_plane_ellipsoid",
         "test_sphere_cylinder", "test_sphere_box", "test_plane_capsule", "test_plane_box", "test_plane_cylinder"]


class _WP:
    @staticmethod
    def mat33(*a):
        return [float(x) for x in (a[0] if len(a) == 1 else a)]

    @staticmethod
    def vec3(*a):
        return [float(x) for x in a]

    float32 = float


def to_jsonable(x):
    if isinstance(x, (list, tuple)):
        return [to_jsonable(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, float, int, np.integer)):
        return float(x) if not isinstance(x, (int, np.integer)) else int(x)
    if isinstance(x, bool):
        return bool(x)
    raise TypeError(type(x))


def main():
    tree = ast.parse(open(SRC).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in TESTS:
            stmts = []
            for st in node.body:
                stmts.append(st)
                if isinstance(st, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "test_cases" for t in st.targets):
                    break
            ns = {"np": np, "wp": _WP, "MAXVAL": 1e10}
            mod = ast.Module(body=[s for s in stmts if not (isinstance(s, ast.Expr) and isinstance(s.value, ast.Constant))],
                             type_ignores=[])
            exec(compile(mod, SRC, "exec"), ns)  # only literal assignments precede the table
            out[node.name] = to_jsonable(ns["test_cases"])
    json.dump({"source": "newton/tests/test_collision_primitives.py (reference checkout 2026-08-21)", "tables": out},
              open(OUT, "w"), indent=1)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
