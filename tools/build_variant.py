#!/usr/bin/env python
"""Build a throw-away measurement variant of libnewton_hip.so with extra -D flags (never the product library):
    python tools/build_variant.py build_ab/libnewton_ablation.so -DNT_ABLATION
    python tools/build_variant.py build_ab/libnewton_timing.so -DNT_PHASE_TIMING
Use it through tools/with_lib.py (the product loader has no override; the reassignment is announced on stderr).  Run here (hipcc cross-compiles) so that no GPU-minutes
go into compiling."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

out = os.path.abspath(sys.argv[1])
os.makedirs(os.path.dirname(out), exist_ok=True)
# --units a.hip,b.hip: only these translation units see the extra flags; the others are linked from the product's cached objects
only = None
extra = []
for a in sys.argv[2:]:
    if a.startswith("--units="):
        only = a.split("=", 1)[1].split(",")
    elif a == "--no-unit-flags":
        pass
    else:
        extra.append(a)
flags = [f for f in g.HIP_FLAGS] + extra
objs = []
tmp = os.path.join(os.path.dirname(out), "_variant_obj")
os.makedirs(tmp, exist_ok=True)
compile_flags = [f for f in flags if f != "-shared"] + ["-c"]
for u in g.UNITS:
    if only is None or u in only:
        obj = os.path.join(tmp, os.path.basename(out) + "." + u.replace(".hip", ".o"))
        uf = [] if "--no-unit-flags" in sys.argv else g.unit_flags(u)  # (the product's per-unit flags, unless the variant is about them)
        cmd = [g.HIPCC, *compile_flags, *uf, f'-DNT_BUILD_ID="variant{"".join(extra)}"', os.path.join(g.CSRC, u), "-o", obj]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    else:
        key = g.source_hash() if u == "nt_build_id.hip" else g._unit_hash(u)
        obj = os.path.join(g.OBJ_DIR, f"{u.replace('.hip', '')}.{key}.o")
        if not os.path.exists(obj):  # the product was not rebuilt since this unit (or, for nt_build_id.hip, any unit) changed
            base = [f for f in g.HIP_FLAGS if f != "-shared"] + ["-c"]
            cmd = [g.HIPCC, *base, *g.unit_flags(u), f'-DNT_BUILD_ID="{g.source_hash()}"', os.path.join(g.CSRC, u), "-o", obj]
            print(" ".join(cmd), flush=True)
            os.makedirs(g.OBJ_DIR, exist_ok=True)
            subprocess.run(cmd, check=True)
        objs.append(obj)
subprocess.run([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True)
print("built", out)
