#!/usr/bin/env python
"""Register / scratch / LDS footprint of every kernel in a built library (code-object metadata, no GPU needed):
    python tools/kernel_resources.py [newton_amd/libnewton_hip.so] [name filter]
Columns: VGPR, AGPR, SGPR, spilled VGPR / SGPR, private (scratch) bytes, static LDS bytes."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "newton_amd", "libnewton_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={lib}"], capture_output=True)
    # the fat binary sits in .hip_fatbin: unbundle the gfx950 code object
    out = os.path.join(d, "co")
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={lib}", f"--output={out}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out):
        fb = os.path.join(d, "fb")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", f"--output={out}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
rows = []
for blk in notes.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
    agpr = blk.split("\n")[0].strip()
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if flt in name:
        rows.append((name, g("vgpr_count"), agpr, g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                     g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print(f"{'VGPR':>5}{'AGPR':>5}{'SGPR':>5}{'vspl':>5}{'sspl':>5}{'scr':>6}{'LDS':>7}  kernel")
for r in sorted(rows):
    print(f"{r[1]:>5}{r[2]:>5}{r[3]:>5}{r[4]:>5}{r[5]:>5}{r[6]:>6}{r[7]:>7}  {r[0][:150]}")
