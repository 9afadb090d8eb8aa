"""TEST INFRASTRUCTURE ONLY -- builds tests/emu/_build/libnewton_emu.so: the gfx950 kernel sources of newton_amd/csrc compiled
for the host against tests/emu/hip_emu.h (one OS thread per GPU thread, std::barrier for __syncthreads).

The product sources are not modified: they are copied into the build directory with three textual substitutions
(HIP runtime include -> the shim, the dynamic-LDS declaration -> the shim's buffer, the wave-level fence of the Featherstone
Cholesky -> a rendezvous of the participating lanes).  The wave ballots / shuffles of nt_broadphase.hip are shim functions.
Used by tests/test_emu_*.py to run the kernels' logic against the oracle without a GPU; never loadable from newton_amd."""
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "newton_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libnewton_emu.so")
FILES = ["nt_step_preamble.hpp", "nt_featherstone.hip", "nt_math.hpp", "nt_primitives.hpp", "nt_convex.hpp", "nt_layout.hpp", "nt_ctx.hpp", "nt_collide.hpp", "nt_xpbd.hpp", "nt_xpbd_kernels.hpp",
         "nt_semi_implicit.hpp", "nt_featherstone.hpp", "nt_featherstone_kernels.hpp", "nt_kernels.hip", "nt_broadphase_core.hpp", "nt_broadphase.hip", "nt_contact_reduce.hpp",
         "nt_sdf.hip", "nt_build_id.hip", "nt_match.hip", "nt_model_build.hip", "nt_flat_contacts.hip", "nt_sdf_pipeline.hip", "nt_graph.hip", "nt_mesh_plane.hip", "nt_mesh_triangle.hip"]

WAVE_SYNC = re.compile(r"#define FS_WAVE_SYNC\(\)\s*\\\n(?:.*\\\n)*.*while \(0\)")
HY_SYNC = re.compile(r"#define HY_WAVE_SYNC_HW\(\)\s*\\\n(?:.*\\\n)*.*while \(0\)")
# (EPB is the tile code: environments per workgroup in its low byte, the uniform-parameter flag in bit 8)
WAVE_SYNC_EMU = ("#define FS_WAVE_SYNC() emu_wave_sync((unsigned)(G * (((int)c.a.m.env_count - (int)blockIdx.x * (EPB & 255)) < (EPB & 255) ? "
                 "((int)c.a.m.env_count - (int)blockIdx.x * (EPB & 255)) : (EPB & 255))))")


def transform(text: str) -> str:
    text = text.replace("#include <hip/hip_runtime.h>", '#include "hip_emu.h"')
    text = text.replace("extern __shared__ __align__(16) float lds[];", "float* lds = emu::dynamic_lds();")
    text = text.replace('#include "../../include/newton_hip.h"', f'#include "{os.path.join(ROOT, "include", "newton_hip.h")}"')
    for ext in ("broadphase", "mesh"):
        text = text.replace(f'#include "../../include/newton_hip_{ext}.h"', f'#include "{os.path.join(ROOT, "include", f"newton_hip_{ext}.h")}"')
    text, n = WAVE_SYNC.subn(WAVE_SYNC_EMU, text)
    text = HY_SYNC.sub("#define HY_WAVE_SYNC_HW() emu_wave_sync(64)", text)  # wave-level LDS ordering of the staged hydroelastic kernels
    # only under -DNT_XPBD_FAST_MATH (a measurement variant, never built here)
    text = text.replace("__builtin_amdgcn_rcpf", "emu_rcpf").replace("__builtin_amdgcn_sqrtf", "sqrtf")
    text = text.replace("__builtin_amdgcn_readfirstlane", "emu_uniform")  # a wave-uniform value: itself
    return text


def build(force: bool = False) -> str:
    """(serialised by a file lock: pytest-xdist workers that find a stale library would otherwise rebuild it at the same time and load
    each other's half-written output)"""
    import fcntl

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool) -> str:
    files = [f for f in FILES if os.path.exists(os.path.join(CSRC, f))]  # (tools/emu_bitcheck.py builds older revisions too)
    srcs = [os.path.join(CSRC, f) for f in files] + [os.path.join(HERE, "hip_emu.h"),
                                                     os.path.abspath(__file__),
                                                     os.path.join(ROOT, "include", "newton_hip.h"),
                                                     os.path.join(ROOT, "include", "newton_hip_broadphase.h"),
                                                     os.path.join(ROOT, "include", "newton_hip_mesh.h")]
    digest = hashlib.sha1(b"".join(open(s, "rb").read() for s in srcs)).hexdigest()
    stamp = os.path.join(OUT, "stamp")
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    for f in files:
        text = transform(open(os.path.join(CSRC, f)).read()).replace('#include "nt_broadphase_core.hpp"', '#include "nt_broadphase_core.hpp"')
        assert "hip_runtime" not in text and "__builtin_amdgcn" not in text, f
        open(os.path.join(OUT, f.replace(".hip", ".cpp")), "w").write(text)
    cmd = ["g++", "-std=c++20", "-O1", "-DNT_ALL_SHAPES", "-DNT_EMULATED_GRID=4", "-DNT_POISON_LDS", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-pthread", "-w",
           # the two stepping units (nt_kernels / nt_featherstone) both carry the plain (non-inline) __device__ helpers of nt_convex.hpp: on
           # the device each code object has its own copy, on the host they are identical external definitions
           "-Wl,--allow-multiple-definition",
           f"-I{HERE}", os.path.join(OUT, "nt_kernels.cpp"), os.path.join(OUT, "nt_broadphase.cpp"),
           *([os.path.join(OUT, "nt_featherstone.cpp")] if "nt_featherstone.hip" in files else []),
           *([os.path.join(OUT, "nt_sdf.cpp")] if "nt_sdf.hip" in files else []),
           *([os.path.join(OUT, "nt_build_id.cpp")] if "nt_build_id.hip" in files else []),
           *([os.path.join(OUT, "nt_match.cpp")] if "nt_match.hip" in files else []),
           *([os.path.join(OUT, "nt_model_build.cpp")] if "nt_model_build.hip" in files else []),
           *([os.path.join(OUT, "nt_graph.cpp")] if "nt_graph.hip" in files else []),
           *([os.path.join(OUT, "nt_flat_contacts.cpp")] if "nt_flat_contacts.hip" in files else []),
           *([os.path.join(OUT, "nt_sdf_pipeline.cpp")] if "nt_sdf_pipeline.hip" in files else []),
           *([os.path.join(OUT, "nt_mesh_plane.cpp")] if "nt_mesh_plane.hip" in files else []),
           *([os.path.join(OUT, "nt_mesh_triangle.cpp")] if "nt_mesh_triangle.hip" in files else []), "-o", LIB]
    subprocess.run(cmd, check=True)
    open(stamp, "w").write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
