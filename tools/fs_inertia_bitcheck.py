#!/usr/bin/env python
"""TEST TOOL -- bit check of fs_transform_spatial_inertia (newton_amd/csrc/nt_featherstone.hpp): the shipped function sums only the
terms of T^T I T whose factor is not a structural zero; this tool compiles it for the host next to the dense 6 x 6 products the
reference performs (transform_spatial_inertia, newton/_src/solvers/featherstone/kernels.py:66-139) and compares 2 000 000 random
and degenerate inputs bit for bit (identity / axis-aligned rotations, zero offsets, zero mass, diagonal and zero inertia).
    python tools/fs_inertia_bitcheck.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "newton_amd", "csrc")
DENSE = r"""
NT_DI void dense_tsi(const xform& t, float mass, const mat33& Ib, mat66& out) {
    xform t_inv = xform_inverse(t);
    quat q = t_inv.q;
    vec3 p = t_inv.p;
    vec3 r1 = quat_rotate(q, vec3(1.0f, 0.0f, 0.0f));
    vec3 r2 = quat_rotate(q, vec3(0.0f, 1.0f, 0.0f));
    vec3 r3 = quat_rotate(q, vec3(0.0f, 0.0f, 1.0f));
    float R[3][3] = {{r1.x, r2.x, r3.x}, {r1.y, r2.y, r3.y}, {r1.z, r2.z, r3.z}};
    float K[3][3] = {{0.0f, -p.z, p.y}, {p.z, 0.0f, -p.x}, {-p.y, p.x, 0.0f}};
    float S[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) sum += K[i][k] * R[k][j];
            S[i][j] = sum;
        }
    mat66 T, I;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            T.a[i][j] = 0.0f;
            I.a[i][j] = 0.0f;
        }
    const float Im[3][3] = {{Ib.m00, Ib.m01, Ib.m02}, {Ib.m10, Ib.m11, Ib.m12}, {Ib.m20, Ib.m21, Ib.m22}};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        I.a[i][i] = mass;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T.a[i][j] = R[i][j];
            T.a[i][j + 3] = S[i][j];
            T.a[i + 3][j + 3] = R[i][j];
            I.a[i + 3][j + 3] = Im[i][j];
        }
    }
    mat66 A;  // T^T I
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) sum += T.a[k][i] * I.a[k][j];
            A.a[i][j] = sum;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) sum += A.a[i][k] * T.a[k][j];
            out.a[i][j] = sum;
        }
}

"""
MAIN = r"""
int main() {
    std::mt19937 rng(7);
    std::normal_distribution<float> N(0.f, 1.f);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    long bad = 0, total = 0;
    for (int it = 0; it < 2000000; ++it) {
        quat q(N(rng), N(rng), N(rng), N(rng));
        q = normalize(q);
        if (it % 7 == 0) q = quat(0, 0, 0, 1);
        if (it % 11 == 0) q = quat(1, 0, 0, 0);
        vec3 p(N(rng), N(rng), N(rng));
        if (it % 13 == 0) p = vec3(0, 0, 0);
        if (it % 17 == 0) p = vec3(0.5f, 0, 0);
        float m = U(rng) * 10.f;
        if (it % 19 == 0) m = 0.f;
        float a = N(rng), b = N(rng), c = N(rng), d = N(rng), e = N(rng), f = N(rng);
        mat33 I(a * a + 1, b * 0.1f, c * 0.1f, b * 0.1f, d * d + 1, e * 0.1f, c * 0.1f, e * 0.1f, f * f + 1);
        if (it % 5 == 0) I = mat33(a * a, 0, 0, 0, d * d, 0, 0, 0, f * f);
        if (it % 23 == 0) I = mat33(0, 0, 0, 0, 0, 0, 0, 0, 0);
        mat66 A, B;
        dense_tsi(xform(p, q), m, I, A);
        fs_transform_spatial_inertia(xform(p, q), m, I, B);
        total++;
        if (std::memcmp(&A, &B, sizeof(A)) != 0) bad++;
    }
    printf("checked %ld inputs, %ld differ\n", total, bad);
    return bad != 0;
}
"""


def main():
    fs = open(os.path.join(CSRC, "nt_featherstone.hpp")).read()
    a = fs.index("NT_DI void fs_transform_spatial_inertia")
    b = fs.index("// compute_link_velocity (kernels.py:764-866)")
    with tempfile.TemporaryDirectory() as tmp:
        math = open(os.path.join(CSRC, "nt_math.hpp")).read().replace("#include <hip/hip_runtime.h>", '#include "hip_emu.h"')
        open(os.path.join(tmp, "nt_math.hpp"), "w").write(math)
        src = ('#include "hip_emu.h"\n#include <cstdio>\n#include <cstring>\n#include <random>\n#include "nt_math.hpp"\nusing namespace nt;\n'
               "struct mat66 { float a[6][6]; };\n" + DENSE + fs[a:b] + MAIN)
        open(os.path.join(tmp, "chk.cpp"), "w").write(src)
        exe = os.path.join(tmp, "chk")
        subprocess.run(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-fno-fast-math", f"-I{tmp}", f"-I{os.path.join(ROOT, 'tests', 'emu')}",
                        "-pthread", "-w", os.path.join(tmp, "chk.cpp"), "-o", exe], check=True)
        return subprocess.run([exe]).returncode


if __name__ == "__main__":
    sys.exit(main())
