// TEST INFRASTRUCTURE ONLY: the emulated library does not contain the standalone broad phases (their kernels use wave ballots);
// these stubs only satisfy the product loader's symbol check when the GPU test files are dry-run on the CPU (emu_plugin.py).
#include <stdint.h>
extern "C" {
int32_t nt_broadphase_nxn(const void*, const int32_t*, const int32_t*, int32_t, int32_t, int32_t, int32_t*, int32_t*, int32_t, void*) { return -3; }
int32_t nt_broadphase_sap(const void*, const int32_t*, const int32_t*, int32_t, int32_t, int32_t, int32_t*, int32_t*, int32_t, void*) { return -3; }
int32_t nt_broadphase_explicit(const void*, const int32_t*, int32_t, int32_t*, int32_t*, int32_t, void*) { return -3; }
}
