// nt_build_id.hip -- nt_build_info(): names the build a measurement was taken on.  Its own translation unit, so that the source
// hash (kernel sources + header + compiler flags, __graft_entry__.source_hash) can change without recompiling the kernels.
#include "../../include/newton_hip.h"

#ifndef NT_BUILD_ID
#define NT_BUILD_ID "unknown"
#endif

extern "C" const char* nt_build_info(void) { return "libnewton_hip gfx950 (CDNA4) fp32, -ffp-contract=off, src " NT_BUILD_ID; }
