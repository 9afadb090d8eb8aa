// TEST INFRASTRUCTURE ONLY -- CPU oracle: convex (MPR/GJK/manifold) pair path.
// Round-1 status: NOT YET RESTATED.  Pairs the reference routes to narrow_phase_kernel_gjk_mpr
// (newton/_src/geometry/narrow_phase.py:1040-1216: box-box, capsule-box, ...) produce no contacts
// here; tests that need them are marked xfail and DESIGN.md lists the row as open.
#include "oracle_common.h"
namespace orc {
int convex_pair_contacts(const o_model*, int, int, const float*, const float*, const float*, const float*, const float*,
                         o_contacts*) {
    return 0;
}
}  // namespace orc
