// nt_step_preamble.hpp -- what every stepping translation unit (nt_kernels.hip: collide / XPBD / SemiImplicit; nt_featherstone.hip:
// SolverFeatherstone + eval_fk) starts with: the math helpers in both arithmetic namespaces, the LDS layout, the per-lane context and
// the phase headers, inside one anonymous namespace per unit.  Two units because their kernels want different compilers: the XPBD
// rollouts gain 5 % from LLVM's iterative ILP scheduling strategy, SolverFeatherstone's lose 13 % to it (profiles/r06E_ab_workloads.txt)
// -- and the strategy is a per-unit option (__graft_entry__.UNIT_FLAGS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/newton_hip.h"
#include "nt_math.hpp"
#include "nt_primitives.hpp"
#include "nt_convex.hpp"

// Two arithmetic namespaces in one translation unit (the fused kernels run both on the same LDS tile):
//  * ieee  -- nt:: helpers, -ffp-contract=off, correctly rounded division / sqrt: everything that decides pair sets, contact
//             counts and contact geometry (shape transforms, AABBs, broad phase, primitive / MPR-GJK narrow phase, the contact
//             writer), SolverSemiImplicit and SolverFeatherstone.  CollisionPipeline.collide stays bit-identical to the contact
//             arrays the reference's own kernels produce (tests/golden/collide_reference_vectors.npz).
//  * fused -- ntf:: helpers (a second copy of nt_math.hpp) + the XPBD phases of nt_xpbd.hpp under `#pragma clang fp
//             contract(fast)` and -DNT_XPBD_FAST_MATH: a * b + c contracts to v_fma_f32, the divisions / square roots of the
//             projection phases are v_rcp_f32 / v_sqrt_f32 (1 ulp).  Within SURVEY.md 8(c)'s contract (1e-5 single step, 1e-4
//             rollout against the reference); measured on the MI355X headline: 0.363 -> 0.308 ms per 10-substep launch for the whole
//             kernel (profiles/r04a_*).  clang attaches the contraction permission to an operation where it is WRITTEN, hence the
//             second copy of the helpers instead of a flag.
// The step and the rollout kernels share the fused phases, so a rollout stays bitwise equal to the call-by-call loop.
// -DNT_XPBD_IEEE (measurement builds only, tools/build_variant.py): the XPBD phases with IEEE arithmetic -- no contraction, correctly
// rounded division / square root, the literal rotation formulas -- to measure what the fast arithmetic contributes to the threshold
// events of an open-loop frame (tests/tolerances.py: explain_rollout_outliers).
#define NT_MATH_NS ntf
#ifndef NT_XPBD_IEEE
#pragma clang fp contract(fast)
#endif
#include "nt_math.hpp"
#pragma clang fp contract(off)
#undef NT_MATH_NS

namespace {

#include "nt_layout.hpp"

namespace ieee {
using namespace nt;
#include "nt_ctx.hpp"
#include "nt_collide.hpp"
#include "nt_xpbd.hpp"
#define NT_SI_PHASES_ONLY  // (the SolverSemiImplicit kernel itself follows below, once)
#include "nt_semi_implicit.hpp"
#undef NT_SI_PHASES_ONLY
#include "nt_featherstone.hpp"
}  // namespace ieee

#ifndef NT_XPBD_IEEE
#pragma clang fp contract(fast)
#define NT_XPBD_FAST_MATH
#endif
namespace fused {
using namespace ntf;
#include "nt_ctx.hpp"
#include "nt_xpbd.hpp"
}  // namespace fused
#undef NT_XPBD_FAST_MATH
#pragma clang fp contract(off)

namespace ieee {
#include "nt_xpbd_kernels.hpp"
#define NT_SI_KERNEL_ONLY
#include "nt_semi_implicit.hpp"
#undef NT_SI_KERNEL_ONLY
#include "nt_featherstone_kernels.hpp"
}  // namespace ieee
using namespace ieee;

constexpr size_t LDS_BYTES_PER_CU = 160 * 1024;

bool model_ok(const nt_model* m) {
    return m && m->env_count > 0 && m->env_stride >= m->env_count && (m->env_stride % 64) == 0 && m->nb > 0 &&
           (m->cpp == 4 || m->cpp == 5) && m->np_analytic >= 0 && m->np_analytic <= m->np &&
           (m->np_analytic == m->np || m->cpp == 5);
}

}  // namespace
