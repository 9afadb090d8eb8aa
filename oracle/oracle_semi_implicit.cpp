// TEST INFRASTRUCTURE ONLY -- CPU oracle: SolverSemiImplicit rigid path (placeholder until restated).
#include "oracle_common.h"
extern "C" void o_semi_implicit_step(const o_model*, const o_semi_implicit_params*, o_state*, o_state*, const o_control*,
                                     const o_contacts*, float) {}
