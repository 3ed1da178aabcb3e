#include "g_units.hpp"
// g_gen_all.hip's kernel with the evaluation riding with the generation (policy.hpp GenPT<true>: every wire a unit stores is loaded back and compared): what an in-order
// calculator with pob_set_inorder bit 2 launches per level.  The RLP units run on the plain policy (circuits.hpp unit_run_ride).
POB_DEFINE_G_LAUNCH(launch_g_gen_all_ride, GenRideP, FAM_LIGHT | FAM_HEAVY | FAM_BIT(F_SC), 4)
POB_DEFINE_G_LAUNCH(launch_g_gen_all_ride_fault, GenRideFaultP, FAM_LIGHT | FAM_HEAVY | FAM_BIT(F_SC), 4)      // (tests: pob_debug_store_fault)
