#include "g_units.hpp"
// every generation unit but the lane-spread Poseidon blocks and the gadget mains in ONE kernel: what an in-order calculator launches per LEVEL, so that a level's narrow BN254 /
// SubstringCheck units and its wide light units overlap instead of following each other on the calculator's one stream (pob_host.hip).
// (Round 6 let slices of the round expansion ride along with the levels behind the sponge chains -- bandwidth-bound wavefronts beside latency-bound ones in one launch: a
//  level then lasts about as much longer as its slice would have taken by itself, 0.20 of the 0.25 ms came back; profiles/round6_experiments.txt.  Removed.)
POB_DEFINE_G_LAUNCH(launch_g_gen_all, GenP, FAM_LIGHT | FAM_HEAVY | FAM_BIT(F_SC), 4)
