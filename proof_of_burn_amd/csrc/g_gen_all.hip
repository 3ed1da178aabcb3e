#include "g_units.hpp"
// every generation unit but the lane-spread Poseidon blocks and the gadget mains in ONE kernel (100 VGPRs, no spills): what an in-order calculator launches per LEVEL, so
// that a level's narrow BN254 / SubstringCheck launches and its wide light launch overlap instead of following each other on the calculator's one stream (pob_host.hip)
POB_DEFINE_G_LAUNCH(launch_g_gen_all, GenP, FAM_LIGHT | FAM_HEAVY | FAM_BIT(F_SC), 4)
