#include "g_units.hpp"
// the four NARROW evaluation families (MISC, RL, POS, N2B: 219 units of the production circuit, 58 + 15 + 75 + 71, each family a launch whose duration is its longest unit)
// in one kernel: an in-order calculator launches them together, so that their long poles overlap.  Register budget: profiles/round6_spill_table.txt.
// (3 or 4 wavefronts per SIMD -- the transposed decompositions made the 128-VGPR build possible: 7 spilled registers instead of 289 -- measured: the whole evaluation alone
//  0.82 -> 0.94 ms, the loop unchanged: the kernel lasts as long as its longest chain of Montgomery products, which more resident wavefronts only slow down.)
#ifndef POB_NARROW_WAVES
#define POB_NARROW_WAVES 2
#endif
POB_DEFINE_G_LAUNCH(launch_g_check_narrow, CheckP, FAM_BIT(F_MISC) | FAM_BIT(F_RL) | FAM_BIT(F_POS) | FAM_BIT(F_N2B), POB_NARROW_WAVES)
