// gadget-level mains (gadget_mains.hpp; reference tests/test.py:146-201): one kernel over the whole template list, not on the production path
#define POB_GM_KERNELS
#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_gm, GmPol<CheckP>, FAM_BIT(F_GM), 1)
