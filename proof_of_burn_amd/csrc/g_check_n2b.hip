#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_n2b, CheckP, FAM_BIT(F_N2B), 2)
