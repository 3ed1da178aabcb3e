#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_range, CheckP, FAM_BIT(F_RANGE), 6)
