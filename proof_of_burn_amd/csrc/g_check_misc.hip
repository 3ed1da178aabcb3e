#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_misc, CheckP, FAM_BIT(F_MISC), 7)
