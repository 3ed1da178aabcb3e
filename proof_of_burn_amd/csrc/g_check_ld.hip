#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_ld, CheckP, FAM_BIT(F_LD), 8)
