#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_pos, CheckP, FAM_BIT(F_POS), 4)
