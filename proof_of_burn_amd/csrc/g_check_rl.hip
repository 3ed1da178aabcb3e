#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_rl, CheckP, FAM_BIT(F_RL), 5)
