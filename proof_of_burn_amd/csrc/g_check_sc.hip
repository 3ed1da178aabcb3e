#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_sc, CheckP, FAM_BIT(F_SC), 4)
