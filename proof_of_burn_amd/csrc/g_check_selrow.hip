#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_check_selrow, CheckP, FAM_BIT(F_SELROW), 8)
