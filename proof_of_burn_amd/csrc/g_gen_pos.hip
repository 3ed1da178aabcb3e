#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_gen_pos, GenP, FAM_HEAVY, 1, true)
