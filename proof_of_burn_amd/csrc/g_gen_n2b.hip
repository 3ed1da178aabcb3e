#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_gen_n2b, GenP, FAM_HEAVY, 4)
