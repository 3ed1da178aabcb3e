#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_gen_light, GenP, FAM_LIGHT, 8)
