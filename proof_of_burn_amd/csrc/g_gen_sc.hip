#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_gen_sc, GenP, FAM_BIT(F_SC), 4)
