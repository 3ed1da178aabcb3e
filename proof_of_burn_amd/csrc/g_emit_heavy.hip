#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_emit_heavy, EmitP, FAM_HEAVY, 2)
POB_DEFINE_G_LAUNCH(launch_g_emit_sc, EmitP, FAM_BIT(F_SC), 4)
