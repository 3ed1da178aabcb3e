#include "g_units.hpp"
POB_DEFINE_G_LAUNCH(launch_g_emit_light, EmitP, FAM_LIGHT, 4)
