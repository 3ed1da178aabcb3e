#include "g_units.hpp"
void launch_g_emit(const GArgs& A, bool heavy, uint32_t nunits, hipStream_t st) {
    if (heavy) hipLaunchKernelGGL((g_units<EmitP, true>), dim3(nunits, 1), dim3(64), A.stage_lds ? sizeof(POS_TABLE_MONT) : 0, st, A);
    else hipLaunchKernelGGL((g_units<EmitP, false>), dim3(nunits, 1), dim3(64), 0, st, A);
}
