#include "g_units.hpp"
void launch_g_emit(const GArgs& A, int cls, uint32_t nunits, hipStream_t st) {
    if (cls == 1) hipLaunchKernelGGL((g_units<EmitP, 1>), dim3(nunits, 1), dim3(64), A.stage_lds ? sizeof(POS_TABLE_MONT) : 0, st, A);
    else if (cls == 2) hipLaunchKernelGGL((g_units<EmitP, 2>), dim3(nunits, 1), dim3(64), 0, st, A);
    else hipLaunchKernelGGL((g_units<EmitP, 0>), dim3(nunits, 1), dim3(64), 0, st, A);
}
