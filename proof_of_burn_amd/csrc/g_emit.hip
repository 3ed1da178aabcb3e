#include "g_units.hpp"
void launch_g_emit(const GArgs& A, uint32_t nunits, hipStream_t st) {
    hipLaunchKernelGGL(g_units<EmitP>, dim3(nunits, 1), dim3(64), A.stage_lds ? sizeof(POS_TABLE_MONT) : 0, st, A);
}
