#include "g_units.hpp"
void launch_g_gen_light(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st) {
    hipLaunchKernelGGL((g_units<GenP, 0>), dim3(nunits, ngroups), dim3(64), A.stage_lds ? sizeof(POS_TABLE_MONT) : 0, st, A);
}
