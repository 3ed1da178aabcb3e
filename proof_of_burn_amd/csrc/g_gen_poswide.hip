#include "poseidon_wide.hpp"
// nunits Poseidon blocks (U_POS_WIDE units from A.order[A.first..]) x ngroups: 8 wavefronts per (unit, group)
void launch_pos_wide(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st) {
    hipLaunchKernelGGL(k_poseidon_wide, dim3(nunits * 8, ngroups), dim3(64), POSW_LDS_BYTES, st, A);
}
