#include "g_units.hpp"
// The BN254 units that are NOT Poseidon chains (Num2Bits_strict / byte conversions / range checks), compiled for <= 128 VGPRs:
// while the Keccak expansion runs, every SIMD holds four of its 128-VGPR waves, and only a wave that fits the slot one of them
// frees can start beside it -- the 485-VGPR build of these units waited for the whole Keccak grid to drain.
void launch_g_gen_heavy_small(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st) {
    hipLaunchKernelGGL((g_units<GenP, 3>), dim3(nunits, ngroups), dim3(64), 0, st, A);
}
