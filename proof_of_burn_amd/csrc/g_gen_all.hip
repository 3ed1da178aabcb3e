#include "g_units.hpp"
#include "keccak_kernels.hpp"
// every generation unit but the lane-spread Poseidon blocks and the gadget mains in ONE kernel: what an in-order calculator launches per LEVEL, so that a level's narrow BN254 /
// SubstringCheck units and its wide light units overlap instead of following each other on the calculator's one stream (pob_host.hip).
// The launch may also carry a SLICE of the round expansion (k_rounds_gen's wavefronts for permutations [K.first, K.first + nk / chunks)): nothing of the generation reads a
// round block, so the expansion of a sponge whose chain is done rides along with the levels that follow it -- bandwidth-bound wavefronts beside latency-bound ones in one
// launch -- instead of being a launch of its own at the end of the calculator's stream.  grid = (groups, nunits + nk): the units first (longest first), the slice behind them.
__global__ void __launch_bounds__(64, 4) k_gen_level(GArgs A, KArgs K, uint32_t nunits) {
    const uint32_t g = blockIdx.x, it = blockIdx.y;
    if (it < nunits) g_units_body<GenP, FAM_LIGHT | FAM_HEAVY | FAM_BIT(F_SC)>(A, g, it);
    else rounds_gen_body<POB_KGEN_ROUNDS>(K, it - nunits, g);
}
void launch_g_gen_all(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st) {
    KArgs K; memset(&K, 0, sizeof K);
    hipLaunchKernelGGL(k_gen_level, dim3(ngroups, nunits), dim3(64), 0, st, A, K, nunits);
}
// nperms permutations from K.first ride along
void launch_gen_level(const GArgs& A, const KArgs& K, uint32_t nunits, uint32_t nperms, uint32_t ngroups, hipStream_t st) {
    const uint32_t nk = nperms * (24 / POB_KGEN_ROUNDS);
    hipLaunchKernelGGL(k_gen_level, dim3(ngroups, nunits + nk), dim3(64), 0, st, A, K, nunits);
}
