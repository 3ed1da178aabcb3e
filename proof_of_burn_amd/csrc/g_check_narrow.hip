#include "g_units.hpp"
#include "keccak_kernels.hpp"
// the four NARROW evaluation families (MISC, RL, POS, N2B: 219 units of the production circuit, 58 + 15 + 75 + 71, each family a launch whose duration is its longest unit)
// in one kernel: an in-order calculator launches them together, so that their long poles overlap.  Register budget: profiles/round6_spill_table.txt.
// The launch can also carry the sponge-chain evaluation (one wavefront per permutation: 36 VGPRs, 1 344 wavefronts of coalesced loads) behind its units: the narrow
// families are a few thousand wavefronts whose duration is their longest unit's, the chain evaluation fills the SIMDs they leave idle.  grid = (groups, nunits + nchain)
#ifndef POB_NARROW_WAVES
#define POB_NARROW_WAVES 2
#endif
#define FAM_NARROW (FAM_BIT(F_MISC) | FAM_BIT(F_RL) | FAM_BIT(F_POS) | FAM_BIT(F_N2B))
__global__ void __launch_bounds__(64, POB_NARROW_WAVES) k_check_narrow(GArgs A, KArgs K, uint32_t nunits) {
    const uint32_t g = blockIdx.x, it = blockIdx.y;
    if (it < nunits) g_units_body<CheckP, FAM_NARROW>(A, g, it);
    else chain_check_body(K, it - nunits, g);
}
void launch_g_check_narrow(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st) {
    KArgs K; memset(&K, 0, sizeof K);
    hipLaunchKernelGGL(k_check_narrow, dim3(ngroups, nunits), dim3(64), 0, st, A, K, nunits);
}
void launch_check_narrow_chain(const GArgs& A, const KArgs& K, uint32_t nunits, uint32_t nperms, uint32_t ngroups, hipStream_t st) {
    hipLaunchKernelGGL(k_check_narrow, dim3(ngroups, nunits + nperms), dim3(64), 0, st, A, K, nunits);
}
