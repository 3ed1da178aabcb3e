#include "poseidon_wide.hpp"
#include "keccak_kernels.hpp"
// nunits Poseidon blocks (U_POS_WIDE units from A.order[A.first..]) x ngroups: 8 wavefronts per (unit, group)
void launch_pos_wide(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st) {
    hipLaunchKernelGGL(k_poseidon_wide, dim3(nunits * 8, ngroups), dim3(64), POSW_LDS_BYTES, st, A);
}
// The Poseidon blocks and a sponge chain that does not depend on them in ONE launch: both are a few hundred wavefronts that walk a long serial chain each (304 Montgomery
// products; the header's 17 permutations), alone on their SIMDs -- as two launches of a calculator's stream the second waited 0.5 ms for the first with the machine idle.
// grid = (groups, 8 * npos + nsponges): the Poseidon wavefronts first (the longer chain).
__global__ void __launch_bounds__(64, 2) k_pos_chain(GArgs A, KArgs K, uint32_t npos8) {
    const uint32_t g = blockIdx.x, it = blockIdx.y;
    if (it < npos8) poswide_body(A, it, g);
    else chain_body<false>(K, it - npos8, g);
}
void launch_pos_chain(const GArgs& A, const KArgs& K, uint32_t npos, uint32_t nsponges, uint32_t ngroups, hipStream_t st) {
    hipLaunchKernelGGL(k_pos_chain, dim3(ngroups, npos * 8 + nsponges), dim3(64), POSW_LDS_BYTES, st, A, K, npos * 8);
}
