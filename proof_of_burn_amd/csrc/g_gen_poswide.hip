#include "poseidon_wide.hpp"
#include "keccak_kernels.hpp"
// nunits Poseidon blocks (U_POS_WIDE units from A.order[A.first..]) x ngroups: 8 wavefronts per (unit, group), POSW_WAVES per workgroup
void launch_pos_wide(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st, bool ride, bool fault) {
    const dim3 grid(nunits * (8 / POSW_WAVES), ngroups), block(64 * POSW_WAVES);
    if (ride && fault) hipLaunchKernelGGL((k_poseidon_wide<true, true>), grid, block, POSW_LDS_BYTES, st, A);
    else if (ride) hipLaunchKernelGGL((k_poseidon_wide<true, false>), grid, block, POSW_LDS_BYTES, st, A);
    else hipLaunchKernelGGL((k_poseidon_wide<false, false>), grid, block, POSW_LDS_BYTES, st, A);
}
// The Poseidon blocks and a sponge chain that does not depend on them in ONE launch: both are a few hundred wavefronts that walk a long serial chain each (304 Montgomery
// products; the header's 17 permutations), alone on their SIMDs -- as two launches of a calculator's stream the second waited 0.5 ms for the first with the machine idle.
// grid = (groups, workgroups of POSW_WAVES wavefronts): the Poseidon workgroups first (the longer chain), then the sponges, POSW_WAVES per workgroup.
template <bool RIDE, bool FAULT> __global__ void __launch_bounds__(64 * POSW_WAVES) k_pos_chain(GArgs A, KArgs K, uint32_t npos_wg, uint32_t nsponges) {
    const uint32_t g = blockIdx.x, it = blockIdx.y, w = threadIdx.x >> 6;
    if (it < npos_wg) poswide_body<RIDE, FAULT>(A, POSW_WAVES * it + w, g);
    else { const uint32_t sp = POSW_WAVES * (it - npos_wg) + w; if (sp < nsponges) chain_body(K, sp, g); }
}
void launch_pos_chain(const GArgs& A, const KArgs& K, uint32_t npos, uint32_t nsponges, uint32_t ngroups, hipStream_t st, bool ride, bool fault) {
    const uint32_t npos_wg = npos * (8 / POSW_WAVES);
    const dim3 grid(ngroups, npos_wg + (nsponges + POSW_WAVES - 1) / POSW_WAVES), block(64 * POSW_WAVES);
    if (ride && fault) hipLaunchKernelGGL((k_pos_chain<true, true>), grid, block, POSW_LDS_BYTES, st, A, K, npos_wg, nsponges);
    else if (ride) hipLaunchKernelGGL((k_pos_chain<true, false>), grid, block, POSW_LDS_BYTES, st, A, K, npos_wg, nsponges);
    else hipLaunchKernelGGL((k_pos_chain<false, false>), grid, block, POSW_LDS_BYTES, st, A, K, npos_wg, nsponges);
}
