#include "g_units.hpp"
#include "keccak_kernels.hpp"
// The Keccak round evaluation (the batch's one bandwidth-bound kernel: 8 064 wavefronts that stream 1.77 GB) and the four WIDE evaluation families (RANGE, SELROW, LD, SC:
// 63 k short latency-bound wavefronts) in ONE launch, interleaved in dispatch order: nothing of the evaluation depends on anything else of it, and as launches of one stream
// they followed each other -- 0.31 ms of streaming with idle issue slots, then 0.2 ms of round trips with an idle memory system.  Every q-th position of the launch is a
// round-evaluation item, the positions between them the units (longest first), so both kinds are resident from the first wavefront to the last.
// grid = (groups, nk + nunits); 125 VGPRs, no scratch (the round evaluation's budget, 4 wavefronts per SIMD: its ring of prefetched arrays is 11 deep here -- 12 as in the kernel of
// its own needs one register more than the 128 and spills it; 8-24 ahead measured the same, profiles/round5_experiments.txt 1).
#define FAM_WIDE (FAM_BIT(F_RANGE) | FAM_BIT(F_SELROW) | FAM_BIT(F_LD) | FAM_BIT(F_SC))
__global__ void __launch_bounds__(64) POB_WAVES_PER_SIMD(4) k_check_wide(GArgs A, KArgs K, uint32_t nk, uint32_t q) {
    const uint32_t g = blockIdx.x, p = blockIdx.y;
    const uint32_t kq = p / q;
    if (nk && p % q == 0 && kq < nk) { POB_ROUNDS_CHECK_BODY(K, kq, g, true, POB_KCHK_ROUNDS, 11); }
    else {
        const uint32_t before = nk ? (p + q - 1) / q : 0;                 // round-evaluation items at positions < p
        g_units_body<CheckP, FAM_WIDE>(A, g, p - (before < nk ? before : nk));
    }
}
void launch_check_wide(const GArgs& A, const KArgs& K, uint32_t nunits, uint32_t nperms, uint32_t ngroups, hipStream_t st) {
    const uint32_t nk = nperms * (24 / POB_KCHK_ROUNDS), n = nk + nunits;
    const uint32_t q = nk ? (n / nk ? n / nk : 1) : 1;
    hipLaunchKernelGGL(k_check_wide, dim3(ngroups, n), dim3(64), 0, st, A, K, nk, q);
}
