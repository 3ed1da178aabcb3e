#include <stdlib.h>
#define POB_KECCAK_TU
#include "keccak_kernels.hpp"
void launch_k_chain(const KArgs& K, bool check, uint32_t nsponges, uint32_t ngroups, hipStream_t st) {
    // check: nsponges is the number of PERMUTATIONS (local evaluation, one wavefront per (sponge, block))
    if (check) hipLaunchKernelGGL(k_chain_check, dim3(nsponges, ngroups), dim3(64), 0, st, K);
    else hipLaunchKernelGGL(k_chain, dim3(nsponges, ngroups), dim3(64), 0, st, K);
}
int pob_kchk_rounds() { return POB_KCHK_ROUNDS; }
void launch_k_rounds(const KArgs& K, bool check, uint32_t nperms, uint32_t ngroups, hipStream_t st) {
    // evaluation: POB_KCHK_ROUNDS (4) rounds per wavefront, 4 wavefronts per SIMD (124 VGPRs, no scratch), 12 arrays requested ahead.  Round 5's sweep, alone / in the step:
    // 3 / 4 / 6 / 8 rounds and 8-24 arrays ahead are within 3 % of each other; without the ring (the compiler's one load per wait) the same alone; 5 wavefronts per SIMD spill
    // and lose 40 %; a persistent grid of ONE wavefront per SIMD saturates HBM alone (0.261 ms) and is slower in the step (profiles/round5_experiments.txt 1-2).
    // generation: 8 rounds per wavefront (1 / 2 / 4 / 8: 0.329 / 0.302 / 0.283 / 0.261 ms alone: midRound[r0] is read once per chunk; experiments 3)
    if (check) hipLaunchKernelGGL((k_rounds_check<true, POB_KCHK_ROUNDS, 4, 12>), dim3(nperms * (24 / POB_KCHK_ROUNDS), ngroups), dim3(64), 0, st, K);
    else hipLaunchKernelGGL(k_rounds_gen<POB_KGEN_ROUNDS>, dim3(nperms * (24 / POB_KGEN_ROUNDS), ngroups), dim3(64), 0, st, K);
}
// generation + evaluation of the round blocks in one launch (k_rounds_gc): POB_KGC_ROUNDS rounds per wavefront, POB_KGC_DP arrays requested back and not yet compared
#ifndef POB_KGC_ROUNDS
#define POB_KGC_ROUNDS 8
#endif
#ifndef POB_KGC_WAVES
#define POB_KGC_WAVES 4
#endif
#ifndef POB_KGC_DP
#define POB_KGC_DP 4
#endif
#ifndef POB_KGC_NTM
#define POB_KGC_NTM false
#endif
int pob_kgc_rounds() { return POB_KGC_ROUNDS; }
void launch_k_rounds_gc(const KArgs& K, uint32_t nperms, uint32_t ngroups, bool fault, hipStream_t st) {
    if (fault) hipLaunchKernelGGL((k_rounds_gc<POB_KGC_ROUNDS, POB_KGC_DP, POB_KGC_WAVES, true, POB_KGC_NTM>), dim3(nperms * (24 / POB_KGC_ROUNDS), ngroups), dim3(64), 0, st, K);
    else hipLaunchKernelGGL((k_rounds_gc<POB_KGC_ROUNDS, POB_KGC_DP, POB_KGC_WAVES, false, POB_KGC_NTM>), dim3(nperms * (24 / POB_KGC_ROUNDS), ngroups), dim3(64), 0, st, K);
}
void launch_k_emit_bits(const u64* G, uint8_t* out, uint32_t wire_base, uint32_t bit_base, uint32_t count, uint32_t sel, hipStream_t st) {
    uint32_t blocks = (count + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_emit_bits, dim3(blocks), dim3(256), 0, st, G, out, wire_base, bit_base, count, sel);
}
void launch_k_emit_bits_red(const u64* G, uint8_t* out, uint32_t wire0, uint32_t bit_base, uint32_t count, uint32_t sel, const unsigned long long* rbits, const uint32_t* rpre,
                            uint32_t k0, uint32_t kn, hipStream_t st) {
    uint32_t blocks = (count + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_emit_bits_red, dim3(blocks), dim3(256), 0, st, G, out, wire0, bit_base, count, sel, rbits, rpre, k0, kn);
}
void launch_k_emit_absorb(const u64* G, uint8_t* out, uint32_t ab, uint32_t o0, uint32_t count, uint32_t sel, const uint16_t* tab, hipStream_t st) {
    uint32_t blocks = (count + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_emit_absorb, dim3(blocks), dim3(256), 0, st, G, out, ab, o0, count, sel, tab);
}
void launch_k_emit_absorb_red(const u64* G, uint8_t* out, uint32_t wire0, uint32_t ab, uint32_t o0, uint32_t count, uint32_t sel, const uint16_t* tab, const unsigned long long* rbits,
                              const uint32_t* rpre, uint32_t k0, uint32_t kn, hipStream_t st) {
    uint32_t blocks = (count + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_emit_absorb_red, dim3(blocks), dim3(256), 0, st, G, out, wire0, ab, o0, count, sel, tab, rbits, rpre, k0, kn);
}
bool keccak_alias_table_host(uint16_t* tab) { return keccak_round_alias_table(tab); }
