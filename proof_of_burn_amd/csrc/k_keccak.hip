#include <stdlib.h>
#include "keccak_kernels.hpp"
void launch_k_chain(const KArgs& K, bool check, uint32_t nsponges, uint32_t ngroups, hipStream_t st) {
    // check: nsponges is the number of PERMUTATIONS (local evaluation, one wavefront per (sponge, block))
    if (check) hipLaunchKernelGGL(k_chain_check, dim3(nsponges, ngroups), dim3(64), 0, st, K);
    else hipLaunchKernelGGL(k_chain<false>, dim3(nsponges, ngroups), dim3(64), 0, st, K);
}
void launch_k_rounds(const KArgs& K, bool check, uint32_t nperms, uint32_t ngroups, hipStream_t st) {
    // (non-temporal loads in the evaluation: 4.35 -> 4.09 ms per launch at batch 1024, 0.785 -> 0.836 of the HBM peak)
    // (compiled for 3 / 4 waves per SIMD -- 168 VGPRs + 64 B scratch / 128 + 152 B -- it is slower alone (4.4-4.7 / 4.6-5.2 ms vs 4.09) and in the step
    //  (13.3-13.7 / 13.7-14.0 ms vs 13.15-13.2): profiles/round3_experiments.txt)
    if (check) hipLaunchKernelGGL((k_rounds<true, true>), dim3(nperms * 24, ngroups), dim3(64), 0, st, K);
    else hipLaunchKernelGGL(k_rounds<false>, dim3(nperms * 24, ngroups), dim3(64), 0, st, K);
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
