#include <stdlib.h>
#include "keccak_kernels.hpp"
void launch_k_chain(const KArgs& K, bool check, uint32_t nsponges, uint32_t ngroups, hipStream_t st) {
    // check: nsponges is the number of PERMUTATIONS (local evaluation, one wavefront per (sponge, block))
    if (check) hipLaunchKernelGGL(k_chain_check, dim3(nsponges, ngroups), dim3(64), 0, st, K);
    else hipLaunchKernelGGL(k_chain<false>, dim3(nsponges, ngroups), dim3(64), 0, st, K);
}
// EXPERIMENT SWITCH (round 5, to be removed): POB_X_KCHK = <rounds per wavefront> * 1000 + <waves per SIMD> * 100 + <loads in flight per wavefront>, e.g. 4412
#define KCHK_VARIANTS(X) X(4,4,0) X(4,4,8) X(4,4,12) X(4,4,16) X(4,4,20) X(4,5,8) X(4,5,12) X(4,3,0) X(4,3,16) X(4,3,24) X(6,4,12) X(6,4,16) X(3,4,12) X(3,4,16) X(8,4,16) X(2,4,16) X(1,2,0)
static int g_kchk = -1;
static int kchk_variant() { if (g_kchk < 0) { const char* e = getenv("POB_X_KCHK"); g_kchk = e ? atoi(e) : 4412; } return g_kchk; }
extern "C" void pob_x_set_kchk(int v) { g_kchk = v; }
static int g_kwaves[2] = {0, 0};        // EXPERIMENT: wavefronts per group of the round expansion / evaluation launch (0 = one per item)
extern "C" void pob_x_set_kwaves(int which, int v) { g_kwaves[which & 1] = v; }
int pob_kchk_rounds() { return kchk_variant() / 1000; }
void launch_k_rounds(const KArgs& K, bool check, uint32_t nperms, uint32_t ngroups, hipStream_t st) {
    KArgs A = K;
    if (check) {
        switch (kchk_variant()) {
#define X(kr, w, f) case kr * 1000 + w * 100 + f: { A.count = nperms * (24 / kr); const uint32_t gx = g_kwaves[1] > 0 && (uint32_t)g_kwaves[1] < A.count ? g_kwaves[1] : A.count; \
        hipLaunchKernelGGL((k_rounds_check<true, kr, w, f>), dim3(gx, ngroups), dim3(64), 0, st, A); } break;
        KCHK_VARIANTS(X)
#undef X
        default: abort();
        }
    } else { A.count = nperms * 24; const uint32_t gx = g_kwaves[0] > 0 && (uint32_t)g_kwaves[0] < A.count ? g_kwaves[0] : A.count; hipLaunchKernelGGL(k_rounds_gen, dim3(gx, ngroups), dim3(64), 0, st, A); }
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
