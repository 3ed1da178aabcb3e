// Kernel argument blocks and host-callable launchers shared by the translation units of libpob_hip.so.
// The library is split so the heavy template instantiations (generator / constraint evaluator / emitter of the
// G units, Keccak kernels) compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "circuits.hpp"

typedef unsigned long long u64;

struct GArgs {
    const UnitDesc* units; const uint32_t* order; uint32_t first;
    CircuitLayout* L;                                  // read-only on the device (only the host planner writes reference tables)
    uint64_t* bits; int32_t* sm; uint32_t* fr;
    uint64_t bits_stride, sm_stride, fr_stride;       // elements per group
    const uint32_t* pos_tab; const uint32_t* inv_lut; const uint32_t* pow256; uint32_t npow256;
    const uint8_t* in_fr; const int32_t* in_sm; uint32_t nfr_in, nsm_in;
    uint32_t* status; uint32_t* chk_status; uint32_t* bad_wire;
    uint8_t* emit_out; uint32_t emit_sel, emit_group;
    uint32_t emit_w0, emit_wn;                         // emission window: wires [emit_w0, emit_w0 + emit_wn) land at emit_out + 32 * (w - emit_w0)
    unsigned long long* emit_probe;                    // probe pass: nothing is written, bit (w / emit_wn) of emit_probe[unit] is set instead
    const unsigned long long* emit_rbits; const uint32_t* emit_rpre;   // reduced witness: kept-wire bitmap + per-word rank (policy.hpp EmitP); null = O0 payload
    uint32_t* emit_sites; uint32_t emit_sites_cap;     // self-check site-recording pass (policy.hpp EmitP::sites)
    uint32_t* emit_counters;                           // which of the emitter's inverse paths ran (policy.hpp EmitP::ctr, pob_debug_emit_counters)
    uint32_t fault_cls, fault_group, fault_idx; uint64_t fault_lanes;      // pob_debug_store_fault (tests): the riding kernels' FAULT instantiations (policy.hpp GenPT<true, true>, poseidon_wide.hpp); fault_cls 0xFFFFFFFF = none
};
struct KArgs {
    u64* bits;                 // BIT slabs, all groups
    uint64_t group_stride;     // words per group
    const SpongeDesc* sponges;
    const uint32_t* perm_sponge;   // flattened (sponge, block) list for the round kernels
    const uint32_t* perm_block;
    uint32_t* bad_wire;        // per witness: lowest inconsistent wire (CheckIO)
    uint32_t first, count;
    uint32_t fault_group; uint64_t fault_word, fault_mask;   // k_rounds_gc<FAULT> (tests): the store at BIT word fault_word of group fault_group goes out XORed with fault_mask
};

// grid = (nunits, ngroups) wavefronts.  Generation: one kernel per scheduling class (light | SubstringCheck BN254 | the other BN254
// units at <= 128 VGPRs) + the Poseidon blocks (poseidon_wide.hpp, 8 wavefronts per unit and group); constraint evaluation: one kernel
// per family (circuits.hpp Fam); emission: light | BN254 | SubstringCheck.
void launch_g_gen_light(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_gen_sc(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_pos_wide(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st, bool ride = false, bool fault = false);      // ride: every element stored is loaded back and compared (poseidon_wide.hpp)
void launch_g_gen_n2b(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_gen_all(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);       // light | BN254 | SubstringCheck units in one kernel (in-order calculators)
void launch_g_gen_all_ride(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);  // ... whose evaluation rides with them (policy.hpp GenPT<true>)
void launch_g_gen_all_ride_fault(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);      // ... with one corrupted store (tests)
void launch_g_check_narrow(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);  // MISC | RL | POS | N2B evaluation families in one kernel (in-order calculators)
void launch_g_check_misc(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_range(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_selrow(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_ld(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_rl(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_sc(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_pos(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_n2b(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_emit_light(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_emit_heavy(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_emit_sc(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
// gadget-level mains (gadget_mains.hpp): generation / evaluation / emission of family F_GM on the policies that resolve input references
void launch_g_gen_gm(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_check_gm(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
void launch_g_emit_gm(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st);
// fused launch of the in-order schedule: npos Poseidon blocks + nsponges sponge chains from K.first in one launch (g_gen_poswide.hip)
void launch_pos_chain(const GArgs& A, const KArgs& K, uint32_t npos, uint32_t nsponges, uint32_t ngroups, hipStream_t st, bool ride = false, bool fault = false);
void launch_k_chain(const KArgs& K, bool check, uint32_t nsponges, uint32_t ngroups, hipStream_t st);
void launch_k_rounds(const KArgs& K, bool check, uint32_t nperms, uint32_t ngroups, hipStream_t st);
// generation + evaluation of the round blocks in one launch (keccak_kernels.hpp k_rounds_gc); fault: the tests' instantiation that corrupts one store
void launch_k_rounds_gc(const KArgs& K, uint32_t nperms, uint32_t ngroups, bool fault, hipStream_t st);
int pob_kchk_rounds();        // rounds per wavefront of the round evaluation (k_keccak.hip)
int pob_kgc_rounds();         // ... of the launch that expands AND evaluates the round blocks (k_rounds_gc)
void launch_k_emit_bits(const u64* G, uint8_t* out, uint32_t wire_base, uint32_t bit_base, uint32_t count, uint32_t sel, hipStream_t st);
// reduced form: wires [wire0, wire0 + count) with BIT ranks from bit_base; kept wires land at out + 32 * (rank - k0) when rank - k0 < kn
void launch_k_emit_bits_red(const u64* G, uint8_t* out, uint32_t wire0, uint32_t bit_base, uint32_t count, uint32_t sel, const unsigned long long* rbits, const uint32_t* rpre, uint32_t k0, uint32_t kn, hipStream_t st);
// an Absorb block (circuits.hpp ABSORB_BITS): wires at offsets [o0, o0 + count) of the block whose storage starts at BIT rank ab; tab = the round
// blocks' alias table on the device (keccak_alias_table_host fills the host copy: KECCAKF_ROUND_WIRES codes, false = the walk is inconsistent)
void launch_k_emit_absorb(const u64* G, uint8_t* out, uint32_t ab, uint32_t o0, uint32_t count, uint32_t sel, const uint16_t* tab, hipStream_t st);
void launch_k_emit_absorb_red(const u64* G, uint8_t* out, uint32_t wire0, uint32_t ab, uint32_t o0, uint32_t count, uint32_t sel, const uint16_t* tab, const unsigned long long* rbits,
                              const uint32_t* rpre, uint32_t k0, uint32_t kn, hipStream_t st);
bool keccak_alias_table_host(uint16_t* tab);
