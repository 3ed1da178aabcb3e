// The G-unit kernel: one wavefront = one unit (circuits.hpp) for one group of 64 witnesses, lane = witness.
#pragma once
#include "kernels_common.hpp"

extern __shared__ uint32_t g_lds[];

// CLS: 0 = light units (BIT/SM only), 1 = BN254 units (Poseidon, Num2Bits_strict, ...), 2 = SubstringCheck's BN254 units
template <class P, int CLS> __global__ void __launch_bounds__(64, CLS == 0 ? 8 : CLS == 2 ? 4 : 1) g_units(GArgs A) {
    const uint32_t lane = threadIdx.x;
    const uint32_t g = P::is_emit ? A.emit_group : blockIdx.y;
    P p;
    p.m.bits = A.bits + (uint64_t)g * A.bits_stride;
    p.m.sm = A.sm + (uint64_t)g * A.sm_stride;
    p.m.fr = A.fr + (uint64_t)g * A.fr_stride;
    p.m.inv_lut = A.inv_lut; p.m.pow256 = A.pow256; p.m.npow256 = A.npow256;
    p.m.nfr_in = A.nfr_in; p.m.nsm_in = A.nsm_in;
    p.m.in_fr = A.in_fr + (uint64_t)g * 64 * A.nfr_in * 32;
    p.m.in_sm = A.in_sm + (uint64_t)g * 64 * A.nsm_in;
    p.m.lane = lane;
    if (CLS == 1 && A.stage_lds) {   // Poseidon round constants + MDS/sparse matrices -> LDS, broadcast reads from there
        for (uint32_t i = lane; i < POS_TABLE_LEN * 8; i += 64) g_lds[i] = A.pos_tab[i];
        __syncthreads();
        p.m.pos_tab = g_lds;
    } else p.m.pos_tab = A.pos_tab;
    if constexpr (P::is_gen) p.status = 0;
    if constexpr (P::is_check) { p.status = 0; p.bad_wire = 0xFFFFFFFFu; p.pend_s = p.pend_x = p.rdiff = 0; p.attribute = false; }
    if constexpr (P::is_emit) { p.out = A.emit_out; p.sel = A.emit_sel; }
    for (int pass = 0;; pass++) {
        const UnitDesc d = A.units[A.order[A.first + blockIdx.x]];      // (re-read for the replay: nothing of it stays live across the body)
        if constexpr (CLS == 1) unit_run_heavy<P>(p, d, *A.L); else if constexpr (CLS == 2) unit_run_sc<P>(p, d, *A.L); else unit_run_light<P>(p, d, *A.L);
        if constexpr (P::is_check) {      // a lane-distributed run differed: replay the unit attributing wire by wire
            p.run_flush();
            if (pass == 0 && __ballot(p.rdiff != 0)) { p.attribute = true; continue; }
        }
        break;
    }
    if constexpr (P::is_gen) { if (p.status) atomicMin(&A.status[g * 64 + lane], p.status); }
    if constexpr (P::is_check) {
        if (p.status) atomicMin(&A.chk_status[g * 64 + lane], p.status);
        if (p.bad_wire != 0xFFFFFFFFu) atomicMin(&A.bad_wire[g * 64 + lane], p.bad_wire);
    }
}
