// The G-unit kernels: one wavefront = one unit (circuits.hpp) for one group of 64 witnesses, lane = witness.
#pragma once
#include "kernels_common.hpp"

// MASK = families (circuits.hpp Fam) this kernel serves; WAVES = waves per SIMD it is compiled for (VGPR budget 512 / WAVES)
// the BODY: a device function of the launch arguments and the (group, position in the launch's unit list) of its wavefront, so that a launch can carry the wavefronts of
// several independent kernels (the fused launches of g_gen_all.hip / g_gen_poswide.hip / g_check_wide.hip / g_check_narrow.hip)
template <class P, uint32_t MASK> __device__ __forceinline__ void g_units_body(const GArgs& A, uint32_t g_in, uint32_t ux) {
    // the few long BN254 chains share their SIMDs with thousands of short light / Keccak waves: let the arbiter favour them
    if constexpr ((MASK & ~FAM_LIGHT) != 0) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x;
    const uint32_t g = P::is_emit ? A.emit_group : g_in;
    P p;
    p.m.bits = A.bits + (uint64_t)g * A.bits_stride;
    p.m.sm = A.sm + (uint64_t)g * A.sm_stride;
    p.m.fr = A.fr + (uint64_t)g * A.fr_stride;
    p.m.inv_lut = A.inv_lut; p.m.pow256 = A.pow256; p.m.npow256 = A.npow256;
    p.m.nfr_in = A.nfr_in; p.m.nsm_in = A.nsm_in;
    p.m.in_fr = A.in_fr + (uint64_t)g * 64 * A.nfr_in * 32;
    p.m.in_sm = A.in_sm + (uint64_t)g * 64 * A.nsm_in;
    p.m.lane = lane; p.m.lane4 = lane * 4;
    p.m.fault_cls = g == A.fault_group ? A.fault_cls : 0xFFFFFFFFu; p.m.fault_idx = A.fault_idx; p.m.fault_lanes = A.fault_lanes;
    {   // raw buffer resources (gfx9 word3: 32-bit data format); BIT ranks index 8-byte words: i << 3 < 2^32 needs bits_stride < 2^29
        const uint64_t nb = A.bits_stride * 8, ns = A.sm_stride * 4, nf = A.fr_stride * 4;
        p.m.rs_bits = __builtin_amdgcn_make_buffer_rsrc(p.m.bits, 0, (int)(nb > 0xFFFFFFFFull ? 0xFFFFFFFFull : nb), 0x00020000);
        p.m.rs_sm = __builtin_amdgcn_make_buffer_rsrc(p.m.sm, 0, (int)(ns > 0xFFFFFFFFull ? 0xFFFFFFFFull : ns), 0x00020000);
        p.m.rs_fr = __builtin_amdgcn_make_buffer_rsrc(p.m.fr, 0, (int)(nf > 0xFFFFFFFFull ? 0xFFFFFFFFull : nf), 0x00020000);
    }
    p.m.pos_tab = A.pos_tab;
    p.decl_order = A.L->decl_order;
    if constexpr (P::is_gen) { p.status = 0; if constexpr (P::ride) p.ride_init(); }
    if constexpr (P::is_check) { p.status = 0; p.bad_wire = 0xFFFFFFFFu; p.pend_s = p.pend_x = p.rdiff = 0; p.pend_w = 0; p.attribute = false; }
    if constexpr (P::is_emit) { p.out = A.emit_out; p.sel = A.emit_sel; p.w0 = A.emit_w0; p.wn = A.emit_wn; p.probe = A.emit_probe; p.rbits = A.emit_rbits; p.rpre = A.emit_rpre; p.ctr = A.emit_counters; p.sites = A.emit_sites; p.sites_cap = A.emit_sites_cap; p.unit = A.order[A.first + ux]; }
    for (int pass = 0;; pass++) {
        const UnitDesc d = A.units[A.order[A.first + ux]];      // (re-read for the replay: nothing of it stays live across the body)
        if constexpr ((MASK & ~FAM_LIGHT) == 0) { if (d.cost >= 2500) __builtin_amdgcn_s_setprio(2); }     // long serial light units (RLP assembly, ...)
        if constexpr (P::is_gen) { if constexpr (P::ride) unit_run_ride<MASK>(p, d, *A.L); else unit_run<P, MASK>(p, d, *A.L); }
        else unit_run<P, MASK>(p, d, *A.L);
        if constexpr (P::is_check) {      // a lane-distributed run differed: replay the unit attributing wire by wire
            p.run_flush();
            if (pass == 0 && __ballot(p.rdiff != 0)) { p.attribute = true; continue; }
        }
        break;
    }
    if constexpr (P::is_gen) {
        if (p.status) atomicMin(&A.status[g * 64 + lane], p.status);
        if constexpr (P::ride) {        // the evaluation that rode with the unit: its pending compares, then the verdict where CheckP would have put it
            p.ride_flush();
            if (p.status) atomicMin(&A.chk_status[g * 64 + lane], p.status);
            if (p.bad_wire != 0xFFFFFFFFu) atomicMin(&A.bad_wire[g * 64 + lane], p.bad_wire);
        }
    }
    if constexpr (P::is_check) {
        if (p.status) atomicMin(&A.chk_status[g * 64 + lane], p.status);
        if (p.bad_wire != 0xFFFFFFFFu) atomicMin(&A.bad_wire[g * 64 + lane], p.bad_wire);
    }
}
// grid = (groups, units): the group index runs fastest, so the units of a launch start in the order of the list (longest first, circuits.hpp cost) for ALL groups at
// once -- as (units, groups) the long units of the last groups started when everything of the groups before them had been dispatched: the launch's tail
template <class P, uint32_t MASK, int WAVES> __global__ void __launch_bounds__(64, WAVES) g_units(GArgs A) { g_units_body<P, MASK>(A, blockIdx.x, blockIdx.y); }
// one launcher per kernel (each in its own translation unit, compiled in parallel)
#define POB_DEFINE_G_LAUNCH(name, POL, MASK, WAVES)                                                                            \
    void name(const GArgs& A, uint32_t nunits, uint32_t ngroups, hipStream_t st) {                                             \
        hipLaunchKernelGGL((g_units<POL, (MASK), WAVES>), dim3(ngroups, nunits), dim3(64), 0, st, A);                          \
    }
