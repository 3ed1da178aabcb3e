// Poseidon(T-1) generation with the STATE spread over lanes (circomlib poseidon.circom, optimised schedule; the block layout is
// the one gadgets.hpp gPoseidon walks: reference circuits/proof_of_burn.circom:113-119, utils/burn_address.circom:55-57,
// spend.circom:43-44 are the call sites).
//
// The lane = witness form (gadgets.hpp) runs a Poseidon block as ONE dependent chain of 600-1 040 Montgomery products per
// wavefront, 16 wavefronts per batch of 1 024: 1.1-1.7 ms alone and 3-5 ms beside a streaming kernel.  Here a wavefront holds
// 8 witnesses x 8 state slots (lane = 8 * witness + element, T <= 5 slots in use):
//   * full round: the T S-boxes of a witness are three products deep (x^2, x^4, x^5) instead of 3T; Mix is T products per lane
//     with the round's inputs broadcast inside the 8-lane group (ds_bpermute);
//   * partial round: lane 0 runs the S-box while lanes 1.. multiply their MixS coefficient into their element in the SAME product
//     (x^2 | S_j * in_j), then one more product for every lane (S_0 * s0 | in_0 * S'_j) and a 3-step sum over the group:
//     4 products deep instead of 3 + (2T - 1);
// critical path 304 products instead of 1 040 for T = 5 (280 / 784 for T = 4, 276 / 600 for T = 3), and 8x the wavefronts.
// The HBM layout is untouched (limb planes [wire][limb][64 witnesses]): a lane stores its element's wires for its witness, the
// 8 witnesses of a wave are 32 contiguous bytes of a row.  Evaluation (CK_POS_SEG units) and emission keep the lane = witness code.
//
// Generation only.  One wavefront = (unit, group, 8-witness slice); a workgroup = POSW_WAVES wavefronts = consecutive slices of ONE unit, which share one copy of the
// constants of the unit's T in LDS (22 KB for T = 5: with a copy per wavefront -- round 5 -- the LDS held five wavefronts per CU, and beside other calculators' Poseidon
// blocks that occupancy, not the multiplier, bounded the launch).
#pragma once
#include "kernels_common.hpp"

extern __shared__ uint32_t g_lds[];

// which wires feed the block (FR ranks; POSW_NONE = absent).  Input 0 is always POSEIDON_PREFIX + pre (constants.circom:3-14).
#define POSW_NONE 0xFFFFFFFFu

// RIDE (in-order calculators, pob_set_inorder bit 2): the evaluation rides with the generation as in policy.hpp GenPT<true> -- every element a lane stores is loaded back behind
// the store and compared with the value stored, one load (8 limbs) in flight; a mismatch marks the element's wire (the lowest per lane; poswide_body reduces over the 8 lanes of a
// witness).  With it the Poseidon segments of pob_constraint_check (CK_POS_SEG: the stored states recomputed from stored operands) need not run.
template <bool RIDE, bool FAULT = false> struct PosWideT {
    __amdgpu_buffer_rsrc_t rs;     // the group's FR slab
    uint32_t slot4;                // witness slot (0..63) * 4
    bool act;                      // this lane's state slot is in use (element < T)
    const uint32_t* ktab;          // Poseidon table of this T in LDS, indexed from the first constant of T
    uint32_t kbase;                // table index of that first constant
    uint32_t w_of_f;               // wire index of FR rank f inside the block = f + w_of_f (the block's wires are all field elements, in rank order)
    uint32_t w_also;               // ... of the caller's copy of the hash (outside the block): reported at the block's first wire
    mutable Fr pl, pv; mutable uint32_t pw, bad;      // RIDE: the pending compare (loaded, stored, wire) and the lowest wire that differed
    uint32_t fault_f; bool fault_me;                  // FAULT (tests, pob_debug_store_fault): FR rank whose store reaches memory with bit 0 flipped, for this lane's witness
    __device__ __forceinline__ void ride_init() { pl = pv = fr_zero(); pw = 0; bad = 0xFFFFFFFFu; }
    __device__ __forceinline__ void ride_resolve() const { if (!fr_eq(pl, pv) && pw < bad) bad = pw; POB_OPAQUE(bad); }
    __device__ __forceinline__ Fr ld(uint32_t f) const {
        Fr v; const uint32_t off = act ? (f << 11) + slot4 : 0xFFFFF000u;       // inactive: past the slab, reads 0
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off + 256u * k), 0, 0);
        return v;
    }
    __device__ __forceinline__ void st(bool on, uint32_t f, const Fr& v, bool also = false) const {
        const uint32_t off = on ? (f << 11) + slot4 : 0xFFFFF000u;              // off: past the slab, dropped
        const uint32_t flip = (FAULT && fault_me && f == fault_f) ? 1u : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_amdgcn_raw_buffer_store_b32((int)(k == 0 ? v.l[0] ^ flip : v.l[k]), rs, (int)(off + 256u * k), 0, 0);
        if constexpr (RIDE) {
            Fr l;                                                               // (off: reads 0 ...)
#pragma unroll
            for (int k = 0; k < 8; k++) l.l[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off + 256u * k), 0, 0);
            POB_RIDE_BARRIER();
            ride_resolve();
            pl = l; pv = on ? v : fr_zero(); pw = also ? w_also : f + w_of_f;   // (... and expects 0)
        }
    }
    __device__ __forceinline__ Fr kc(uint32_t idx) const {                      // table constant (per-lane index)
        Fr v; const uint32_t* q = ktab + (size_t)(idx - kbase) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = q[k];
        return v;
    }
};
typedef PosWideT<false> PosWide;
__device__ __forceinline__ Fr posw_from(const Fr& v, uint32_t src_lane) {      // every lane := lane src_lane's element
    Fr r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.l[k] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v.l[k]);
    return r;
}
__device__ __forceinline__ Fr posw_group_sum(Fr v, uint32_t lane) {            // sum over the 8 lanes of a witness, in every lane
#pragma unroll
    for (uint32_t m = 4; m >= 1; m >>= 1) v = fr_add(v, posw_from(v, lane ^ m));
    return v;
}

// off = FR rank of the round's first wire; cr = index of its Ark constants; mat = matrix
template <int T, class PW> __device__ __forceinline__ Fr posw_full(const PW& W, uint32_t lane, uint32_t j, Fr x, uint32_t off, uint32_t cr, uint32_t mat) {
    // T x Sigma [out | in | in2, in4], Ark [out[T] | in[T]], Mix [out[T] | in[T]]
    const Fr x2 = fr_mul(x, x), x4 = fr_mul(x2, x2), x5 = fr_mul(x4, x);
    const uint32_t sg = off + 4 * j;
    W.st(W.act, sg + 1, x); W.st(W.act, sg + 2, x2); W.st(W.act, sg + 3, x4); W.st(W.act, sg, x5);
    const uint32_t jc = W.act ? j : 0;                           // (idle slots index the table like slot 0)
    const Fr y = fr_add(x5, W.kc(cr + jc));
    W.st(W.act, off + 5 * T + j, x5); W.st(W.act, off + 4 * T + j, y);
    W.st(W.act, off + 7 * T + j, y);
    Fr acc = fr_zero();
#pragma unroll 1
    for (uint32_t i = 0; i < (uint32_t)T; i++) acc = fr_add(acc, fr_mul(W.kc(mat + jc * T + i), posw_from(y, (lane & ~7u) | i)));
    W.st(W.act, off + 6 * T + j, acc);
    return acc;
}

// the unit's descriptor as scalars (read through the scalar cache: the unit index is wave-uniform)
struct PosWDesc { uint32_t base, pre, in2, in3, in4, sub, also; };
template <int T, class PW> __device__ __forceinline__ void posw_run(const GArgs& A, const PosWDesc& d, const PW& W, uint32_t lane) {
    const PosOff k = pos_off(T);
    const uint32_t j = lane & 7u, base = d.base;
    const bool act = W.act, l0 = j == 0;
    const uint32_t jc = act ? j : 0;                             // (idle slots index the table like slot 0)
    // ---- head: out | inputs[T-1] | PoseidonEx.out | PoseidonEx.inputs[T-1], initialState | Ark0 [out[T] | in[T]]
    Fr x = fr_zero();
    if (j == 1) x = A.L->prefix[d.pre];
    {
        uint32_t a2 = j == 2 ? d.in2 : 0u, a3 = j == 3 ? d.in3 : 0u, a4 = j == 4 ? d.in4 : 0u;
        POB_OPAQUE(a2); POB_OPAQUE(a3); POB_OPAQUE(a4);           // (left alone, the selects become a dynamically indexed table in scratch)
        const uint32_t aj = a2 | a3 | a4;
        const uint32_t src = (j >= 2 && j < (uint32_t)T) ? aj : POSW_NONE;
        PosWideT<false> Wi; Wi.rs = W.rs; Wi.slot4 = W.slot4; Wi.act = src != POSW_NONE;
        const Fr v = Wi.ld(src);
        if (src != POSW_NONE) x = v;
        const uint32_t sub = (j == (uint32_t)T - 1) ? d.sub : POSW_NONE;
        Wi.act = sub != POSW_NONE;
        const Fr s = Wi.ld(sub);
        if (sub != POSW_NONE) x = fr_sub(x, s);
    }
    if (!act) x = fr_zero();
    W.st(act && !l0, base + j, x);                               // inputs[j-1]
    W.st(act && !l0, base + T + j, x);                           // PoseidonEx.inputs[j-1]
    W.st(act && l0, base + 2 * T, x);                            // initialState = 0
    W.st(act, base + 3 * T + 1 + j, x);                          // Ark0.in
    x = fr_add(x, W.kc(k.C + jc));
    W.st(act, base + 2 * T + 1 + j, x);                          // Ark0.out
    uint32_t off = base + 4 * T + 1;
    // ---- 4 full rounds (the fourth mixes with P)
#pragma unroll 1
    for (uint32_t r = 0; r < 4; r++) { x = posw_full<T>(W, lane, j, x, off, k.C + (r + 1) * T, r == 3 ? k.Pm : k.M); off += 8 * T; }
    // ---- partial rounds: Sigma [out | in | in2, in4] on element 0, MixS [out[T] | in[T]]
#pragma unroll 1
    for (uint32_t r = 0; r < (uint32_t)k.rp; r++) {
        const uint32_t sb = k.S + (2 * T - 1) * r;
        // product 1: lane 0 x^2 | lanes 1.. S_j * in_j
        const Fr c1 = W.kc(sb + jc);
        const Fr m1 = fr_mul(x, l0 ? x : c1);
        const Fr x4 = fr_mul(m1, m1);                            // (lane 0 only)
        const Fr x5 = fr_mul(x4, x);
        const Fr s0 = fr_add(x5, W.kc(k.C + 5 * T + r));
        W.st(act && l0, off + 1, x); W.st(act && l0, off + 2, m1); W.st(act && l0, off + 3, x4); W.st(act && l0, off, x5);
        const Fr in = l0 ? s0 : x;
        W.st(act, off + 4 + T + j, in);
        // product 2: lane 0 S_0 * in_0 | lanes 1.. in_0 * S'_j
        const Fr in0 = posw_from(s0, lane & ~7u);
        const Fr c2 = W.kc(jc == 0 ? sb : sb + T + jc - 1);
        const Fr m2 = fr_mul(c2, in0);
        Fr part = l0 ? m2 : m1;
        if (!act) part = fr_zero();
        const Fr sum = posw_group_sum(part, lane);
        x = l0 ? sum : fr_add(x, m2);
        W.st(act, off + 4 + j, x);
        off += 4 + 2 * T;
    }
    // ---- 3 full rounds
#pragma unroll 1
    for (uint32_t r = 0; r < 3; r++) { x = posw_full<T>(W, lane, j, x, off, k.C + 5 * T + k.rp + r * T, k.M); off += 8 * T; }
    // ---- tail: T x Sigma, MixLast [out | in[T]], PoseidonEx.out, out (and the caller's copy of the hash)
    {
        const Fr x2 = fr_mul(x, x), x4 = fr_mul(x2, x2), x5 = fr_mul(x4, x);
        const uint32_t sg = off + 4 * j;
        W.st(act, sg + 1, x); W.st(act, sg + 2, x2); W.st(act, sg + 3, x4); W.st(act, sg, x5);
        W.st(act, off + 4 * T + 1 + j, x5);
        Fr part = fr_mul(W.kc(k.M + jc), x5);
        if (!act) part = fr_zero();
        const Fr h = posw_group_sum(part, lane);
        W.st(act && l0, off + 4 * T, h);
        W.st(act && l0, base + T, h);
        W.st(act && l0, base, h);
        W.st(act && l0 && d.also != POSW_NONE, d.also, h, true);
    }
}

// bx = 8 * unit + witness slice (unit = position in the launch's list), g = group; the wavefronts of a workgroup have the same unit
#ifndef POSW_WAVES
#define POSW_WAVES 4
#endif
static_assert(8 % POSW_WAVES == 0, "the slices of a workgroup belong to one unit");
template <bool RIDE, bool FAULT = false> __device__ __forceinline__ void poswide_body(const GArgs& A, uint32_t bx, uint32_t g) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & 63u;
    const UnitDesc* dp = A.units + POB_UNI(A.order[A.first + (bx >> 3)]);
    const int T = (int)POB_UNI(dp->a[0]);
    const PosWDesc d = {POB_UNI(dp->cur.f), POB_UNI(dp->a[1]), POB_UNI(dp->a[2]), POB_UNI(dp->a[3]), POB_UNI(dp->a[4]), POB_UNI(dp->a[5]), POB_UNI(dp->a[6])};
    const PosOff k = pos_off(T);
    const uint32_t kend = T == 3 ? POS_OFF_C_4 : T == 4 ? POS_OFF_C_5 : POS_TABLE_LEN;     // the constants of one T are contiguous
    for (uint32_t i = threadIdx.x; i < (kend - k.C) * 8; i += blockDim.x) g_lds[i] = A.pos_tab[(size_t)k.C * 8 + i];
    __syncthreads();
    PosWideT<RIDE, FAULT> W;
    uint32_t* frp = A.fr + (uint64_t)g * A.fr_stride;
    const uint64_t nf = A.fr_stride * 4;
    W.rs = __builtin_amdgcn_make_buffer_rsrc(frp, 0, (int)(nf > 0xFFFFF000ull ? 0xFFFFF000ull : nf), 0x00020000);
    W.slot4 = (8 * (bx & 7u) + (lane >> 3)) * 4;
    W.act = (lane & 7u) < (uint32_t)T;
    W.ktab = g_lds; W.kbase = k.C;
    W.w_of_f = POB_UNI(dp->cur.w) - d.base; W.w_also = POB_UNI(dp->cur.w);
    if constexpr (RIDE) W.ride_init();
    if constexpr (FAULT) { W.fault_f = A.fault_idx; W.fault_me = A.fault_cls == 2 && g == A.fault_group && ((A.fault_lanes >> (8 * (bx & 7u) + (lane >> 3))) & 1); }
    if (T == 3) posw_run<3>(A, d, W, lane); else if (T == 4) posw_run<4>(A, d, W, lane); else posw_run<5>(A, d, W, lane);
    if constexpr (RIDE) {          // the last pending compare; the lowest differing wire over the 8 lanes (elements) of a witness, reported by its first lane
        W.ride_resolve();
        uint32_t b = W.bad;
#pragma unroll
        for (uint32_t m = 4; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ m) << 2), (int)b); b = o < b ? o : b; }
        if ((lane & 7u) == 0 && b != 0xFFFFFFFFu) atomicMin(&A.bad_wire[g * 64 + 8 * (bx & 7u) + (lane >> 3)], b);
    }
}
// grid = (8 / POSW_WAVES * nunits, ngroups) workgroups of POSW_WAVES wavefronts
template <bool RIDE, bool FAULT> __global__ void __launch_bounds__(64 * POSW_WAVES) k_poseidon_wide(GArgs A) { poswide_body<RIDE, FAULT>(A, POSW_WAVES * blockIdx.x + (threadIdx.x >> 6), blockIdx.y); }
static_assert(POS_TABLE_LEN - POS_OFF_C_5 >= POS_OFF_C_5 - POS_OFF_C_4 && POS_TABLE_LEN - POS_OFF_C_5 >= POS_OFF_C_4 - POS_OFF_C_3, "the T = 5 constants are the largest set");
#define POSW_LDS_BYTES ((POS_TABLE_LEN - POS_OFF_C_5) * 32u)
