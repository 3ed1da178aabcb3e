// Witness-vector layout + execution policies.
//
// The O0 witness of the reference circuits (every declared signal is a wire; reference Makefile:2-3)
// is kept resident in HBM in a COMPACT TYPED form, one slab per group of 64 witnesses (= one
// wavefront, lane = witness):
//
//   BIT  wires (provably 0/1 in a valid witness; 99.4 % of all wires, all of Keccak):
//        one uint64 per wire per group; bit l = value in witness l.  This is exactly the format of a
//        CDNA wave lane-mask (VCC/EXEC), so gate logic on BIT wires is one 64-bit scalar op for 64
//        witnesses and a comparison result becomes a wire with one v_cmp (ballot).
//   SM   wires (bytes, lengths, small signed differences): int32 [wire][64 lanes] (256 B rows).
//   FR   wires (genuine field elements): 8 x uint32 limb planes [wire][limb][64 lanes], Montgomery form.
//   DV   wires (DERIVED): every wire that is a function of stored wires / inputs / constants within its unit and that no other unit
//        reads.  They are NOT stored (p.dvs(n) allocates wire indices only): the emitter rebuilds them from the same expressions
//        (EmitP::derived / derived_inv / derived_fr / derived_fr_inv: value, field inverse with 0 -> 0), generation and evaluation
//        skip them and define / check the gadgets' BIT outputs directly (their relations hold by construction; that the rebuilt
//        values are the right ones is what the payload comparison with the oracle checks).  These are
//          * the operand wires of every IsZero / IsEqual gadget (in[0], in[1], IsZero.in, IsZero.inv): the selectors' IsEqual([select, i])
//            of the leaf detectors, SelectorArray1D and the Keccak outputs, Pad's IsEqual([i, inLen]), ShiftLeft's n^2 IsEqual,
//            SubstringCheck's isLastIndex and -- over field elements -- its IsEqual(exists) (gIsEqualFd);
//          * copies: Selector.vals[], SelectorArray1D.arrays / arraysT, Pad's and AssertByteString's byte copies, SubstringCheck.mainInput[];
//          * running sums that are functions of stored wires: Selector.sum[], SubstringCheck.M[];
//        1.29 M of the 1.36 M non-BIT wires of the production circuit.  Rounds 1-2 stored them (int32 rows, the Keccak selectors'
//        as an int8 class of their own, IsZero.inv as its operand code, M[] and the IsEqual(exists) operands as field elements).
//        Round 4 added BIT-valued derived wires, written by the emitter as lane-distributed runs (run_derived):
//          * copies of a stored bit: five of the six copies of every padded byte's bits (only Keccak's inBlocks is stored), nine of the ten copies of the
//            hash bits (only Selector.out), vals[] / arrays / arraysT of the Keccak output selectors (copies of Final.s), the IsEqual children's outputs
//            under every stored isEq[] / isLast[] bit;
//          * functions of a stored value: the Keccak output selectors' isEq[] / sum[] (of numBlocks and Final.s), AssertByteString's 16 bits per byte;
//        and the products temps[][] / temp[][] of ShiftRight / ShiftLeft (isEq[i] * in[j]) and parts[] of CompConstant: 3.54 M derived wires in all.
//   AL   wires (ALIAS, round 4): the wires of a KeccakfRound block other than its 76 stored gate-output arrays -- copies of a stored array / of the round's
//        input or output state, possibly rotated or negated, or constants.  skip_alias() advances the wire index only; keccak_kernels.hpp names each of
//        them once (the walker that generates and evaluates, run on a symbolic value type) and the emitter expands through that table.  197 M wires.
//
// Storage index of a wire = its rank among the wires of its class in wire order, so any contiguous
// run of same-class wires (e.g. a whole Keccak-f block, 2 506 944 BIT wires) is contiguous in HBM.
//
// Every circuit template is written ONCE as `template <class P>` code (gadgets.hpp) and instantiated
// with four policies:
//   CountP  (host)   only advances the cursor: layout planner, no memory traffic;
//   GenP    (device) computes values and stores them                          -> witness generation;
//   CheckP  (device) re-reads every wire and verifies it against its defining expression evaluated
//                    on the STORED operands (and every ===)                   -> constraint evaluator;
//   EmitP   (device) re-reads every wire and writes it as a canonical 32-byte LE field element
//                    at its O0 wire index                                     -> .wtns emitter.
#pragma once
#include <stdint.h>
#include "fr_dev.hpp"

typedef uint64_t B;   // BIT value: lane mask over the 64 witnesses of the group (wave-uniform)
typedef int32_t S;    // SM value of this lane's witness
typedef Fr F;         // FR value (Montgomery) of this lane's witness

struct Cur { uint32_t w, b, s, f, q; };   // next free: wire index, BIT rank, SM rank, FR rank; q = derived wires so far (no storage)
HD Cur cur_add(Cur a, Cur d, uint32_t k) { Cur r = {a.w + d.w * k, a.b + d.b * k, a.s + d.s * k, a.f + d.f * k, a.q + d.q * k}; return r; }
struct BitRef { uint32_t w, i; HD BitRef operator+(uint32_t k) const { BitRef r = {w + k, i + k}; return r; } };
struct SmRef  { uint32_t w, i; HD SmRef  operator+(uint32_t k) const { SmRef  r = {w + k, i + k}; return r; } };
struct FrRef  { uint32_t w, i; HD FrRef  operator+(uint32_t k) const { FrRef  r = {w + k, i + k}; return r; } };

// failure codes: (template id << 12) | source line of the failing assert / === in the reference circuits
enum : uint32_t {
    T_NUM2BITS = 1, T_ISZERO, T_ALIASCHECK, T_ASSERT_LT, T_ASSERT_LE, T_ASSERT_GE, T_DIVIDE, T_SELECTOR,
    T_POW, T_POB, T_INPUT, T_COMPCONST, T_MISC
};
#define FAILCODE(tpl, line) (((uint32_t)(tpl) << 12) | (uint32_t)(line))

// Two circomlib templates instantiate a child in a different order than they DECLARE it (Num2Bits_strict: aliasCheck is declared first,
// n2b initialised first; MultiAND(n > 2): and2 is declared first, initialised last).  Which of the two orders circom's --O0 numbering
// follows is not pinned here (no circom).  Default: initialisation order.  POB_DECL_ORDER=1 in the environment (read once,
// when the first circuit is planned) flips BOTH statements -- together with ORACLE_DECL_ORDER=1 for the oracle and the circuit model
// (which reads the same POB_DECL_ORDER), so that a circom user can re-diff a real .sym / .wtns with one knob.
#if defined(__HIP_DEVICE_COMPILE__)
#define POB_DECL_ORDER_DEFAULT 0u                     // (device policies copy the flag from the circuit layout)
#else
#include <stdlib.h>
static inline uint32_t pob_decl_order_env() { static const uint32_t v = (getenv("POB_DECL_ORDER") && getenv("POB_DECL_ORDER")[0] == '1') ? 1u : 0u; return v; }
#define POB_DECL_ORDER_DEFAULT pob_decl_order_env()
#endif
struct PolBase {
    Cur cur;
    uint32_t decl_order = POB_DECL_ORDER_DEFAULT;
    HD BitRef bits(uint32_t n) { BitRef r = {cur.w, cur.b}; cur.w += n; cur.b += n; return r; }
    HD SmRef sms(uint32_t n) { SmRef r = {cur.w, cur.s}; cur.w += n; cur.s += n; return r; }
    HD uint32_t dvs(uint32_t n) { const uint32_t w = cur.w; cur.w += n; cur.q += n; return w; }     // n derived wires: no storage, the first one's wire index
    HD FrRef frs(uint32_t n) { FrRef r = {cur.w, cur.f}; cur.w += n; cur.f += n; return r; }
    HD void skip_alias(uint32_t nw, uint32_t nb) { cur.w += nw; cur.b += nb; }       // nw wires of which nb are stored BIT wires, the rest aliases of them (Absorb blocks, circuits.hpp)
};

// ------------------------------------------------------------------ host-side layout planner policy
// sub-blocks of a composite unit that the constraint evaluator runs as wavefronts of their own (circuits.hpp CK_* units);
// the counting policy notes them while the planner walks the composite
enum : uint32_t { NOTE_POSEIDON = 1, NOTE_N2BE = 2, NOTE_SHIFTRIGHT = 3, NOTE_SHIFTLEFT = 4 };
struct PlanNote { uint32_t what, n; Cur cur; uint32_t a[7]; };
#define PLAN_MAX_NOTES 8
struct CountP : PolBase {
    static constexpr bool is_gen = false, is_check = false, is_emit = false, is_count = true, ride = false;
    uint32_t nput = 0;    // wires written: the planner's cost estimate of a unit (long units are dispatched first)
    uint32_t nb = 0, ns = 0, nf = 0;      // ... by storage class (tools/plan_stats.cpp)
    PlanNote notes[PLAN_MAX_NOTES]; uint32_t nnotes = 0;
    bool notes_overflow = false;          // a composite with more split sub-blocks than the table holds: the planner refuses the plan (Plan::take_notes)
    HD void note(uint32_t what, uint32_t n, Cur c, uint32_t a0 = 0, uint32_t a1 = 0, uint32_t a2 = 0, uint32_t a3 = 0, uint32_t a4 = 0, uint32_t a5 = 0, uint32_t a6 = 0) {
        if (nnotes >= PLAN_MAX_NOTES) { notes_overflow = true; return; }
        PlanNote& x = notes[nnotes++]; x.what = what; x.n = n; x.cur = c;
        x.a[0] = a0; x.a[1] = a1; x.a[2] = a2; x.a[3] = a3; x.a[4] = a4; x.a[5] = a5; x.a[6] = a6;
    }
    HD B put(BitRef, B v) { nput++; nb++; return v; }
    HD S put(SmRef, S v) { nput++; ns++; return v; }
    HD F put(FrRef, const F& v) { nput += 8; nf++; return v; }
    HD B hint(BitRef, B v) { nput++; nb++; return v; }
    HD S hint(SmRef, S v) { nput++; ns++; return v; }
    HD F hint(FrRef, const F& v) { nput += 8; nf++; return v; }
    HD B get(BitRef) { return 0; }
    HD S get(SmRef) { return 0; }
    HD S get_lane(SmRef, uint32_t) { return 0; }
    HD void derived(uint32_t, S) {}               // a DERIVED wire (see the header): value v / the inverse of x; only the emitter does anything
    HD void derived_inv(uint32_t, S, bool = false) {}       // (true: the IsZero is the child of an IsEqual [out | in[2]] -- the emitter's self-check, EmitP::w32)
    HD void derived_fr(uint32_t, const F&) {}     // ... with a field-element value (Montgomery) / the field inverse of x (0 for 0)
    HD void derived_fr_inv(uint32_t, const F&, bool = true) {}
    HD void site_m(uint32_t, uint32_t, uint32_t) {}         // self-check site of SubstringCheck's M[] recurrence (EmitP)
    HD void site_c(uint32_t, uint32_t) {}                   // self-check site of a copy constraint a === b between a derived wire and the stored wire it must equal
    HD F get(FrRef) { return fr_zero(); }
    HD void raw_put(FrRef, const F&) {}
    HD B ballot(bool) { return 0; }
    HD bool bit(B) { return false; }
    HD void require(B, uint32_t) {}
    HD void require_lane(bool, uint32_t) {}              // the assert on this lane's own predicate (no ballot)
    HD F kconst(uint32_t) { return fr_zero(); }   // Poseidon table entry (Montgomery)
    HD F k256(uint32_t) { return fr_zero(); }     // 256^i (Montgomery)
    HD F k256r(uint32_t) { return fr_zero(); }    // 256^i * R (so that fr_mul(plain small integer, k256r(i)) is Montgomery(small * 256^i))
    HD F input_fr(uint32_t) { return fr_zero(); }
    HD S input_sm(uint32_t) { return 0; }
    HD uint32_t lane_id() { return 0; }
    // lane-distributed BIT access (see DevPol): n wires
    HD B run_get(uint32_t, uint32_t) { return 0; }
    HD void run_put(uint32_t n, uint32_t, uint32_t, B) { nput += n; nb += n; }
    HD void run_derived(uint32_t, uint32_t, B) {}      // n DERIVED BIT wires as a lane-distributed run (lane k: wire index, that wire's 64-witness mask): the emitter only
    HD B run_bcast(B, uint32_t) { return 0; }
    HD B xpose64(uint32_t, uint32_t, uint32_t) { return 0; }
    HD B xpose(B, uint32_t) { return 0; }
    HD B run_set(B run, uint32_t, B) { return run; }
    HD B run_perm(B run, uint32_t) { return run; }
};

// Two-phase SM batch: sm_load issues the evaluator's loads of N wires BEFORE their expected values are computed, sm_commit
// writes (generation) / compares (evaluation) them.
template <int N> struct SmLoaded { S s[N]; };
template <class P, int N> HD __attribute__((always_inline)) SmLoaded<N> sm_load(P& p, const SmRef (&r)[N]) {
    SmLoaded<N> h;
    if constexpr (P::is_check) {
#pragma unroll
        for (int k = 0; k < N; k++) h.s[k] = p.get(r[k]);
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) h.s[k] = 0;
    }
    return h;
}
template <class P, int N> HD __attribute__((always_inline)) void sm_commit(P& p, const SmRef (&r)[N], const SmLoaded<N>& h, const S (&v)[N]) {
    if constexpr (P::is_check) {
#pragma unroll
        for (int k = 0; k < N; k++) p.mark(h.s[k] != v[k], r[k].w);
        p.pin();
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) p.put(r[k], v[k]);
    }
}
// n SM wires ref(i) := val(i), BATCH at a time: the evaluator's loads of a batch are requested together, ahead of their compares (as single puts in a loop that is not
// unrolled every wire is its own memory round trip -- the compare of one is resolved before the next load is issued: 0.4-0.5 us per wire; the byte conversions and the fixed
// concatenations of the BN254 composites were 100-300 such wires in a row: their evaluation units the longest of the narrow kernel)
template <class P, int BATCH, class RefFn, class ValFn> HD __attribute__((always_inline)) void sm_puts_at(P& p, int n, RefFn ref, ValFn val) {
    for (int i0 = 0; i0 < n; i0 += BATCH) {
        SmRef rr[BATCH]; S vv[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; q++) rr[q] = ref(i0 + q < n ? i0 + q : n - 1);       // (a short last batch repeats its last wire: same wire, same value)
        const SmLoaded<BATCH> h = sm_load(p, rr);
#pragma unroll
        for (int q = 0; q < BATCH; q++) vv[q] = val(i0 + q < n ? i0 + q : n - 1);
        sm_commit(p, rr, h, vv);
    }
}
template <class P, int BATCH, class ValFn> HD __attribute__((always_inline)) void sm_puts(P& p, SmRef base, int n, ValFn val) {
    sm_puts_at<P, BATCH>(p, n, [&](int i) { return base + (uint32_t)i; }, val);
}
// Lane-distributed bit vector of up to 256 BIT wires: bit 64q + k lives in lane k of r[q] as that wire's 64-witness mask.  A
// decomposition (Num2Bits and everything copied from it) is built ONCE from the witnesses' canonical values and then written /
// verified as runs -- no wire of it is ever read back.
struct BV { B r[4]; };
// The decomposition of the 64 witnesses' canonical values IS a bit-matrix transposition: lane l holds the 64-bit word x_l (two limbs) and lane k must end up with
// the mask whose bit l is bit k of x_l.  p.xpose64 does it across the lanes (device: six butterfly stages, one ds_bpermute each -- DevPol::xpose64); rounds 1-5 built
// the run from 254 ballots, each a compare into an SGPR pair and two selects under a per-lane `lane == k` mask (another 64 SGPR pairs, hoisted and spilled:
// 3.6-5.5 k spilled SGPRs in the kernels that decompose field elements).
template <class P> HD __attribute__((always_inline)) BV bv_from_canon(P& p, const F& c, int n) {
    BV v;
#pragma unroll
    for (int q = 0; q < 4; q++) v.r[q] = (64 * q < n) ? p.xpose64(c.l[2 * q], c.l[2 * q + 1], (uint32_t)(n - 64 * q)) : 0;
    return v;
}
// wires dst + 0 .. n-1 := bits 0 .. n-1
template <class P> HD __attribute__((always_inline)) void bv_put(P& p, BitRef dst, int n, const BV& v) {
    const uint32_t ln = p.lane_id();
#pragma unroll
    for (int q = 0; q < 4; q++) if (64 * q < n) p.run_put((uint32_t)(n - 64 * q < 64 ? n - 64 * q : 64), dst.w + 64 * q + ln, dst.i + 64 * q + ln, v.r[q]);
}
// the same bits as the BIT wires of n/per consecutive child components (each: `per` bits at wire offset `off`, `cw` wires in all)
template <class P> HD __attribute__((always_inline)) void bv_put_children(P& p, uint32_t w0, uint32_t b0, uint32_t cw, uint32_t off, uint32_t per, int n, const BV& v) {
    const uint32_t ln = p.lane_id();
#pragma unroll
    for (int q = 0; q < 4; q++) if (64 * q < n) {
        const uint32_t idx = 64 * q + ln;
        p.run_put((uint32_t)(n - 64 * q < 64 ? n - 64 * q : 64), w0 + (idx / per) * cw + off + idx % per, b0 + idx, v.r[q]);
    }
}
HD bool canon_bit(const F& c, int k) {          // bit k of a canonical value (k compile-time or uniform)
    bool b = false;
#pragma unroll
    for (int j = 0; j < 8; j++) if ((k >> 5) == j) b = (c.l[j] >> (k & 31)) & 1;
    return b;
}
HD S canon_byte(const F& c, int i) {            // little-endian byte i
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) if ((i >> 2) == j) w = c.l[j];
    return (S)((w >> (8 * (i & 3))) & 0xffu);
}
// dst[i] = src[i], i < n, BATCH loads ahead of the stores / compares (a copy loop otherwise pays one memory round trip per wire)
template <class P, int BATCH> HD __attribute__((always_inline)) void sm_copy(P& p, SmRef dst, SmRef src, int n, bool rev = false) {   // rev: dst[n-1-i] = src[i]
    for (int i0 = 0; i0 < n; i0 += BATCH) {
        SmRef rr[BATCH]; S vv[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; q++) { const int i = i0 + q < n ? i0 + q : n - 1; rr[q] = dst + (uint32_t)(rev ? n - 1 - i : i); }
        const SmLoaded<BATCH> h = sm_load(p, rr);
#pragma unroll
        for (int q = 0; q < BATCH; q++) vv[q] = p.get(src + (uint32_t)(i0 + q < n ? i0 + q : n - 1));
        sm_commit(p, rr, h, vv);
    }
}
// dst[i] = src[i], i < n: SM wires in batches of 16, BIT wires as lane-distributed runs of 64
template <class P> HD __attribute__((always_inline)) void copy_n(P& p, SmRef dst, SmRef src, int n) { sm_copy<P, 8>(p, dst, src, n); }
template <class P> HD __attribute__((always_inline)) void copy_n(P& p, BitRef dst, BitRef src, int n) {
    const uint32_t ln = p.lane_id();
    for (int q = 0; q < n; q += 64) {
        const uint32_t m = (uint32_t)(n - q < 64 ? n - q : 64);
        const B v = p.run_get(m, src.i + q + ln);
        p.run_put(m, dst.w + q + ln, dst.i + q + ln, v);
    }
}
// Two-phase FR batch (Poseidon rounds): the evaluator requests all N stored elements (8 N loads) before the expected values --
// N Montgomery products deep -- are computed, then compares; generation stores.
template <int N> struct FrLoaded { F s[N]; };
template <class P, int N> HD __attribute__((always_inline)) FrLoaded<N> fr_load(P& p, const FrRef (&r)[N]) {
    FrLoaded<N> h;
    if constexpr (P::is_check) {
#pragma unroll
        for (int k = 0; k < N; k++) h.s[k] = p.get(r[k]);
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) h.s[k] = fr_zero();
    }
    return h;
}
template <class P, int N> HD __attribute__((always_inline)) void fr_commit(P& p, const FrRef (&r)[N], const FrLoaded<N>& h, const F (&v)[N]) {
    if constexpr (P::is_check) {
#pragma unroll
        for (int k = 0; k < N; k++) { p.mark(!fr_eq(h.s[k], v[k]), r[k].w); p.pin(); }
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) p.put(r[k], v[k]);
    }
}
// n DERIVED wires w0 + t*dw, t < n, that all carry the value v (inv: the field inverse of v, 0 for 0): nothing but the emitter touches them
template <class P> HD __attribute__((always_inline)) void derived_rows_same(P& p, uint32_t w0, uint32_t dw, uint32_t n, S v, bool inv = false) {
    if constexpr (P::is_emit) {
        for (uint32_t t = 0; t < n; t++) { if (inv) p.derived_inv(w0 + t * dw, v, true); else p.derived(w0 + t * dw, v); }      // (inv: the IsZero.inv wires of IsEqual children)
    } else { (void)p; (void)w0; (void)dw; (void)n; (void)v; (void)inv; }
}
#ifdef __HIPCC__
// A BIT wire is shared by the 64 lanes of its wavefront (one lane stores the mask, every lane may load it later).  On the GPU the
// wavefront executes in lockstep, so a later load sees the store; the CPU test shim (tests/hostsim) runs the lanes as independent
// fibers and needs a meeting point after such a store.  Nothing on the GPU.
#ifdef POB_HOSTSIM
#define POB_WAVE_FENCE() __syncthreads()
#else
#define POB_WAVE_FENCE() ((void)0)
#endif
// ------------------------------------------------------------------ device policies
struct DevMem {
    uint64_t* bits;     // this group's BIT slab
    int32_t* sm;        // this group's SM slab  [NS][64]
    uint32_t* fr;       // this group's FR slab  [NF][8][64]
    const uint32_t* pos_tab;  // Poseidon table (Montgomery, [idx][8]) -- LDS copy when staged
    const uint32_t* inv_lut;  // canonical inverses of -4096..4096, [k+4096][8]
    const uint32_t* pow256;   // [2][n][8]: 256^i in Montgomery form, then 256^i * R^2 mod p
    const uint8_t* in_fr;     // this group's packed inputs: FR inputs [64 lanes][nfr][32 B canonical LE]
    const int32_t* in_sm;     //                              SM inputs [64 lanes][nsm]
    uint32_t nfr_in, nsm_in, npow256;
    uint32_t lane;
    uint32_t fault_cls, fault_idx; uint64_t fault_lanes;      // GenPT<true, true> (tests, pob_debug_store_fault): the store of storage class / rank reaches memory corrupted for these witnesses
    // buffer resources over the three slabs: a wire access is `buffer_load/store v, v_lane_offset, s[rsrc], s_wire_offset offen`
    // -- the per-wire part of the address stays scalar, no 64-bit per-lane address arithmetic (or registers) per access
    __amdgpu_buffer_rsrc_t rs_bits, rs_sm, rs_fr;
    uint32_t lane4;     // lane * 4
};
typedef int pob_v2i __attribute__((ext_vector_type(2)));
#define POB_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))      // wire indices are wave-uniform by construction (get_lane is the exception)

struct DevPol : PolBase {
    DevMem m;
    __device__ __forceinline__ B ballot(bool pred) { return __ballot(pred); }
    __device__ __forceinline__ bool bit(B v) { return (v >> m.lane) & 1; }
    __device__ __forceinline__ uint32_t lane_id() { return m.lane; }
    __device__ __forceinline__ B ld(BitRef r) { return m.bits[r.i]; }
    __device__ __forceinline__ S ld(SmRef r) { return __builtin_amdgcn_raw_buffer_load_b32(m.rs_sm, (int)m.lane4, (int)(POB_UNI(r.i) << 8), 0); }
    __device__ __forceinline__ void derived(uint32_t, S) {}
    __device__ __forceinline__ void derived_inv(uint32_t, S, bool = false) {}
    __device__ __forceinline__ void site_m(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void site_c(uint32_t, uint32_t) {}
    __device__ __forceinline__ void derived_fr(uint32_t, const F&) {}
    __device__ __forceinline__ void derived_fr_inv(uint32_t, const F&, bool = true) {}
    __device__ __forceinline__ void run_derived(uint32_t, uint32_t, B) {}
    __device__ __forceinline__ F ld(FrRef r) {
        F v; const uint32_t so = POB_UNI(r.i) << 11;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(m.rs_fr, (int)(m.lane4 + 256u * k), (int)so, 0);
        return v;
    }
    // one lane stores the wave-uniform mask (a 64-lane same-address store costs the texture-address unit 64 lanes of work:
    // measured 1.5x slower on the selector-row stage)
    __device__ __forceinline__ void st(BitRef r, B v) { if (m.lane == 0) m.bits[r.i] = v; POB_WAVE_FENCE(); }
    __device__ __forceinline__ void st(SmRef r, S v) { __builtin_amdgcn_raw_buffer_store_b32(v, m.rs_sm, (int)m.lane4, (int)(POB_UNI(r.i) << 8), 0); }
    __device__ __forceinline__ void st(FrRef r, const F& v) {
        const uint32_t so = POB_UNI(r.i) << 11;
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_amdgcn_raw_buffer_store_b32((int)v.l[k], m.rs_fr, (int)(m.lane4 + 256u * k), (int)so, 0);
    }
    // Lane-distributed BIT access: lane k < n owns ONE wire (its own BIT rank i, wire index w) and holds that wire's 64-witness
    // mask, i.e. the wave works on up to 64 wires x 64 witnesses at once, bit-sliced like the Keccak kernels.  A run of consecutive
    // wires is one coalesced 512-byte access instead of 64 single-lane ones.
    // inactive lanes (>= n) address past the end of the slab: a raw-buffer load there returns 0 and a store is dropped, so the
    // access needs no exec-mask branch
    __device__ __forceinline__ uint32_t run_off(uint32_t n, uint32_t i) { return m.lane < n ? (i << 3) : 0xFFFFFFF8u; }
    __device__ __forceinline__ B run_ld_off(uint32_t off) {
        const pob_v2i q = __builtin_amdgcn_raw_buffer_load_b64(m.rs_bits, (int)off, 0, 0);
        return ((B)(uint32_t)q.y << 32) | (uint32_t)q.x;
    }
    __device__ __forceinline__ B run_get(uint32_t n, uint32_t i) { return run_ld_off(run_off(n, i)); }
    __device__ __forceinline__ B run_bcast(B x, uint32_t k) {     // lane k's value, wave-uniform
        return ((B)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)k);
    }
    __device__ __forceinline__ B run_set(B run, uint32_t k, B mask) { return m.lane == k ? mask : run; }   // lane k of the run := wave-uniform mask
    __device__ __forceinline__ B run_perm(B run, uint32_t src_lane) {   // this lane := lane src_lane's value
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)run);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)(run >> 32));
        return ((B)hi << 32) | lo;
    }
    // 64 x 64 bit-matrix transposition across the wavefront: in: this lane's (witness') 64-bit word lo | hi << 32; out: lane k holds the mask whose bit l is bit k of
    // lane l's word -- the lane-distributed run of the 64 BIT wires "bit k of the value", lanes k >= n zeroed.  Butterfly over j = 32, 16, .., 1: the lanes l and l ^ j
    // exchange the off-diagonal j x j blocks; each lane needs 32 of its partner's 64 bits, packed into ONE word per stage (one ds_bpermute).
    __device__ __forceinline__ B xpose64(uint32_t lo, uint32_t hi, uint32_t n) {
        {   // j = 32: the off-diagonal blocks are whole words
            const bool up = m.lane & 32;
            const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((m.lane ^ 32u) << 2), (int)(up ? lo : hi));
            if (up) lo = got; else hi = got;
        }
#pragma unroll
        for (int st = 0; st < 5; st++) {
            const uint32_t j = 16u >> st;
            const uint32_t mk = j == 16 ? 0x0000FFFFu : j == 8 ? 0x00FF00FFu : j == 4 ? 0x0F0F0F0Fu : j == 2 ? 0x33333333u : 0x55555555u;      // bit positions with (pos & j) == 0
            const bool up = m.lane & j;
            // what the partner needs of this lane: its columns of the other half, both words packed into one
            const uint32_t send = up ? ((lo & mk) | ((hi & mk) << j)) : (((lo & ~mk) >> j) | (hi & ~mk));
            const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((m.lane ^ j) << 2), (int)send);
            if (up) { lo = (lo & ~mk) | (got & mk); hi = (hi & ~mk) | ((got & ~mk) >> j); }
            else { lo = (lo & mk) | ((got & mk) << j); hi = (hi & mk) | (got & ~mk); }
        }
        const B r = ((B)hi << 32) | lo;
        return m.lane < n ? r : 0;
    }
    __device__ __forceinline__ B xpose(B word, uint32_t n) { return xpose64((uint32_t)word, (uint32_t)(word >> 32), n); }      // bit t of this witness' word -> lane t's 64-witness mask
    __device__ __forceinline__ B get(BitRef r) { return ld(r); }
    __device__ __forceinline__ S get(SmRef r) { return ld(r); }
    // SM wire base + k with a PER-WITNESS k (generation shortcuts that index by a witness value)
    __device__ __forceinline__ S get_lane(SmRef base, uint32_t k) { return __builtin_amdgcn_raw_buffer_load_b32(m.rs_sm, (int)(m.lane4 + ((base.i + k) << 8)), 0, 0); }
    __device__ __forceinline__ F get(FrRef r) { return ld(r); }
    __device__ __forceinline__ F kconst(uint32_t idx) {
        F v; const uint32_t* q = m.pos_tab + (size_t)idx * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = q[k];
        return v;
    }
    __device__ __forceinline__ F k256(uint32_t i) {
        F v; const uint32_t* q = m.pow256 + (size_t)i * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) v.l[k] = q[k];
        return v;
    }
    __device__ __forceinline__ F k256r(uint32_t i) { return k256(i + m.npow256); }
    __device__ __forceinline__ F input_fr(uint32_t k) {   // canonical LE bytes -> Montgomery
        const uint32_t* q = (const uint32_t*)(m.in_fr + ((size_t)m.lane * m.nfr_in + k) * 32);
        F c;
#pragma unroll
        for (int j = 0; j < 8; j++) c.l[j] = q[j];
        return fr_to_mont(c);
    }
    __device__ __forceinline__ S input_sm(uint32_t k) { return m.in_sm[(size_t)m.lane * m.nsm_in + k]; }
};

// Generation.  RIDE (in-order calculators, pob_set_inorder bit 2: g_gen_all_ride.hip): the evaluation rides with the generation -- every wire a unit stores is LOADED back behind
// the store and compared with the value stored; for small values, single bits and runs one load is in flight per class (the compare of a put is resolved when the NEXT put of its class
// has issued its load, the way CheckP::run_put does it), a field element is compared at once; a mismatch marks the wire (bad_wire, the lowest per witness).  With that, every `<==` / `<--` holds between the STORED wires as it does between the
// computed ones, and every `===` was evaluated on operands equal to the stored ones: what CheckP establishes by recomputing every unit from stored operands -- the seven G-family
// launches of pob_constraint_check -- without a second pass over the vector (the loads hit L2: the store is a few instructions old).  A debug poke, or a vector that no launch has
// just written, is evaluated by CheckP as before.
#ifndef POB_RIDE_BARRIER
#define POB_RIDE_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
template <bool RIDE, bool FAULT = false> struct GenPT : DevPol {
    static constexpr bool is_gen = true, is_check = false, is_emit = false, is_count = false, ride = RIDE;
    // FAULT (tests): what goes to memory for the armed (class, rank): bit 0 of the value flipped for the witnesses of fault_lanes -- the unit goes on with the right value
    __device__ __forceinline__ bool hit(uint32_t cls, uint32_t i) const { return FAULT && m.fault_cls == cls && m.fault_idx == i; }
    uint32_t status;   // this lane's first failing assert (0 = none yet)
    // RIDE: lowest wire whose loaded value differs from the value stored (per lane), and the one pending compare per class
    uint32_t bad_wire;
    S ps_l, ps_v; uint32_t ps_w;                  // SM: loaded, stored, wire
    B pb_l, pb_v; uint32_t pb_w;                  // BIT (one wave-uniform word)
    B pr_l, pr_x; uint32_t pr_w;                  // lane-distributed run: lane k holds wire pr_w's 64-witness mask
    __device__ __forceinline__ void ride_init() { bad_wire = 0xFFFFFFFFu; ps_l = ps_v = 0; ps_w = 0; pb_l = pb_v = 0; pb_w = 0; pr_l = pr_x = 0; pr_w = 0; }
    __device__ __forceinline__ void mark(bool bad, uint32_t w) { if (bad && w < bad_wire) bad_wire = w; }
    __device__ __forceinline__ void res_s() { mark(ps_l != ps_v, ps_w); }
    __device__ __forceinline__ void res_b() { mark(((pb_l ^ pb_v) >> m.lane) & 1, pb_w); }
    __device__ __forceinline__ void res_r() { if (__ballot((pr_l ^ pr_x) != 0)) ride_attribute(); }
    // end of the unit: the pending compares; a run that differed is attributed wire by wire (corrupted stores only)
    __device__ __forceinline__ void ride_flush() {
        res_s(); res_b(); res_r();
        ps_l = ps_v = 0; pb_l = pb_v = 0; pr_l = pr_x = 0;
    }
    __device__ __forceinline__ B put(BitRef r, B v) {
        st(r, hit(0, r.i) ? v ^ m.fault_lanes : v);
        if constexpr (RIDE) { const B l = run_ld_off(POB_UNI(r.i) << 3); POB_RIDE_BARRIER(); res_b(); pb_l = l; pb_v = v; pb_w = r.w; }
        return v;
    }
    __device__ __forceinline__ S put(SmRef r, S v) {
        st(r, (hit(1, r.i) && ((m.fault_lanes >> m.lane) & 1)) ? v ^ 1 : v);
        if constexpr (RIDE) { const S l = ld(r); POB_RIDE_BARRIER(); res_s(); ps_l = l; ps_v = v; ps_w = r.w; }
        return v;
    }
    __device__ __forceinline__ F put(FrRef r, const F& v) {
        if (hit(2, r.i) && ((m.fault_lanes >> m.lane) & 1)) { F w = v; w.l[0] ^= 1u; st(r, w); } else st(r, v);
        if constexpr (RIDE) {
            const F l = ld(r); mark(!fr_eq(l, v), r.w); POB_OPAQUE(bad_wire);      // (resolved at once: a pending field element is 16 VGPRs the kernel does not have at four wavefronts per SIMD)
        }
        return v;
    }
    __device__ __forceinline__ B hint(BitRef r, B v) { return put(r, v); }
    __device__ __forceinline__ S hint(SmRef r, S v) { return put(r, v); }
    __device__ __forceinline__ F hint(FrRef r, const F& v) { return put(r, v); }
    __device__ __forceinline__ void raw_put(FrRef r, const F& v) { put(r, v); }
    __device__ __forceinline__ void require(B ok, uint32_t code) { if (!bit(ok) && status == 0) status = code; }
    __device__ __forceinline__ void require_lane(bool ok, uint32_t code) { if (!ok && status == 0) status = code; }
    __device__ __forceinline__ void run_put(uint32_t n, uint32_t w, uint32_t i, B x) {
        const B xs = hit(0, i) ? x ^ m.fault_lanes : x;       // (i: this lane's BIT rank)
        pob_v2i q; q.x = (int)(uint32_t)xs; q.y = (int)(uint32_t)(xs >> 32);
        const uint32_t off = run_off(n, i);
        __builtin_amdgcn_raw_buffer_store_b64(q, m.rs_bits, (int)off, 0, 0);
        POB_WAVE_FENCE();
        if constexpr (RIDE) {
            const B l = run_ld_off(off);                    // (inactive lanes read 0 ...)
            POB_RIDE_BARRIER();
            res_r();
            pr_l = l; pr_x = m.lane < n ? x : 0; pr_w = w;  // (... and expect 0)
        }
    }
    // a run differed (a corrupted store): which wires, for which witnesses -- lane j holds the difference mask of wire pr_w(j)
    __device__ __forceinline__ void ride_attribute() {
        const B d = pr_l ^ pr_x;
        uint64_t act = __ballot(d != 0);
        while (act) {
            const int j = __builtin_ctzll(act); act &= act - 1;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d >> 32), j);
            const uint32_t wj = (uint32_t)__builtin_amdgcn_readlane((int)pr_w, j);
            if (((((B)hi << 32) | lo) >> m.lane) & 1) mark(true, wj);
        }
    }
};
typedef GenPT<false> GenP;
typedef GenPT<true> GenRideP;
typedef GenPT<true, true> GenRideFaultP;
// a plain policy with the state of a riding one (memory, cursor, status): the unit kinds that generate without the riding evaluation inside a riding kernel run on it
// (circuits.hpp unit_run_ride) and hand their status back; the copies are register moves
template <bool FAULT> __device__ __forceinline__ GenPT<false> plain_of(const GenPT<true, FAULT>& p) {
    GenPT<false> q;
    static_cast<DevPol&>(q) = static_cast<const DevPol&>(p);
    q.status = p.status;
    return q;
}

// slow path of CheckP::run_put (by value: the policy object stays in registers): lane j holds the difference mask d of wire w
__device__ __forceinline__ uint32_t check_attribute_run(B d, uint32_t w, uint32_t lane, uint32_t bad_wire) {
    uint64_t act = __ballot(d != 0);
    while (act) {
        const int j = __builtin_ctzll(act); act &= act - 1;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d >> 32), j);
        const uint32_t wj = (uint32_t)__builtin_amdgcn_readlane((int)w, j);
        const B dj = ((B)hi << 32) | lo;
        if (((dj >> lane) & 1) && wj < bad_wire) bad_wire = wj;
    }
    return bad_wire;
}

// Constraint evaluator: `put(w, e)` = "the STORED wire w must equal the expression e", and it returns the STORED value of w, so
// every later expression is evaluated on stored operands: each `<==` / `===` is checked as a relation between stored wires,
// locally -- a wrong wire flags its own definition and the relations of its direct consumers, nothing downstream is re-derived
// from it.  That locality is what lets the evaluator cut the generator's serial chains (Poseidon rounds, byte conversions) into
// independent units that start from stored wires (circuits.hpp CK_* units).
struct CheckP : DevPol {
    static constexpr bool is_gen = false, is_check = true, is_emit = false, is_count = false, ride = false;
    uint32_t status;     // first failing === site of this lane
    uint32_t bad_wire;   // lowest wire index whose stored value contradicts its definition (per lane)
    __device__ __forceinline__ void mark(bool bad, uint32_t w) { if (bad && w < bad_wire) bad_wire = w; }
    __device__ __forceinline__ void pin() { POB_OPAQUE(bad_wire); }      // the compares so far are resolved HERE (see put(FrRef))
    __device__ __forceinline__ B put(BitRef r, B v) { const B s = ld(r); mark(((s ^ v) >> m.lane) & 1, r.w); return s; }
    __device__ __forceinline__ S put(SmRef r, S v) { const S s = ld(r); mark(s != v, r.w); return s; }
    // (the opaque asm pins the compare HERE: left alone, LLVM sinks every compare to the end of the kernel -- bad_wire is only used
    //  there -- and keeps both 8-limb operands of every relation alive until then: thousands of spilled VGPRs)
    __device__ __forceinline__ F put(FrRef r, const F& v) { const F s = ld(r); mark(!fr_eq(s, v), r.w); POB_OPAQUE(bad_wire); return s; }
    __device__ __forceinline__ void put_flush() {}
    __device__ __forceinline__ B hint(BitRef r, B) { return ld(r); }
    __device__ __forceinline__ S hint(SmRef r, S) { return ld(r); }
    __device__ __forceinline__ F hint(FrRef r, const F&) { return ld(r); }
    __device__ __forceinline__ void raw_put(FrRef, const F&) {}
    __device__ __forceinline__ void require(B ok, uint32_t code) { if (!bit(ok) && status == 0) status = code; }
    __device__ __forceinline__ void require_lane(bool ok, uint32_t code) { if (!ok && status == 0) status = code; }
    // lane-distributed runs: a run's difference is folded into `rdiff` when the NEXT run's load has been issued (one load is
    // always in flight).  Only if a unit ends with rdiff != 0 (corrupted vector) it is replayed with `attribute` set, which
    // resolves every run on the spot and finds the lowest mismatching wire of each witness.
    B pend_s, pend_x, rdiff; uint32_t pend_w; bool attribute;
    __device__ __forceinline__ void run_resolve() {
        const B d = pend_s ^ pend_x;
        rdiff |= d;
        if (attribute) { if (__ballot(d != 0)) bad_wire = check_attribute_run(d, pend_w, m.lane, bad_wire); }
    }
    __device__ __forceinline__ void run_put(uint32_t n, uint32_t w, uint32_t i, B x) {
        const B s = run_ld_off(run_off(n, i));          // inactive lanes read 0 ...
        __builtin_amdgcn_sched_barrier(0);              // keep the issue of this load ahead of the wait on the previous one
        run_resolve();                                  // (the previous run: its load has had this one's issue to complete)
        pend_s = s; pend_x = m.lane < n ? x : 0; pend_w = w;   // ... and expect 0
    }
    __device__ __forceinline__ void run_flush() { run_resolve(); pend_s = pend_x = 0; }
};

// .wtns emitter for ONE witness of the group (lane `sel`): canonical 32-byte LE value at wire index.
struct EmitP : DevPol {
    static constexpr bool is_gen = false, is_check = false, is_emit = true, is_count = false, ride = false;
    uint8_t* out;      // canonical witness payload of the wires [w0, w0 + wn) (the emission window), 32 B per wire
    uint32_t sel, w0, wn, unit;
    unsigned long long* probe;      // probe pass (once per window size): which windows does this unit write to?  bit w / wn of probe[unit]
    // reduced witness (pob_emit_begin_reduced): bit w of rbits = wire w survives, rpre[w / 64] = kept wires below 64 * (w / 64); a
    // surviving wire lands at its RANK among the kept wires and windows count kept wires.  Null: the O0 payload, position = wire index.
    const unsigned long long* rbits; const uint32_t* rpre;
    // test hook (pob_debug_emit_counters): [0] IsZero.inv from the table of small inverses, [1] by Fermat exponentiation (|operand| > 4096),
    // [2] field-element IsZero.inv of a NON-zero operand (Kaliski inversion), [3] of a zero operand -- counted per emitted wire, probe passes excluded
    uint32_t* ctr;
    // SELF-CHECK (pob_emit_selfcheck): the relations of the derived wires are evaluated on the values WRITTEN INTO THE WINDOW by a kernel that
    // re-reads them (pob_host.hip k_selfcheck_*).  Where those relations live is found once per handle by a site-recording pass: `sites` non-null
    // = nothing is written, every IsZero.inv wire (the calls of w32 that carry a path) appends its gadget's first wire -- IsZero [out | in | inv],
    // bit 31: child of an IsEqual [out | in[2]] -- to sites[4 ..], every step of SubstringCheck's M[] recurrence a triple behind them, every copy constraint
    // between a derived wire and a STORED one (an IsEqual child's outputs === the parent's stored isEq[] bit) a pair behind those.  sites[0..2] = the three counts.
    // path: 0..3 = the inverse path counted, | 4 = IsEqual parent.
    uint32_t* sites; uint32_t sites_cap;
    __device__ __forceinline__ void site_m(uint32_t w_next, uint32_t w_byte, uint32_t k) {      // M[k+1] (wire w_next) === M[k] (w_next - 1) + mainInput[k] (w_byte) * 256^k
        if (sites && m.lane == sel) { const uint32_t i = atomicAdd(sites + 1, 1u); if (i < sites_cap) { uint32_t* q = sites + 4 + sites_cap + 3 * (size_t)i; q[0] = w_next; q[1] = w_byte; q[2] = k; } }
    }
    __device__ __forceinline__ void site_c(uint32_t w, uint32_t src) {              // wire w === wire src  (one lane calls)
        if (sites) { const uint32_t i = atomicAdd(sites + 2, 1u); if (i < sites_cap) { uint32_t* q = sites + 4 + 4 * (size_t)sites_cap + 2 * (size_t)i; q[0] = w > src ? w : src; q[1] = w > src ? src : w; } }
    }
    __device__ __forceinline__ void w32(uint32_t w, const F& canon, int path = -1) {
        if (sites) { if (path >= 0) { const uint32_t i = atomicAdd(sites, 1u); if (i < sites_cap) sites[4 + i] = (w - 2) | ((path & 4) ? 0x80000000u : 0u); } return; }
        if (rbits) {
            const unsigned long long word = rbits[w >> 6];
            if (!((word >> (w & 63)) & 1)) return;
            w = rpre[w >> 6] + (uint32_t)__popcll(word & ((1ull << (w & 63)) - 1));
        }
        if (probe) { atomicOr(probe + unit, 1ull << (w / wn)); return; }
        if (w - w0 >= wn) return;
        if (path >= 0) atomicAdd(ctr + (path & 3), 1u);
        uint4* q = (uint4*)(out + (size_t)(w - w0) * 32);
        q[0] = make_uint4(canon.l[0], canon.l[1], canon.l[2], canon.l[3]);
        q[1] = make_uint4(canon.l[4], canon.l[5], canon.l[6], canon.l[7]);
    }
    __device__ __forceinline__ F small(S k) {
        Fr c = {{(uint32_t)(k < 0 ? -k : k), 0, 0, 0, 0, 0, 0, 0}};
        return (k < 0) ? fr_sub(fr_zero(), c) : c;     // canonical arithmetic: p - |k|
    }
    __device__ __forceinline__ B put(BitRef r, B) {
        B s = ld(r);
        if (m.lane == 0) { Fr c = {{(uint32_t)((s >> sel) & 1), 0, 0, 0, 0, 0, 0, 0}}; w32(r.w, c); }
        return s;
    }
    __device__ __forceinline__ S put(SmRef r, S) { S s = ld(r); if (m.lane == sel) w32(r.w, small(s)); return s; }
    __device__ __forceinline__ F put(FrRef r, const F&) { F s = ld(r); if (m.lane == sel) w32(r.w, fr_from_mont(s)); return s; }
    __device__ __forceinline__ B hint(BitRef r, B v) { return put(r, v); }
    __device__ __forceinline__ S hint(SmRef r, S v) { return put(r, v); }
    __device__ __forceinline__ F hint(FrRef r, const F& v) { return put(r, v); }
    __device__ __forceinline__ void derived(uint32_t w, S v) { if (m.lane == sel) w32(w, small(v)); }          // (shadow DevPol's no-ops)
    __device__ __forceinline__ void derived_inv(uint32_t w, S x, bool iseq = false) { emit_inv(w, x, iseq); }
    __device__ __forceinline__ void derived_fr(uint32_t w, const F& v) { if (m.lane == sel) w32(w, fr_from_mont(v)); }
    __device__ __forceinline__ void derived_fr_inv(uint32_t w, const F& x, bool iseq = true) { if (m.lane == sel) { const bool z = fr_is_zero(x); w32(w, z ? fr_zero() : fr_from_mont(fr_inv(x)), (z ? 3 : 2) | (iseq ? 4 : 0)); } }       // (iseq: an IsEqual's child, gadgets.hpp iseqf_derived; false: a bare IsZero, gIsZeroFd)
    // the selected witness' value in every lane (a unit that has many inverses to rebuild spreads them over the lanes: circuits.hpp U_SC_RANGE)
    __device__ __forceinline__ F bcast_sel(const F& v) {
        F r;
#pragma unroll
        for (int k = 0; k < 8; k++) r.l[k] = (uint32_t)__builtin_amdgcn_readlane((int)v.l[k], (int)sel);
        return r;
    }
    __device__ __forceinline__ void emit_inv(uint32_t w, S k, bool iseq = false) {
        if (m.lane == sel) {
            F c;
            if (k >= -4096 && k <= 4096) {
                const uint32_t* q = m.inv_lut + (size_t)(k + 4096) * 8;
#pragma unroll
                for (int j = 0; j < 8; j++) c.l[j] = q[j];
            } else {
                c = fr_from_mont(fr_inv_fermat(fr_from_i64(k)));
            }
            w32(w, c, ((k >= -4096 && k <= 4096) ? 0 : 1) | (iseq ? 4 : 0));
        }
    }
    __device__ __forceinline__ void raw_put(FrRef, const F&) {}
    __device__ __forceinline__ void require(B, uint32_t) {}
    __device__ __forceinline__ void require_lane(bool, uint32_t) {}
    // derived BIT wires: lane k < n writes wire w (its own) from the mask x it holds
    __device__ __forceinline__ void run_derived(uint32_t n, uint32_t w, B x) { if (m.lane < n) { Fr c = {{(uint32_t)((x >> sel) & 1), 0, 0, 0, 0, 0, 0, 0}}; w32(w, c); } }
    __device__ __forceinline__ void run_put(uint32_t n, uint32_t w, uint32_t i, B) {
        if (m.lane < n) { B s = run_ld_off(i << 3); Fr c = {{(uint32_t)((s >> sel) & 1), 0, 0, 0, 0, 0, 0, 0}}; w32(w, c); }
    }
};

// Gadget-level mains (gadget_mains.hpp: the reference's test wrappers around single templates, tests/test.py:146-201): the main component
// IS the template, so its SM input arrays are not wires of a parent but the packed inputs.  A reference with w >= GM_INPUT_W stands for
// "packed SM input i"; the policies the gadget-main kernels run resolve it in get() / get_lane(), everything else is the base policy.
#define GM_INPUT_W 0xFFF00000u
template <class Base> struct GmPol : Base {
    using Base::get;
    __device__ __forceinline__ S get(SmRef r) { return r.w >= GM_INPUT_W ? this->input_sm(r.i) : Base::get(r); }
    __device__ __forceinline__ S get_lane(SmRef base, uint32_t k) { return base.w >= GM_INPUT_W ? this->input_sm(base.i + k) : Base::get_lane(base, k); }
};
#endif  // __HIPCC__
