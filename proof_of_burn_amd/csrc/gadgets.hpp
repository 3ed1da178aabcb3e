// Circuit templates of the reference (circuits/utils/*.circom + the circomlib templates they include),
// written once over a policy P (policy.hpp): layout planning, witness generation, constraint evaluation
// and .wtns emission are four instantiations of the SAME code, so the O0 wire order is stated once.
//
// Conventions
//   * lane = witness: scalar signal values are per-lane (S / F) or lane masks (B); arrays are passed as
//     references to wires that were already written (SmRef / BitRef) and re-read through L2.
//   * own signals are declared first, in circom's O0 order: outputs | inputs | intermediates (declaration
//     order); sub-components follow in initialisation order.  Comments give
//     `[out | in | mid] || children` and the reference file:line.
//   * `<==`  -> p.put(ref, expr)   `<--` -> p.hint(ref, value)   `===`/assert -> p.require(mask, code)
#pragma once
#include <type_traits>
#include "policy.hpp"

#if defined(POB_GM_KERNELS) && defined(__HIPCC__)
#define GD __host__ __device__ inline            // the gadget-level mains' kernels (g_*_gm.hip): 40 templates in one switch, off the production path -- called, not inlined (compile time)
#elif defined(__HIPCC__) && (defined(__HIP_DEVICE_COMPILE__) || defined(POB_HOSTSIM))
#define GD __host__ __device__ __forceinline__   // the policy object must stay in registers: one non-inlined callee taking P& forces it (and every p.cur/p.m access) through memory
#elif defined(__HIPCC__)
#define GD __host__ __device__ inline            // host pass (layout planner): no forced inlining -- it only made the planner's translation unit slow to compile
#else
#define GD
#endif

// ============================================================================ circomlib: gates.circom
template <class P> HD B gXOR(P& p, B a, B b) { BitRef o = p.bits(3); a = p.put(o + 1, a); b = p.put(o + 2, b); return p.put(o, a ^ b); }
template <class P> HD B gAND(P& p, B a, B b) { BitRef o = p.bits(3); a = p.put(o + 1, a); b = p.put(o + 2, b); return p.put(o, a & b); }
template <class P> HD B gOR(P& p, B a, B b) { BitRef o = p.bits(3); a = p.put(o + 1, a); b = p.put(o + 2, b); return p.put(o, a | b); }

// ============================================================================ circomlib: bitify.circom
// Num2Bits(n)  [out[n] | in];  out[i] <-- (in>>i)&1;  sum out[i] 2^i === in   (n <= 31 here)
// `top`: value of out[n-1]; `also`: a second array that must equal out[] (the caller's copy) -- both avoid re-reading
// just-written wires through memory.
template <class P> GD BitRef gNum2BitsS(P& p, int n, S in, B* top = nullptr, const BitRef* also = nullptr) {
    BitRef o = p.bits(n); SmRef i = p.sms(1);
    S x = p.put(i, in);
    uint32_t acc = 0;
    for (int k = 0; k < n; k++) {
        B b = p.hint(o + k, p.ballot(((uint32_t)x >> k) & 1));
        acc |= (uint32_t)p.bit(b) << k;
        if (also) p.put(*also + k, b);
        if (top) *top = b;
    }
    p.require(p.ballot((uint32_t)x == acc), FAILCODE(T_NUM2BITS, 38));   // negative / too wide values fail here
    return o;
}
// Num2Bits(8) with the 8 output masks returned in registers (fully unrolled)
template <class P> HD void gNum2Bits8(P& p, S in, B* outv) {
    BitRef o = p.bits(8); SmRef i = p.sms(1);
    S x = p.put(i, in);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        outv[k] = p.hint(o + k, p.ballot(((uint32_t)x >> k) & 1));
        acc |= (uint32_t)p.bit(outv[k]) << k;
    }
    p.require(p.ballot((uint32_t)x == acc), FAILCODE(T_NUM2BITS, 38));
}
// field-element flavour (n <= 254): the bits are produced lane-distributed (BV) from the canonical value; callers that need a bit
// of their own witness read it from the canonical value (cout), callers that copy the bits reuse the vector (vout).
// Evaluation: out[k] must equal the bits of the canonical `in` (unique below 2^254 < 2p: an aliased decomposition is a mismatch)
// and `in` must fit n bits (the `lc1 === in` of bitify.circom:38).
template <class P> GD BitRef gNum2BitsF(P& p, int n, const F& in, BV* vout = nullptr, F* cout = nullptr) {
    BitRef o = p.bits(n); FrRef i = p.frs(1);
    F x = p.put(i, in);
    F c = fr_from_mont(x);
    const BV v = bv_from_canon(p, c, n);
    bv_put(p, o, n, v);
    bool ok = true;
    if (n < 254) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (32 * j >= n) ok = ok && c.l[j] == 0;
            else if (32 * j + 32 > n) ok = ok && (c.l[j] >> (n - 32 * j)) == 0;
        }
    }
    p.require(p.ballot(ok), FAILCODE(T_NUM2BITS, 38));
    if (vout) *vout = v;
    if (cout) *cout = c;
    return o;
}
// Bits2Num(8)  [out | in[8]]
template <class P> HD S gBits2Num8(P& p, BitRef src) {
    SmRef o = p.sms(1); BitRef in = p.bits(8);
    S v = 0;
    for (int k = 0; k < 8; k++) { B b = p.put(in + k, p.get(src + k)); v |= (S)p.bit(b) << k; }
    return p.put(o, v);
}

// ============================================================================ circomlib: comparators.circom
// IsZero  [out | in | inv];  inv <-- in!=0 ? 1/in : 0;  out <== -in*inv+1;  in*out === 0
// Small operands: `in` and `inv` are DERIVED wires (policy.hpp) -- the emitter rebuilds them from the caller's operand; out = [in == 0].
template <class P> HD B gIsZeroS(P& p, S in, bool iseq = false) {        // iseq: the child of an IsEqual (the emitter's self-check also evaluates in === in[1] - in[0], out === IsZero.out)
    BitRef o = p.bits(1); const uint32_t w = p.dvs(2);
    p.derived(w, in); p.derived_inv(w + 1, in, iseq);
    return p.put(o, p.ballot(in == 0));
}
// IsZero over a field element with DERIVED in/inv wires: out = [in == 0] needs no inverse; the emitter rebuilds in and inv and its self-check
// evaluates in*inv === 1 - out, in*out === 0 on the written values
template <class P> GD B gIsZeroFd(P& p, const F& in) {
    BitRef o = p.bits(1); const uint32_t w = p.dvs(2);
    if constexpr (P::is_emit) { p.derived_fr(w, in); p.derived_fr_inv(w + 1, in, false); }
    return p.put(o, p.ballot(fr_is_zero(in)));
}
// IsEqual  [out | in[2]] || IsZero(in[1]-in[0])      (small operands: in[] are derived wires)
template <class P> HD B gIsEqualS(P& p, S a, S b) {
    BitRef o = p.bits(1); const uint32_t w = p.dvs(2);
    p.derived(w, a); p.derived(w + 1, b);
    return p.put(o, gIsZeroS(p, (S)((uint32_t)b - (uint32_t)a), true));
}
// IsEqual over field elements with DERIVED operand wires (policy.hpp): [out | in[2]] || IsZero [out | in | inv]; out = [a == b], the four
// field-element wires are rebuilt by the emitter (the inverse with one exponentiation per emitted wire -- emission is for sampled witnesses)
template <class P> GD B gIsEqualFz(P& p, B z, uint32_t* w0) {              // the block with out = z given; *w0 = its first wire (for iseqf_derived)
    *w0 = p.cur.w;
    BitRef o = p.bits(1); p.dvs(2); BitRef zo = p.bits(1); p.dvs(2);
    return p.put(o, p.put(zo, z));
}
template <class P> GD void iseqf_derived(P& p, uint32_t w0, const F& a, const F& b) {
    const F d = fr_sub(b, a);
    p.derived_fr(w0 + 1, a); p.derived_fr(w0 + 2, b); p.derived_fr(w0 + 4, d); p.derived_fr_inv(w0 + 5, d);
}
template <class P> GD B gIsEqualFd(P& p, const F& a, const F& b) {
    uint32_t w0;
    const B out = gIsEqualFz(p, p.ballot(fr_is_zero(fr_sub(b, a))), &w0);
    if constexpr (P::is_emit) iseqf_derived(p, w0, a, b);
    return out;
}
// LessThan(n)  [out | in[2]] || Num2Bits(n+1)(in0 + 2^n - in1);  out <== 1 - bit n
template <class P> GD B gLessThanS(P& p, int n, S a, S b) {
    BitRef o = p.bits(1); SmRef in = p.sms(2);
    a = p.put(in, a); b = p.put(in + 1, b);
    B top;
    gNum2BitsS(p, n + 1, (S)((uint32_t)a + (1u << n) - (uint32_t)b), &top);
    return p.put(o, ~top);
}
template <class P> GD B gLessThanF(P& p, int n, const F& a, const F& b) {
    BitRef o = p.bits(1); FrRef in = p.frs(2);
    F x = p.put(in, a), y = p.put(in + 1, b);
    Fr e;                                                 // 2^n canonical (n <= 252); no dynamically indexed store: that would put e in scratch
#pragma unroll
    for (int j = 0; j < 8; j++) e.l[j] = (n >> 5) == j ? 1u << (n & 31) : 0u;
    F c;
    gNum2BitsF(p, n + 1, fr_sub(fr_add(x, fr_to_mont(e)), y), nullptr, &c);
    return p.put(o, ~p.ballot(canon_bit(c, n)));
}
// LessEqThan(n) [out | in[2]] || LessThan(n)(in0, in1+1);  GreaterEqThan(n) || LessThan(n)(in1, in0+1)
template <class P> GD B gLessEqThanS(P& p, int n, S a, S b) {
    BitRef o = p.bits(1); SmRef in = p.sms(2);
    a = p.put(in, a); b = p.put(in + 1, b);
    return p.put(o, gLessThanS(p, n, a, (S)((uint32_t)b + 1u)));
}
template <class P> GD B gGreaterEqThanS(P& p, int n, S a, S b) {
    BitRef o = p.bits(1); SmRef in = p.sms(2);
    a = p.put(in, a); b = p.put(in + 1, b);
    return p.put(o, gLessThanS(p, n, b, (S)((uint32_t)a + 1u)));
}
template <class P> GD B gLessEqThanF(P& p, int n, const F& a, const F& b) {
    BitRef o = p.bits(1); FrRef in = p.frs(2);
    F x = p.put(in, a), y = p.put(in + 1, b);
    return p.put(o, gLessThanF(p, n, x, fr_add(y, fr_one_mont())));
}
template <class P> GD B gGreaterEqThanF(P& p, int n, const F& a, const F& b) {
    BitRef o = p.bits(1); FrRef in = p.frs(2);
    F x = p.put(in, a), y = p.put(in + 1, b);
    return p.put(o, gLessThanF(p, n, y, fr_add(x, fr_one_mont())));
}

// MultiAND(n)  [out | in[n]];  n>2: ands[0]=MultiAND(n\2), ands[1]=MultiAND(n-n\2), and2=AND (initialisation order)
template <class P, int N> struct MultiANDg {
    static HD B run(P& p, const B* in) {
        BitRef o = p.bits(1); BitRef i = p.bits(N);
        B v[N];
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = p.put(i + k, in[k]);
        if (p.decl_order) {                                   // and2 = AND() [out | a, b] in front of ands[0], ands[1]
            BitRef a2 = p.bits(3);
            B a = MultiANDg<P, N / 2>::run(p, v);
            B b = MultiANDg<P, N - N / 2>::run(p, v + N / 2);
            a = p.put(a2 + 1, a); b = p.put(a2 + 2, b);
            return p.put(o, p.put(a2, a & b));
        }
        B a = MultiANDg<P, N / 2>::run(p, v);
        B b = MultiANDg<P, N - N / 2>::run(p, v + N / 2);
        return p.put(o, gAND(p, a, b));
    }
};
template <class P> struct MultiANDg<P, 1> {
    static HD B run(P& p, const B* in) { BitRef o = p.bits(1); BitRef i = p.bits(1); B v = p.put(i, in[0]); return p.put(o, v); }
};
template <class P> struct MultiANDg<P, 2> {
    static HD B run(P& p, const B* in) {
        BitRef o = p.bits(1); BitRef i = p.bits(2);
        B a = p.put(i, in[0]), b = p.put(i + 1, in[1]);
        return p.put(o, gAND(p, a, b));
    }
};

// ============================================================================ circomlib: mux1.circom
// Mux1 [out | c[2], s] || MultiMux1(1) [out[1] | c[1][2], s];  out = (c1-c0)*s + c0
template <class P> HD S gMux1S(P& p, S c0, S c1, B s) {
    SmRef o = p.sms(1); SmRef c = p.sms(2); BitRef sr = p.bits(1);
    c0 = p.put(c, c0); c1 = p.put(c + 1, c1); s = p.put(sr, s);
    SmRef mo = p.sms(1); SmRef mc = p.sms(2); BitRef ms = p.bits(1);
    S d0 = p.put(mc, c0), d1 = p.put(mc + 1, c1); B ss = p.put(ms, s);
    return p.put(o, p.put(mo, p.bit(ss) ? d1 : d0));
}
// RlpInteger's use (integer.circom:90): c[0] small, c[1] = the 248-bit input; the result is always < 256
template <class P> GD S gMux1SF(P& p, S c0, const F& c1, B s) {
    SmRef o = p.sms(1); SmRef cs = p.sms(1); FrRef cf = p.frs(1); BitRef sr = p.bits(1);
    c0 = p.put(cs, c0); F f1 = p.put(cf, c1); s = p.put(sr, s);
    SmRef mo = p.sms(1); SmRef mcs = p.sms(1); FrRef mcf = p.frs(1); BitRef ms = p.bits(1);
    S d0 = p.put(mcs, c0); F g1 = p.put(mcf, f1); B ss = p.put(ms, s);
    S lo = (S)fr_from_mont(g1).l[0];
    return p.put(o, p.put(mo, p.bit(ss) ? lo : d0));
}

// ============================================================================ circomlib: compconstant / aliascheck / Num2Bits_strict
// CompConstant(ct = p-1)  [out | in[254] | parts[127], sout] || Num2Bits(135)
template <class P> GD B gCompConstantPm1(P& p, const BV& v, const F& c) {      // v / c: the 254 input bits / their canonical value
    // parts[127] are DERIVED wires (round 4): each is a function of two stored input bits and constants; generation and evaluation only carry their running sum (as 127
    // stored field elements they were the unit's cost: 1 016 limb-row stores in generation, 127 pinned eight-limb compares -- one memory round trip each -- in evaluation)
    BitRef o = p.bits(1); BitRef in = p.bits(254); const uint32_t parts_w = p.dvs(127); FrRef sout = p.frs(1);
    const uint32_t PM1[8] = {0xf0000000u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    bv_put(p, in, 254, v);
    // a = 1 + (2^i - 1), b = 2^128 - 2^i, e = 2^i as plain 160-bit integers (every part is 0, a or b: the four polynomials of compconstant.circom:33-44 evaluated at the
    // two signal bits; the sum stays below 2^136).  Rounds 1-3 carried them as Montgomery field elements through ~4 field additions per part: 508 called additions per
    // CompConstant were most of the 0.2 ms of a Num2BigEndianBytes unit -- the longest unit of two generation levels of the production circuit.
    uint32_t av[5] = {1, 0, 0, 0, 0}, bv[5] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0}, ev[5] = {1, 0, 0, 0, 0}, sv[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        for (int t = 0; t < 16; t++) {
            const int i = 16 * j + t;
            if (i < 127) {
                const uint32_t clsb = (PM1[j] >> (2 * t)) & 1, cmsb = (PM1[j] >> (2 * t + 1)) & 1;
                const bool sl = (c.l[j] >> (2 * t)) & 1, sm = (c.l[j] >> (2 * t + 1)) & 1;
                // which of {0, a, b} the part is
                bool ta, tb;
                if (!cmsb && !clsb) { ta = false; tb = sm || sl; }                                         // -b*sm*sl + b*sm + b*sl
                else if (!cmsb && clsb) { ta = !sm && !sl; tb = sm; }                                      // a*sm*sl - a*sl + b*sm - a*sm + a
                else if (cmsb && !clsb) { ta = !sm; tb = sm && sl; }                                       // b*sm*sl - a*sm + a
                else { ta = !(sm && sl); tb = false; }                                                     // -a*sm*sl + a
                uint32_t pv[5];
#pragma unroll
                for (int q = 0; q < 5; q++) pv[q] = ta ? av[q] : tb ? bv[q] : 0u;
                if constexpr (P::is_emit) {
                    Fr pc = {{pv[0], pv[1], pv[2], pv[3], pv[4], 0, 0, 0}};
                    p.derived_fr(parts_w + (uint32_t)i, fr_to_mont(pc));
                }
                uint32_t cy = 0, bw = 0, cy2 = 0;
#pragma unroll
                for (int q = 0; q < 5; q++) {                                                              // sum += part;  b -= e;  a += e
                    const uint64_t x = (uint64_t)sv[q] + pv[q] + cy; sv[q] = (uint32_t)x; cy = (uint32_t)(x >> 32);
                    const uint64_t y = (uint64_t)bv[q] - ev[q] - bw; bv[q] = (uint32_t)y; bw = (uint32_t)(y >> 63);
                    const uint64_t z = (uint64_t)av[q] + ev[q] + cy2; av[q] = (uint32_t)z; cy2 = (uint32_t)(z >> 32);
                }
#pragma unroll
                for (int q = 4; q > 0; q--) ev[q] = (ev[q] << 1) | (ev[q - 1] >> 31);                      // e *= 2
                ev[0] <<= 1;
            }
        }
    }
    Fr sumc = {{sv[0], sv[1], sv[2], sv[3], sv[4], 0, 0, 0}};
    const F sum = fr_to_mont(sumc);
    F so = p.put(sout, sum), sc;
    gNum2BitsF(p, 135, so, nullptr, &sc);
    return p.put(o, p.ballot(canon_bit(sc, 127)));
}
// AliasCheck [ | in[254]] || CompConstant(-1);  out === 0
template <class P> GD void gAliasCheck(P& p, const BV& v, const F& c) {
    BitRef in = p.bits(254);
    bv_put(p, in, 254, v);
    B gt = gCompConstantPm1(p, v, c);
    p.require(~gt, FAILCODE(T_ALIASCHECK, 31));
}
// Num2Bits_strict [out[254] | in] || Num2Bits(254), AliasCheck   (initialisation order, see oracle header)
template <class P> GD BitRef gNum2BitsStrict(P& p, const F& in, BV* vout = nullptr, F* cout = nullptr) {
    BitRef o = p.bits(254); FrRef i = p.frs(1);
    F x = p.put(i, in);
    BV v; F c;
    if (p.decl_order) {                                       // aliasCheck in front of n2b
        c = fr_from_mont(x);
        v = bv_from_canon(p, c, 254);
        gAliasCheck(p, v, c);
        gNum2BitsF(p, 254, x, &v, &c);
        bv_put(p, o, 254, v);
    } else {
        gNum2BitsF(p, 254, x, &v, &c);
        bv_put(p, o, 254, v);
        gAliasCheck(p, v, c);
    }
    if (vout) *vout = v;
    if (cout) *cout = c;
    return o;
}

// ============================================================================ circomlib: poseidon.circom (optimised schedule)
// table offsets: tools/gen_poseidon.py -> poseidon_consts.h (C | S | M | P per t, Montgomery)
// Poseidon(T-1) [out | inputs[T-1]] || PoseidonEx [out[1] | inputs[T-1], initialState] || ark0; 3x{T Sigma, ark, mix(M)};
// T Sigma, ark4, mix(P); RP x {sigmaP, mixS}; 3x{T Sigma, ark, mix(M)}; T Sigma; mixLast [out | in[T]]
// Every wire is an FR wire, so the block is cut into SEGMENTS at closed-form offsets: head (own + PoseidonEx inputs + ark0),
// 4 + 3 full rounds, the partial rounds in chunks of POS_PCH, and the tail (last S-boxes, mixLast, outputs).  The generator runs
// the segments in sequence with the state in registers; the evaluator runs EACH segment as its own wavefront from the STORED
// output wires of the block before it (every relation of a round is local given stored wires).
struct PosOff { uint32_t C, S, M, Pm; int rp; };
#define POS_PCH 2                                                   // partial rounds per evaluator segment
// Segments: 0 head | 1..8: the four first-half full rounds, each as (Sigma x T + Ark) | (Mix) | 9..: the partial rounds in chunks of POS_PCH | six: the three second-half full
// rounds, two parts each | the tail.  Every segment ends in a block [out[T] | in[T]] (Ark, Mix, MixS), so the state entering the next one is the T stored wires at its
// cursor - 2T.  (Rounds 2-5: whole full rounds and chunks of 4 -- 40 and 48 Montgomery products per wavefront, the longest units of the narrow evaluation kernel; now 25.)
HD uint32_t pos_nchunks(int rp) { return (uint32_t)(rp + POS_PCH - 1) / POS_PCH; }
HD uint32_t pos_nseg(int rp) { return 1 + 8 + pos_nchunks(rp) + 6 + 1; }
HD uint32_t pos_wires(int T, int rp) { return (4 * T + 1) + 7 * 8 * T + (uint32_t)rp * (4 + 2 * T) + (5 * T + 1); }
// FR-wire offset of segment `seg` inside the Poseidon block
HD uint32_t pos_seg_off(int T, int rp, uint32_t seg) {
    const uint32_t nch = pos_nchunks(rp), head = 4 * T + 1, full = 8 * T, part = 4 + 2 * T, partA = 6 * T;
    if (seg == 0) return 0;
    if (seg < 9) return head + ((seg - 1) >> 1) * full + (((seg - 1) & 1) ? partA : 0);
    if (seg < 9 + nch) return head + 4 * full + (seg - 9) * POS_PCH * part;
    const uint32_t q = seg - 9 - nch;                                              // second-half full rounds (q < 6), then the tail (q = 6)
    return head + 4 * full + (uint32_t)rp * part + (q >> 1) * full + ((q & 1) ? partA : 0);
}
// A state element is a VALUE for generation / counting / emission.  The evaluator never holds the state in registers: there a state
// element is a STORED WIRE (+ an optional round constant still to be added) that is loaded where a relation uses it, so a segment
// needs a dozen live VGPRs besides the Montgomery multiplier's own (holding T elements across the multiplier calls spilled).
struct PosLazy { FrRef r; uint32_t c; };                 // stored[r] + (c != ~0u ? kconst(c) : 0)
template <class P> struct PosSt { typedef typename std::conditional<P::is_check, PosLazy, F>::type type; };
template <class P> HD F pv_get(P&, const F& v) { return v; }
template <class P> HD F pv_get(P& p, const PosLazy& v) { const F x = p.get(v.r); return v.c != 0xFFFFFFFFu ? fr_add(x, p.kconst(v.c)) : x; }
template <class P> HD typename PosSt<P>::type pv_at(P&, FrRef r, const F& val) {      // the wire r, whose value is val
    if constexpr (P::is_check) { (void)val; return PosLazy{r, 0xFFFFFFFFu}; } else { (void)r; return val; }
}
template <class P> HD typename PosSt<P>::type pv_addc(P& p, const typename PosSt<P>::type& v, uint32_t cidx) {
    if constexpr (P::is_check) { (void)p; return PosLazy{v.r, cidx}; } else return fr_add(v, p.kconst(cidx));
}
template <class P> HD typename PosSt<P>::type gSigma(P& p, const typename PosSt<P>::type& in) {   // [out | in | in2, in4]
    FrRef o = p.frs(1), i = p.frs(1), m = p.frs(2);
    const F x = p.put(i, pv_get(p, in));
    const F x2 = p.put(m, fr_sqr(x));
    const F x4 = p.put(m + 1, fr_sqr(x2));
    return pv_at(p, o, p.put(o, fr_mul(x4, x)));
}
template <class P, int T> GD void gArk(P& p, const PosOff& k, int r, typename PosSt<P>::type* st) {   // [out[t] | in[t]]
    FrRef o = p.frs(T), i = p.frs(T);
#pragma unroll
    for (int j = 0; j < T; j++) {
        const F x = p.put(i + j, pv_get(p, st[j]));
        st[j] = pv_at(p, o + j, p.put(o + j, fr_add(x, p.kconst(k.C + r + j))));
    }
}
template <class P, int T> GD void gMix(P& p, uint32_t mat, typename PosSt<P>::type* st) {             // out[i] = sum_j A[i][j] in[j]
    FrRef o = p.frs(T), i = p.frs(T);
    typename PosSt<P>::type in[T];
#pragma unroll
    for (int j = 0; j < T; j++) in[j] = pv_at(p, i + j, p.put(i + j, pv_get(p, st[j])));
#pragma unroll
    for (int a = 0; a < T; a++) {
        F acc = fr_zero();
#pragma unroll
        for (int j = 0; j < T; j++) acc = fr_add(acc, fr_mul(p.kconst(mat + a * T + j), pv_get(p, in[j])));
        st[a] = pv_at(p, o + a, p.put(o + a, acc));
    }
}
template <class P, int T> GD void gMixS(P& p, const PosOff& k, int r, typename PosSt<P>::type* st) {
    FrRef o = p.frs(T), i = p.frs(T);
    typename PosSt<P>::type in[T];
#pragma unroll
    for (int j = 0; j < T; j++) in[j] = pv_at(p, i + j, p.put(i + j, pv_get(p, st[j])));
    const uint32_t base = k.S + (2 * T - 1) * r;
    F acc = fr_zero();
#pragma unroll
    for (int j = 0; j < T; j++) acc = fr_add(acc, fr_mul(p.kconst(base + j), pv_get(p, in[j])));
    st[0] = pv_at(p, o, p.put(o, acc));
#pragma unroll
    for (int j = 1; j < T; j++) st[j] = pv_at(p, o + j, p.put(o + j, fr_add(pv_get(p, in[j]), fr_mul(pv_get(p, in[0]), p.kconst(base + T + j - 1)))));
}
// one full round in two parts: T Sigma + ark (constants at C + cr) | mix with matrix `mat`
template <class P, int T> GD void gPosFullPart(P& p, const PosOff& k, int cr, uint32_t mat, uint32_t part, typename PosSt<P>::type* st) {
    if (part == 0) {
#pragma unroll
        for (int j = 0; j < T; j++) st[j] = gSigma(p, st[j]);
        gArk<P, T>(p, k, cr, st);
    } else gMix<P, T>(p, mat, st);
}
template <class P, int T> GD void gPosPartial(P& p, const PosOff& k, int r, typename PosSt<P>::type* st) {
    st[0] = pv_addc(p, gSigma(p, st[0]), k.C + 5 * T + r);
    gMixS<P, T>(p, k, r, st);
}
// segment `seg` of the Poseidon block at `base` (p.cur must stand at the segment's first wire); st = the state entering it
// (seg 0: st[1..T-1] = the caller's inputs, as values)
template <class P, int T> GD void gPoseidonSeg0(P& p, const PosOff& k, const F* inputs, typename PosSt<P>::type* st) {
    FrRef in = p.frs(T - 1);                                       // (own `out`: declared by the caller, written / checked by the tail)
    p.frs(1);                                                      // PoseidonEx.out
    FrRef ein = p.frs(T - 1), einit = p.frs(1);
#pragma unroll
    for (int j = 1; j < T; j++) { const F x = p.put(in + (j - 1), inputs[j - 1]); st[j] = pv_at(p, ein + (j - 1), p.put(ein + (j - 1), x)); }
    st[0] = pv_at(p, einit, p.put(einit, fr_zero()));
    gArk<P, T>(p, k, 0, st);
}
template <class P, int T> GD void gPoseidonSeg(P& p, const PosOff& k, Cur base, uint32_t seg, typename PosSt<P>::type* st) {
    const uint32_t nch = pos_nchunks(k.rp);
    if (seg < 9) { const uint32_t r = (seg - 1) >> 1; gPosFullPart<P, T>(p, k, (int)(r + 1) * T, r == 3 ? k.Pm : k.M, (seg - 1) & 1, st); }
    else if (seg < 9 + nch) {
        const int r0 = (int)(seg - 9) * POS_PCH, r1 = r0 + POS_PCH < k.rp ? r0 + POS_PCH : k.rp;
        for (int r = r0; r < r1; r++) gPosPartial<P, T>(p, k, r, st);
    } else if (seg < 9 + nch + 6) { const uint32_t q = seg - 9 - nch; gPosFullPart<P, T>(p, k, 5 * T + k.rp + (int)(q >> 1) * T, k.M, q & 1, st); }
    else {
#pragma unroll
        for (int j = 0; j < T; j++) st[j] = gSigma(p, st[j]);
        FrRef lo = p.frs(1), li = p.frs(T);
        F acc = fr_zero();
#pragma unroll
        for (int j = 0; j < T; j++) acc = fr_add(acc, fr_mul(p.kconst(k.M + j), p.put(li + j, pv_get(p, st[j]))));
        const F h = p.put(lo, acc);
        const FrRef o = {base.w, base.f}, eo = {base.w + (uint32_t)T, base.f + (uint32_t)T};
        st[0] = pv_at(p, o, p.put(o, p.put(eo, h)));
    }
}
template <class P, int T> GD F gPoseidon(P& p, const PosOff& k, const F* inputs) {
    const Cur base = p.cur;
    const FrRef o = p.frs(1);                                      // out
    typename PosSt<P>::type st[T];
    gPoseidonSeg0<P, T>(p, k, inputs, st);
    const uint32_t ns = pos_nseg(k.rp);
    for (uint32_t seg = 1; seg < ns; seg++) gPoseidonSeg<P, T>(p, k, base, seg, st);
    (void)o;
    return pv_get(p, st[0]);
}
// the evaluator's segment unit: the state entering segment seg >= 1 is the stored out[T] of the block before it
// (ark0 / mix / mixS, all [out[T] | in[T]]: the T wires at cursor - 2T)
template <class P, int T> GD void gPoseidonSegStored(P& p, const PosOff& k, Cur base, uint32_t seg) {
    const uint32_t off = pos_seg_off(T, k.rp, seg);
    p.cur = Cur{base.w + off, base.b, base.s, base.f + off, base.q};
    typename PosSt<P>::type st[T];
#pragma unroll
    for (int j = 0; j < T; j++) { const FrRef r = {p.cur.w - 2 * T + j, p.cur.f - 2 * T + j}; st[j] = pv_at(p, r, p.get(r)); }
    gPoseidonSeg<P, T>(p, k, base, seg, st);
}

// Poseidon as a composite unit sees it.  The block never runs inside the composite on the device:
//   * generation runs it as a U_POS_WIDE unit of its own, one stage EARLIER, with the state spread over lanes (poseidon_wide.hpp):
//     the composite continues from the block's stored output wire.  `src` says which wires feed the block (FR ranks of wires written
//     in earlier stages) and where the caller's copy of the hash goes;
//   * the evaluator checks the head (the inputs) here, leaves segments 1.. to CK_POS_SEG units and continues from the stored output;
//   * counting and emission walk the whole block.
#define POS_NONE 0xFFFFFFFFu
struct PosSrc { uint32_t pre, in[3], sub, also; };   // input 0 = POSEIDON_PREFIX + pre; in[] = inputs 1.. ; the LAST input is in - sub; also = copy of out
template <class P, int T> GD F gPoseidonU(P& p, const PosOff& k, const F* inputs, const PosSrc& src) {
    if constexpr (P::is_count) p.note(NOTE_POSEIDON, (uint32_t)T, p.cur, src.pre, src.in[0], src.in[1], src.in[2], src.sub, src.also);
    if constexpr (P::is_check || P::is_gen) {
        const Cur base = p.cur;
        if constexpr (P::is_check) {
            p.frs(1);
            typename PosSt<P>::type st[T];
            gPoseidonSeg0<P, T>(p, k, inputs, st);
        }
        const uint32_t n = pos_wires(T, k.rp);
        p.cur = Cur{base.w + n, base.b, base.s, base.f + n, base.q};
        return p.get(FrRef{base.w, base.f});
    } else return gPoseidon<P, T>(p, k, inputs);
}

// ============================================================================ circuits/utils/assert.circom
// AssertBits(B) :13-17  [ | in | bits[B]] || Num2Bits(B)
template <class P> GD void gAssertBitsS(P& p, int nb, S in) {
    SmRef i = p.sms(1); BitRef bits = p.bits(nb);
    in = p.put(i, in);
    gNum2BitsS(p, nb, in, nullptr, &bits);
}
template <class P> GD void gAssertBitsF(P& p, int nb, const F& in) {
    FrRef i = p.frs(1); BitRef bits = p.bits(nb);
    F x = p.put(i, in);
    BV v;
    gNum2BitsF(p, nb, x, &v);
    bv_put(p, bits, nb, v);
}
#define FP_ABITS8 (Cur{18, 0, 0, 0, 18})     // AssertBits(8) [in | bits[8]] + Num2Bits(8) [out[8] | in]: every wire is a function of the byte -> DERIVED (round 4: the 16 bits too; rounds 1-3 stored them)
// ---------------------------------------------------------------------------- AssertByteString(N) in ranges (assert.circom:26-31)
// own in[N] (first wire own_w) is declared by the caller; child i (AssertBits(8)) lives at c0 + i*FP_ABITS8.  Everything is a function of the source
// byte and DERIVED (policy.hpp): the three copies of the byte -- own in[i], AssertBits.in, Num2Bits.in -- and, since round 4, the two copies of its 8
// bits: generation and evaluation only hold the assert (byte < 256), the emitter writes the 19 wires per byte
template <class P> GD void abs_range(P& p, Cur c0, uint32_t own_w, SmRef src, uint32_t lo, uint32_t hi) {
    // per byte i: own in[i]; AssertBits(8) [in | bits[8]] || Num2Bits(8) [out[8] | in] at c0 + i*FP_ABITS8.  The 16 bit wires of a
    // byte are consecutive wires (bits[8] then out[8]), so 4 bytes make one lane-distributed run of 64 wires for the emitter.
    const uint32_t ln = p.lane_id();
    for (uint32_t i0 = lo; i0 < hi; i0 += 8) {
        const uint32_t cnt = hi - i0 < 8 ? hi - i0 : 8;
        B compact = 0, wc = 0;                           // lane 8t + k = bit k of byte i0 + t (wc: this witness' eight bytes, transposed below)
        S vs[8];                                         // the eight bytes' loads in flight together (a short last batch repeats its last byte)
#pragma unroll
        for (uint32_t t = 0; t < 8; t++) vs[t] = p.get(src + (i0 + (t < cnt ? t : cnt - 1)));
#pragma unroll
        for (uint32_t t = 0; t < 8; t++) if (t < cnt) {
            const uint32_t i = i0 + t;
            const Cur c = cur_add(c0, FP_ABITS8, i);
            const S v = vs[t];
            p.derived(own_w + i, v); p.derived(c.w, v); p.derived(c.w + 17, v);
            p.require(p.ballot((uint32_t)v < 256u), FAILCODE(T_NUM2BITS, 38));
            if constexpr (P::is_emit) wc |= (B)((uint32_t)v & 0xffu) << (8 * t);
        }
        if constexpr (P::is_emit) {
            compact = p.xpose(wc, 64);
            for (uint32_t h2 = 0; h2 < 2 && 4 * h2 < cnt; h2++) {       // bytes i0 + 4*h2 .. +3: 16 wires each (bits[8] then out[8])
                const uint32_t nb = cnt - 4 * h2 < 4 ? cnt - 4 * h2 : 4;
                const uint32_t t = 4 * h2 + (ln >> 4), q = ln & 15, i = i0 + t;
                const B x = p.run_perm(compact, 8 * t + (q & 7));
                p.run_derived(16 * nb, c0.w + 18 * i + 1 + q, x);
            }
        } else { (void)compact; (void)wc; (void)ln; }
    }
}

// AssertByteString(N) :26-31  [ | in[N]] || AssertBits(8) x N
template <class P> GD void gAssertByteString(P& p, int N, SmRef src) {
    const uint32_t in_w = p.dvs(N);
    const Cur c0 = p.cur;
    abs_range(p, c0, in_w, src, 0, (uint32_t)N);
    p.cur = cur_add(c0, FP_ABITS8, (uint32_t)N);
}
// AssertLessThan(B) :40-47 / AssertLessEqThan :56-63 / AssertGreaterEqThan :72-79   [ | a, b | out]; out === 1
template <class P> GD void gAssertLessThanS(P& p, int nb, S a, S b) {
    SmRef in = p.sms(2); BitRef o = p.bits(1);
    a = p.put(in, a); b = p.put(in + 1, b);
    gAssertBitsS(p, nb, a); gAssertBitsS(p, nb, b);
    p.require(p.put(o, gLessThanS(p, nb, a, b)), FAILCODE(T_ASSERT_LT, 46));
}
template <class P> GD void gAssertLessEqThanS(P& p, int nb, S a, S b) {
    SmRef in = p.sms(2); BitRef o = p.bits(1);
    a = p.put(in, a); b = p.put(in + 1, b);
    gAssertBitsS(p, nb, a); gAssertBitsS(p, nb, b);
    p.require(p.put(o, gLessEqThanS(p, nb, a, b)), FAILCODE(T_ASSERT_LE, 62));
}
template <class P> GD void gAssertGreaterEqThanS(P& p, int nb, S a, S b) {
    SmRef in = p.sms(2); BitRef o = p.bits(1);
    a = p.put(in, a); b = p.put(in + 1, b);
    gAssertBitsS(p, nb, a); gAssertBitsS(p, nb, b);
    p.require(p.put(o, gGreaterEqThanS(p, nb, a, b)), FAILCODE(T_ASSERT_GE, 78));
}
template <class P> GD void gAssertLessEqThanF(P& p, int nb, const F& a, const F& b) {
    FrRef in = p.frs(2); BitRef o = p.bits(1);
    F x = p.put(in, a), y = p.put(in + 1, b);
    gAssertBitsF(p, nb, x); gAssertBitsF(p, nb, y);
    p.require(p.put(o, gLessEqThanF(p, nb, x, y)), FAILCODE(T_ASSERT_LE, 62));
}
template <class P> GD void gAssertGreaterEqThanF(P& p, int nb, const F& a, const F& b) {
    FrRef in = p.frs(2); BitRef o = p.bits(1);
    F x = p.put(in, a), y = p.put(in + 1, b);
    gAssertBitsF(p, nb, x); gAssertBitsF(p, nb, y);
    p.require(p.put(o, gGreaterEqThanF(p, nb, x, y)), FAILCODE(T_ASSERT_GE, 78));
}

// ============================================================================ circuits/utils/array.circom
template <class P> HD void iseq_derived_w(P& p, uint32_t w, S a, S b);
// Filter(N) :26-39  [out[N] | in | isEq[N]] || IsEqual([i, in]) x N;  out[i] = out[i-1]*(1-isEq[i])
// isEq[k] = [in == k], out[k] = prod_{j <= k}(1 - isEq[j]) = [in > k] (unsigned: a value outside 0..N-1 never hits): both are written / verified as lane-distributed
// runs against those functions of the STORED `in` -- no wire is read back (as a loop of single puts, each of the caller's reads of out[k] waited for the stores before
// it: 2-5 us per wire under load, the pole of the RLP assembly units).  The IsEqual children are derived wires (their outputs are copies of isEq[k]).
// oruns (N <= 256): the out[] bits as runs of 64 (lane t of oruns[r] = wire 64 r + t), for a caller that copies them (Mask)
template <class P> GD BitRef gFilter(P& p, int N, S in, B* oruns = nullptr) {
    BitRef o = p.bits(N); SmRef i = p.sms(1); BitRef isEq = p.bits(N);
    in = p.put(i, in);
    const uint32_t kids_w = p.dvs(6u * (uint32_t)N), ln = p.lane_id();
    if constexpr (P::is_check) {        // the evaluator: no stores in front of its loads, the plain loop of single wires pipelines (as runs the RLP leaf tail took 0.35 ms, so 0.25)
        (void)kids_w; (void)ln;
        B prev = ~(B)0;
        for (int k = 0; k < N; k++) {
            const B e = p.put(isEq + (uint32_t)k, p.ballot((uint32_t)in == (uint32_t)k));
            prev = p.put(o + (uint32_t)k, prev & ~e);
        }
        return o;
    }
    B keep0 = 0, keep1 = 0, keep2 = 0, keep3 = 0;               // (four scalars selected by a uniform compare chain: ONE loop body -- unrolled eight times the unit outgrew the
                                                                 //  instruction cache -- and no dynamically indexed register array)
    for (uint32_t k0 = 0; k0 < (uint32_t)N; k0 += 32) {
        const uint32_t n = (uint32_t)N - k0 < 32 ? (uint32_t)N - k0 : 32, r = k0 >> 6, l0 = k0 & 63;
        // this witness' bits of the 32 entries: isEq[k0 + t] = [in == k0 + t], out[k0 + t] = [in > k0 + t] (unsigned); transposed into the wires' runs
        const uint32_t d = (uint32_t)in - k0, nmask = n < 32 ? (1u << n) - 1u : 0xFFFFFFFFu;
        const bool above = (uint32_t)in >= k0;
        const uint32_t wE = (above && d < n) ? 1u << d : 0u;
        const uint32_t wO = !above ? 0u : d >= 32 ? nmask : ((1u << d) - 1u) & nmask;
        const B wK = (above && d < n) ? (B)3 << (2 * d) : 0;
        const B runE = p.xpose64(wE, 0, n), runO = p.xpose64(wO, 0, n), runK = p.xpose(wK, 2 * n);
        const B kp = oruns ? p.xpose((B)wO << l0, 64) : 0;
        if (r == 0) keep0 |= kp; else if (r == 1) keep1 |= kp; else if (r == 2) keep2 |= kp; else keep3 |= kp;
        p.run_put(n, isEq.w + k0 + ln, isEq.i + k0 + ln, runE);
        p.run_put(n, o.w + k0 + ln, o.i + k0 + ln, runO);
        p.run_derived(2 * n, kids_w + 6 * (k0 + (ln >> 1)) + 3 * (ln & 1), runK);      // IsEqual.out / IsZero.out of child k: copies of isEq[k]
        if constexpr (P::is_emit) {
            if (ln < 2 * n) p.site_c(kids_w + 6 * (k0 + (ln >> 1)) + 3 * (ln & 1), isEq.w + k0 + (ln >> 1));
            for (uint32_t t = 0; t < n; t++) iseq_derived_w(p, kids_w + 6 * (k0 + t), (S)(k0 + t), in);
        }
    }
    if (oruns) { oruns[0] = keep0; oruns[1] = keep1; oruns[2] = keep2; oruns[3] = keep3; }
    return o;
}
// Fit(M,N) :47-57  [out[N] | in[M]]
template <class P> GD SmRef gFitS(P& p, int M, int N, SmRef src) {
    SmRef o = p.sms(N), in = p.sms(M);
    copy_n(p, in, src, M); copy_n(p, o, src, M < N ? M : N);
    for (int i = M; i < N; i++) p.put(o + i, 0);
    return o;
}
// Flatten(M,N) :64-72 and Reshape(M,N) :79-87 are the identity on row-major data  [out[MN] | in[MN]]
template <class P> GD SmRef gFlattenS(P& p, int n, SmRef src) {
    SmRef o = p.sms(n), in = p.sms(n);
    { copy_n(p, in, src, (int)(n)); copy_n(p, o, src, (int)(n)); }
    return o;
}
// Reverse(N) :94-100  [out[N] | in[N]]
template <class P> GD SmRef gReverseS(P& p, int N, SmRef src) {
    SmRef o = p.sms(N), in = p.sms(N);
    copy_n(p, in, src, N); sm_copy<P, 8>(p, o, src, N, true);
    return o;
}

// ============================================================================ circuits/utils/divide.circom:17-33
// [out, rem | a, b] || AssertLessThan(N)(rem, b), AssertLessEqThan(N)(out, a);  out*b + rem === a
template <class P> GD void gDivide(P& p, int N, S a, S b, S& q, S& r) {
    SmRef o = p.sms(2), in = p.sms(2);
    a = p.put(in, a); b = p.put(in + 1, b);
    uint32_t ua = (uint32_t)a, ub = (uint32_t)b;
    q = p.hint(o, (S)(ub ? ua / ub : 0));
    r = p.hint(o + 1, (S)(ub ? ua % ub : 0));
    gAssertLessThanS(p, N, r, b);
    gAssertLessEqThanS(p, N, q, a);
    p.require(p.ballot((int64_t)q * b + r == (int64_t)a), FAILCODE(T_DIVIDE, 32));
}

// ============================================================================ circuits/utils/selector.circom
// Selector(N) block at cursor c (selector.circom:21-46): [out | vals[N], select | isEq[N], sum[N+1]] || IsEqual([select, i]) x N;  sum isEq === 1.
// STORED: out (one SM wire) and the 3N bits (isEq[], the children's IsEqual.out / IsZero.out).  DERIVED (policy.hpp): vals[] (copies of the
// source), select, sum[i+1] = [select <= i] * out, the children's operand wires -- the emitter rebuilds them.
struct SelBlk { SmRef o; uint32_t vals_w, sel_w, sum_w; BitRef isEq; Cur kids; };
HD SelBlk sel_blk(Cur c, uint32_t N) {
    SelBlk s;
    s.o = SmRef{c.w, c.s}; s.vals_w = c.w + 1; s.sel_w = c.w + 1 + N;
    s.isEq = BitRef{c.w + 2 + N, c.b}; s.sum_w = c.w + 2 + 2 * N;
    s.kids = Cur{c.w + 3 * N + 3, c.b + N, c.s + 1, c.f, c.q + 2 * N + 2};      // (the children store nothing: their two outputs are copies of isEq[i])
    return s;
}
HD Cur sel_fp(uint32_t N) { Cur r = {9 * N + 3, N, 1, 0, 8 * N + 2}; return r; }      // stored: out (SM) and isEq[N]; vals / select / sum / the IsEqual children: derived
#define FP_ISEQ_S_ (Cur{6, 2, 0, 0, 4})       // IsEqual [out | in[2]] + IsZero [out | in | inv]: two BIT outputs, four derived operand wires
// four derived operand wires of an IsEqual([a, b]) child whose first wire is w
template <class P> HD void iseq_derived_w(P& p, uint32_t w, S a, S b) {
    p.derived(w + 1, a); p.derived(w + 2, b);
    const S x = (S)((uint32_t)b - (uint32_t)a);
    p.derived(w + 4, x); p.derived_inv(w + 5, x, true);
}
// entries [lo, hi) of the Selector block sb over src[i * stride]: the bits as lane-distributed runs (32 entries: isEq[], and the children's two
// outputs each as one run of 64), the derived wires for the emitter.  out = the selected value (src[select]).
template <class P> GD void sel_range(P& p, const SelBlk& sb, SmRef src, uint32_t stride, S select, S out, uint32_t lo, uint32_t hi) {
    const uint32_t ln = p.lane_id();
    for (uint32_t i0 = lo; i0 < hi; i0 += 32) {
        const uint32_t n = hi - i0 < 32 ? hi - i0 : 32;
        // this witness' bits of the 32 entries (isEq[i0 + t] = [select == i0 + t]; the children's two outputs: two bits per entry), transposed into the wires' runs
        const uint32_t d = (uint32_t)select - i0;
        const bool hit = (uint32_t)select >= i0 && d < n;
        const B runE = p.xpose64(hit ? 1u << d : 0u, 0, n), runK = p.xpose(hit ? (B)3 << (2 * d) : 0, 2 * n);
        p.run_put(n, sb.isEq.w + i0 + ln, sb.isEq.i + i0 + ln, runE);
        p.run_derived(2 * n, sb.kids.w + 6 * (i0 + (ln >> 1)) + 3 * (ln & 1), runK);      // IsEqual.out / IsZero.out of child i: copies of isEq[i]
        if constexpr (P::is_emit) { if (ln < 2 * n) p.site_c(sb.kids.w + 6 * (i0 + (ln >> 1)) + 3 * (ln & 1), sb.isEq.w + i0 + (ln >> 1)); }
        if constexpr (P::is_emit) {
            for (uint32_t t = 0; t < n; t++) {
                const uint32_t i = i0 + t;
                p.derived(sb.vals_w + i, p.get(src + i * stride));
                iseq_derived_w(p, sb.kids.w + 6 * i, select, (S)i);
                p.derived(sb.sum_w + i + 1, (uint32_t)select <= i ? out : 0);
            }
        }
    }
}
// the head of a Selector block: out, select, sum[0], the range check (sum isEq === 1  <=>  0 <= select < N); returns the selected value
template <class P> GD S sel_head(P& p, const SelBlk& sb, uint32_t N, SmRef src, uint32_t stride, S select) {
    const uint32_t us = (uint32_t)select;
    p.require(p.ballot(us < N), FAILCODE(T_SELECTOR, 43));
    p.derived(sb.sel_w, select); p.derived(sb.sum_w, 0);
    return p.put(sb.o, us < N ? p.get_lane(src, us * stride) : 0);
}
// Selector(n) :21-46 over src[i * stride]
template <class P> GD S gSelectorS(P& p, int n, SmRef src, S select, uint32_t stride = 1) {
    const SelBlk sb = sel_blk(p.cur, (uint32_t)n);
    p.cur = cur_add(p.cur, sel_fp((uint32_t)n), 1);
    const S out = sel_head(p, sb, (uint32_t)n, src, stride, select);
    sel_range(p, sb, src, stride, select, out, 0, (uint32_t)n);
    return out;
}
// SelectorArray1D(n, q) :62-77  [out[q] | arrays[n][q], select | arraysT[q][n]] || Selector(n) x q.  arrays[][], select and the transposed
// copy are derived wires (copies of the source); selector j reads its column of the source directly (stride q)
template <class P> GD SmRef gSelectorArray1D(P& p, int n, int q, SmRef src, S select) {
    SmRef o = p.sms(q); const uint32_t arr_w = p.dvs(n * q), sel_w = p.dvs(1), T_w = p.dvs(q * n);
    if constexpr (P::is_emit) {
        p.derived(sel_w, select);
        for (int i = 0; i < n; i++) for (int j = 0; j < q; j++) { const S v = p.get(src + (i * q + j)); p.derived(arr_w + i * q + j, v); p.derived(T_w + j * n + i, v); }
    }
    for (int j = 0; j < q; j++) p.put(o + j, gSelectorS(p, n, src + j, select, (uint32_t)q));
    return o;
}

// ============================================================================ circuits/utils/shift.circom
// ShiftLeft(n) :17-37  [out[n] | in[n], count | isEq[n][n], temp[n][n]] || AssertLessEqThan(16)(count, n), IsEqual([i, j-count]) x n^2
// `split`: the caller is a composite unit of the device plan -- the evaluator runs the n rows (n^2 IsEqual gadgets: almost all of the
// block) as CK_SL_ROWS units of their own and the composite only steps over them
template <class P> GD SmRef gShiftLeft(P& p, int n, SmRef src, S count, bool split = false) {
    const Cur blk = p.cur;
    SmRef o = p.sms(n), in = p.sms(n), cn = p.sms(1); BitRef isEq = p.bits(n * n); const uint32_t temp_w = p.dvs(n * n);      // temp[i][j] = isEq[i][j] * in[j]: DERIVED (round 4)
    count = p.put(cn, count);
    copy_n(p, in, src, (int)(n));
    gAssertLessEqThanS(p, 16, count, (S)n);
    if constexpr (P::is_count) { if (split) p.note(NOTE_SHIFTLEFT, (uint32_t)n, blk, p.cur.w, p.cur.b, p.cur.s); }
    if constexpr (P::is_check) {
        if (split) { p.cur = cur_add(p.cur, FP_ISEQ_S_, (uint32_t)(n * n)); return o; }
    }
    if constexpr (P::is_gen) {
        // generation: the n^2 IsEqual outputs are stores only (no in[j] load behind them: 961 memory round trips for n = 31), out[i] = in[i + count] is a per-witness gather
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) p.put(isEq + (i * n + j), gIsEqualS(p, (S)i, (S)(j - count)));
        for (int i0 = 0; i0 < n; i0 += 8) {
            S v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = i0 + q < n ? i0 + q : n - 1;
                const uint32_t idx = (uint32_t)(i + count);
                v[q] = p.get_lane(src, idx < (uint32_t)n ? idx : 0u);
                if (idx >= (uint32_t)n) v[q] = 0;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) if (i0 + q < n) p.put(o + (uint32_t)(i0 + q), v[q]);
        }
        return o;
    }
    for (int i = 0; i < n; i++) {
        S acc = 0;
        for (int j = 0; j < n; j++) {
            B e = p.put(isEq + (i * n + j), gIsEqualS(p, (S)i, (S)(j - count)));
            const S tv = p.bit(e) ? p.get(in + j) : 0;
            p.derived(temp_w + (uint32_t)(i * n + j), tv);
            acc += tv;
        }
        p.put(o + i, acc);
    }
    return o;
}
// rows [i0, i1) of the ShiftLeft(n) block at `blk` (IsEqual children from `ciseq` on), from the stored count / in[]
template <class P> GD void gShiftLeftRows(P& p, int n, Cur blk, Cur ciseq, uint32_t i0, uint32_t i1) {
    const SmRef o = {blk.w, blk.s}, in = o + (uint32_t)n, cn = in + (uint32_t)n;
    const BitRef isEq = {cn.w + 1, blk.b}; const uint32_t temp_w = isEq.w + (uint32_t)(n * n);
    const S count = p.get(cn);
    for (uint32_t i = i0; i < i1; i++) {
        S acc = 0;
        p.cur = cur_add(ciseq, FP_ISEQ_S_, i * (uint32_t)n);
        for (uint32_t j = 0; j < (uint32_t)n; j++) {
            B e = p.put(isEq + (i * n + j), gIsEqualS(p, (S)i, (S)((S)j - count)));
            const S tv = p.bit(e) ? p.get(in + j) : 0;
            p.derived(temp_w + i * (uint32_t)n + j, tv);
            acc += tv;
        }
        p.put(o + i, acc);
    }
}
// ShiftRight(n, ms) :51-75  [out[n+ms] | in[n], count | isEq[ms+1], temps[ms+1][n]] || AssertLessEqThan(16)(count, ms), IsEqual([i, count]) x (ms+1)
// `split`: the caller is a composite unit of the device plan -- the evaluator runs the (ms+1) x n temps[][] (almost all of the block) as
// CK_SR_COLS units of their own
template <class P> GD SmRef gShiftRight(P& p, int n, int ms, SmRef src, S count, bool split = false) {
    (void)split;       // (rounds 2-3: the evaluator ran the stored temps[][] as CK_SR_COLS units of their own; round 4: temps[i][j] = isEq[i] * in[j] are DERIVED wires, nothing to evaluate)
    SmRef o = p.sms(n + ms), in = p.sms(n), cn = p.sms(1); BitRef isEq = p.bits(ms + 1); const uint32_t temps_w = p.dvs((ms + 1) * n);
    count = p.put(cn, count);
    copy_n(p, in, src, (int)(n));
    gAssertLessEqThanS(p, 16, count, (S)ms);
    for (int i = 0; i <= ms; i++) p.put(isEq + i, gIsEqualS(p, (S)i, count));      // isEq[i] = [i == count]
    // temps[i][j] = isEq[i] * in[j]: the emitter rebuilds the (ms + 1) x n products from the stored count and bytes
    if constexpr (P::is_emit) {
        for (int j0 = 0; j0 < n; j0 += 8) {
            S v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = p.get(src + (uint32_t)(j0 + q < n ? j0 + q : n - 1));
            for (int i = 0; i <= ms; i++) {
                const bool hit = (uint32_t)i == (uint32_t)count;
#pragma unroll
                for (int q = 0; q < 8; q++) if (j0 + q < n) p.derived(temps_w + (uint32_t)(i * n + j0 + q), hit ? v[q] : 0);
            }
        }
    }
    // out[t] = sum_{i+j=t} temps[i][j] = in[t - count] when 0 <= t - count < n (exactly one isEq is set), read per witness
    const uint32_t cnt = (uint32_t)count;
    for (int t0 = 0; t0 < n + ms; t0 += 8) {
        S v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t t = (uint32_t)(t0 + q < n + ms ? t0 + q : n + ms - 1);
            const bool inside = cnt <= (uint32_t)ms && t >= cnt && t - cnt < (uint32_t)n;
            v[q] = p.get_lane(src, inside ? t - cnt : 0u);
            if (!inside) v[q] = 0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) if (t0 + q < n + ms) p.put(o + (uint32_t)(t0 + q), v[q]);
    }
    return o;
}
// Mask(n) :18-30  [out[n] | in[n], count | filter[n]] || Filter(n)
template <class P> GD SmRef gMask(P& p, int n, SmRef src, S count) {
    SmRef o = p.sms(n), in = p.sms(n), cn = p.sms(1); BitRef flt = p.bits(n);
    count = p.put(cn, count);
    B fr[4] = {0, 0, 0, 0};
    const BitRef f = gFilter(p, n, count, fr);                   // n <= 256
    const uint32_t ln = p.lane_id();
    // (filter[] <== Filter.out, in[] <== src[], out = filter * in.  The evaluator: all on STORED wires, plain loop.  n > 256 -- gadget mains only, the production
    //  circuits stay below 105 -- takes the same per-wire loop in every policy: the lane-distributed runs below carry 4 x 64 filter bits)
    if (P::is_check || n > 256) {
        (void)ln;
        for (int i = 0; i < n; i++) {
            const B fb = p.put(flt + (uint32_t)i, p.get(f + (uint32_t)i));
            const S v = p.put(in + (uint32_t)i, p.get(src + (uint32_t)i));
            p.put(o + (uint32_t)i, p.bit(fb) ? v : 0);
        }
        return o;
    }
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) {
        const uint32_t k0 = 64 * r;
        if (k0 < (uint32_t)n) p.run_put((uint32_t)n - k0 < 64 ? (uint32_t)n - k0 : 64, flt.w + k0 + ln, flt.i + k0 + ln, fr[r]);      // filter[] <== Filter.out (from the values, not read back)
    }
    // in[] <== src[], out[i] <== filter[i] * in[i].  Generation / emission: batches of 8 (loads of a batch in flight together, no wire read back).  The evaluator has no
    // stores in front of its loads and the compiler pipelines the plain loop; pinned batches measured 0.25 -> 0.29 ms on the RLP leaf tail
    for (int i0 = 0; i0 < n; i0 += 8) {
        SmRef ri[8], ro[8]; S vv[8], ov[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint32_t i = (uint32_t)(i0 + q < n ? i0 + q : n - 1); ri[q] = in + i; ro[q] = o + i; }
        const SmLoaded<8> hi = sm_load(p, ri);
        const SmLoaded<8> ho = sm_load(p, ro);
#pragma unroll
        for (int q = 0; q < 8; q++) vv[q] = p.get(src + (uint32_t)(i0 + q < n ? i0 + q : n - 1));
        sm_commit(p, ri, hi, vv);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = (uint32_t)(i0 + q < n ? i0 + q : n - 1);
            ov[q] = (uint32_t)count > i ? vv[q] : 0;
        }
        sm_commit(p, ro, ho, ov);
    }
    return o;
}
// Concat(A,B) :47-83  [out[A+B], outLen | a[A], aLen, b[B], bLen | maskedA[A], maskedB[B], shiftedB[A+B]]
//   || AssertLessEqThan(16) x2, Mask(A), Mask(B), ShiftRight(B, A)
template <class P> GD SmRef gConcat(P& p, int La, int Lb, SmRef a, S aLen, SmRef b, S bLen, S& outLen, bool split = false) {
    SmRef o = p.sms(La + Lb), ol = p.sms(1), ia = p.sms(La), ial = p.sms(1), ib = p.sms(Lb), ibl = p.sms(1);
    SmRef mA = p.sms(La), mB = p.sms(Lb), sB = p.sms(La + Lb);
    copy_n(p, ia, a, (int)(La));
    aLen = p.put(ial, aLen);
    copy_n(p, ib, b, (int)(Lb));
    bLen = p.put(ibl, bLen);
    gAssertLessEqThanS(p, 16, aLen, (S)La);
    gAssertLessEqThanS(p, 16, bLen, (S)Lb);
    SmRef x = gMask(p, La, ia, aLen);
    copy_n(p, mA, x, (int)(La));
    x = gMask(p, Lb, ib, bLen);
    copy_n(p, mB, x, (int)(Lb));
    x = gShiftRight(p, Lb, La, mB, aLen, split);
    // shiftedB[] <== ShiftRight.out[], out[i] <== maskedA[i] + shiftedB[i]: batches of 8 (as single puts every read of the wires written just before waited for the
    // store in front of it: 2 (La + Lb) store -> load round trips, the longest stretch of the RLP leaf / account assembly)
    if constexpr (P::is_check) {        // (the evaluator: no stores in front of its loads, the plain loop pipelines)
        for (int i = 0; i < La + Lb; i++) {
            const S sv = p.put(sB + (uint32_t)i, p.get(x + (uint32_t)i));
            p.put(o + (uint32_t)i, i < La ? p.get(mA + (uint32_t)i) + sv : sv);
        }
        outLen = p.put(ol, aLen + bLen);
        return o;
    }
    for (int i0 = 0; i0 < La + Lb; i0 += 8) {
        SmRef rs[8], ro[8]; S xv[8], av[8], ov[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint32_t i = (uint32_t)(i0 + q < La + Lb ? i0 + q : La + Lb - 1); rs[q] = sB + i; ro[q] = o + i; }
        const SmLoaded<8> hs = sm_load(p, rs);
        const SmLoaded<8> ho = sm_load(p, ro);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = (uint32_t)(i0 + q < La + Lb ? i0 + q : La + Lb - 1);
            xv[q] = p.get(x + i);
            av[q] = i < (uint32_t)La ? p.get(mA + i) : 0;
        }
        sm_commit(p, rs, hs, xv);
#pragma unroll
        for (int q = 0; q < 8; q++) ov[q] = av[q] + (P::is_check ? hs.s[q] : xv[q]);
        sm_commit(p, ro, ho, ov);
    }
    outLen = p.put(ol, aLen + bLen);
    return o;
}

// ---- Concat(A, B) for the EVALUATOR of a composite unit, cut at stored wires (round 5): the composite keeps the head -- own copies, lengths, the two asserts --, the
// children run as wavefronts of their own: Mask(A) + maskedA[] | Mask(B) + maskedB[] | ShiftRight(B, A) | shiftedB[] / out[] / outLen.  Every relation of a part is between
// stored wires; a part finds the block's own wires from the block's first (wire, SM rank) and its child from the cursor the planner recorded for it (concat_cursors).
struct ConcatOwn { SmRef o, ol, ia, ial, ib, ibl, mA, mB, sB; };
HD ConcatOwn concat_own(uint32_t w0, uint32_t s0, int La, int Lb) {
    ConcatOwn c; uint32_t k = 0;
    auto take = [&](uint32_t n) { SmRef r = {w0 + k, s0 + k}; k += n; return r; };
    c.o = take(La + Lb); c.ol = take(1); c.ia = take(La); c.ial = take(1); c.ib = take(Lb); c.ibl = take(1); c.mA = take(La); c.mB = take(Lb); c.sB = take(La + Lb);
    return c;
}
// the head, in place of gConcat, for a policy whose parts run elsewhere; the cursor is left behind the asserts (nothing after the head uses it)
template <class P> GD SmRef gConcatHead(P& p, int La, int Lb, SmRef a, S aLen, SmRef b, S bLen, S& outLen) {
    const ConcatOwn c = concat_own(p.cur.w, p.cur.s, La, Lb);
    p.sms(2 * (La + Lb) + 2 * La + 2 * Lb + 3);
    copy_n(p, c.ia, a, La);
    aLen = p.put(c.ial, aLen);
    copy_n(p, c.ib, b, Lb);
    bLen = p.put(c.ibl, bLen);
    gAssertLessEqThanS(p, 16, aLen, (S)La);
    gAssertLessEqThanS(p, 16, bLen, (S)Lb);
    outLen = p.get(c.ol);                                     // (outLen <== aLen + bLen is the last part's relation)
    return c.o;
}
// part 0: Mask(A) at p.cur | 1: Mask(B) | 2: ShiftRight(B, A) | 3: the sums (p.cur = the ShiftRight block: its out[] comes first)
template <class P> GD void gConcatPart(P& p, int part, int La, int Lb, uint32_t w0, uint32_t s0) {
    const ConcatOwn c = concat_own(w0, s0, La, Lb);
    const S aLen = p.get(c.ial);
    if (part == 0) { SmRef x = gMask(p, La, c.ia, aLen); copy_n(p, c.mA, x, La); }
    else if (part == 1) { SmRef x = gMask(p, Lb, c.ib, p.get(c.ibl)); copy_n(p, c.mB, x, Lb); }
    else if (part == 2) gShiftRight(p, Lb, La, c.mB, aLen, true);
    else {
        const SmRef x = {p.cur.w, p.cur.s};
        for (int i = 0; i < La + Lb; i++) {
            const S sv = p.put(c.sB + (uint32_t)i, p.get(x + (uint32_t)i));
            p.put(c.o + (uint32_t)i, i < La ? p.get(c.mA + (uint32_t)i) + sv : sv);
        }
        p.put(c.ol, aLen + p.get(c.ibl));
    }
}
// where the three children of the Concat block at c0 begin (host planner: a counting walk)
template <class CP> inline void concat_cursors(const CP& proto, Cur c0, int La, int Lb, Cur out[3]) {
    CP q; q.cur = c0; q.decl_order = proto.decl_order;
    q.sms(2 * (La + Lb) + 2 * La + 2 * Lb + 3);
    gAssertLessEqThanS(q, 16, 0, (S)La); gAssertLessEqThanS(q, 16, 0, (S)Lb);
    out[0] = q.cur; { SmRef z = {0, 0}; gMask(q, La, z, 0); }
    out[1] = q.cur; { SmRef z = {0, 0}; gMask(q, Lb, z, 0); }
    out[2] = q.cur;
}

// ============================================================================ circuits/utils/convert.circom
// LittleEndianBytes2Num(N<=31) :12-26  [out | in[N]] || AssertByteString(N)
template <class P> GD F gLittleEndianBytes2NumF(P& p, int N, SmRef src) {
    FrRef o = p.frs(1); SmRef in = p.sms(N);
    Fr c = fr_zero();
    // in[] <== src[] in batches of 8 (two-phase: the evaluator's loads of a batch are issued together and its compares resolved at the batch's end --
    // as 31 single puts LLVM sank every compare to the end of the unit and the SubstringCheck head spilled 311 VGPRs)
#pragma unroll
    for (int j0 = 0; j0 < 32; j0 += 8) {
        if (j0 < N) {
            SmRef rr[8]; S vv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { const int idx = j0 + q < N ? j0 + q : N - 1; rr[q] = in + (uint32_t)idx; }
            const SmLoaded<8> h = sm_load(p, rr);
#pragma unroll
            for (int q = 0; q < 8; q++) vv[q] = p.get(src + (uint32_t)(j0 + q < N ? j0 + q : N - 1));
            sm_commit(p, rr, h, vv);
#pragma unroll
            for (int q = 0; q < 8; q++) if (j0 + q < N) c.l[(j0 + q) >> 2] |= ((uint32_t)(P::is_check ? h.s[q] : vv[q]) & 0xffu) << (8 * ((j0 + q) & 3));
        }
    }
    gAssertByteString(p, N, in);          // (values >= 256 fail here, so masking above never hides an error)
    return p.put(o, fr_to_mont(c));
}
// BigEndianBytes2Num(N) :33-39  [out | in[N] | inReversed[N]] || Reverse(N), LittleEndianBytes2Num(N)
template <class P> GD F gBigEndianBytes2NumF(P& p, int N, SmRef src) {
    FrRef o = p.frs(1); SmRef in = p.sms(N), rev = p.sms(N);
    copy_n(p, in, src, (int)(N));
    SmRef r = gReverseS(p, N, in);
    copy_n(p, rev, r, (int)(N));
    return p.put(o, gLittleEndianBytes2NumF(p, N, rev));
}
// Num2BitsSafe(N) :46-56
template <class P> GD BitRef gNum2BitsSafeF(P& p, int N, const F& in, BV* vout = nullptr, F* cout = nullptr) {
    BV v; F c;
    if (N >= 254) {   // [out[N] | in | bitsStrict[254]] || Num2Bits_strict, Fit(254, N) [out[N] | in[254]]
        BitRef o = p.bits(N); FrRef i = p.frs(1); BitRef bs = p.bits(254);
        F x = p.put(i, in);
        gNum2BitsStrict(p, x, &v, &c);
        bv_put(p, bs, 254, v);
        BitRef fo = p.bits(N), fi = p.bits(254);
        bv_put(p, fi, 254, v);
        bv_put(p, fo, N, v);                 // bits 254.. of v are zero
        bv_put(p, o, N, v);
        if (vout) *vout = v;
        if (cout) *cout = c;
        return o;
    }
    BitRef o = p.bits(N); FrRef i = p.frs(1);   // [out[N] | in] || Num2Bits(N)
    F x = p.put(i, in);
    gNum2BitsF(p, N, x, &v, &c);
    bv_put(p, o, N, v);
    if (vout) *vout = v;
    if (cout) *cout = c;
    return o;
}
// Num2LittleEndianBytes(N) :69-83  [out[N] | in | bits[8N], byteArrays[N][8]] || Num2BitsSafe(8N), Reshape(N,8), Bits2Num(8) x N
// (Reshape [out[8N] | in[8N]] is the identity on row-major data; Bits2Num(8) [out | in[8]])
template <class P> GD SmRef gNum2LittleEndianBytesF(P& p, int N, const F& in, F* cout = nullptr) {
    SmRef o = p.sms(N); FrRef i = p.frs(1); BitRef bits = p.bits(8 * N), ba = p.bits(8 * N);
    F x = p.put(i, in);
    BV v; F c;
    gNum2BitsSafeF(p, 8 * N, x, &v, &c);
    bv_put(p, bits, 8 * N, v);
    BitRef ro = p.bits(8 * N), ri = p.bits(8 * N);
    bv_put(p, ri, 8 * N, v);
    bv_put(p, ro, 8 * N, v);
    bv_put(p, ba, 8 * N, v);
    const Cur kids = p.cur;                                  // N x Bits2Num(8)
    bv_put_children(p, kids.w, kids.b, 9, 1, 8, 8 * N, v);
    sm_puts_at<P, 8>(p, N, [&](int j) { return SmRef{kids.w + 9u * (uint32_t)j, kids.s + (uint32_t)j}; }, [&](int j) { return canon_byte(c, j); });      // Bits2Num(8).out x N
    sm_puts<P, 8>(p, o, N, [&](int j) { return canon_byte(c, j); });                                                                                         // out[] <== them
    p.cur = Cur{kids.w + 9u * N, kids.b + 8u * N, kids.s + (uint32_t)N, kids.f, kids.q};
    if (cout) *cout = c;
    return o;
}
// Num2BigEndianBytes(N) :90-96  [out[N] | in | littleEndian[N]] || Num2LittleEndianBytes(N), Reverse(N) [out[N] | in[N]]
// `also`: a caller's copy of out[] (written from the same per-witness bytes, not read back)
// (the caller's copy travels by value + flag: a pointer to a local SmRef chosen at run time put the SmRef in scratch)
template <class P> GD SmRef gNum2BigEndianBytesFv(P& p, int N, const F& in, SmRef also_ref, bool has_also, F* cout = nullptr) {
    const SmRef* also = has_also ? &also_ref : nullptr;
    SmRef o = p.sms(N); FrRef i = p.frs(1); SmRef le = p.sms(N);
    F x = p.put(i, in);
    F c;
    gNum2LittleEndianBytesF(p, N, x, &c);
    SmRef ro, ri;
    auto le_byte = [&](int j) { return canon_byte(c, j); };
    auto be_byte = [&](int j) { return canon_byte(c, N - 1 - j); };
    sm_puts<P, 8>(p, le, N, le_byte);
    ro = p.sms(N); ri = p.sms(N);
    sm_puts<P, 8>(p, ri, N, le_byte);
    sm_puts<P, 8>(p, ro, N, be_byte);
    sm_puts<P, 8>(p, o, N, be_byte);
    if (has_also) sm_puts<P, 8>(p, also_ref, N, be_byte);
    if (cout) *cout = c;
    (void)also;
    return o;
}
template <class P> GD SmRef gNum2BigEndianBytesF(P& p, int N, const F& in, const SmRef* also = nullptr, F* cout = nullptr) {
    return gNum2BigEndianBytesFv(p, N, in, also ? *also : SmRef{0, 0}, also != nullptr, cout);
}
// footprint of Num2BigEndianBytes(N) (every wire count is a function of N only)
HD Cur n2be_footprint(int N) { CountP q; q.cur = Cur{0, 0, 0, 0, 0}; gNum2BigEndianBytesF(q, N, fr_zero()); return q.cur; }
// Num2BigEndianBytes as a composite unit sees it: `src` = the caller's stored wire that feeds it.  Generation and the evaluator run
// the block as a CK_N2BE unit of its own and the composite only steps over it; *cout = canonical value of x (the composite takes
// the bytes it needs from there: nothing of the block is read back).  The evaluator's unit starts from the stored `src`; the
// generator's runs in the SAME stage as the composite, so it reads `gen_src`: a wire of an EARLIER stage that carries the same value.
template <class P> GD void gNum2BigEndianBytesFU(P& p, int N, FrRef src, const F& x, Cur fp, FrRef gen_src, const SmRef* also = nullptr, F* cout = nullptr) {
    if constexpr (P::is_count) p.note(NOTE_N2BE, (uint32_t)N, p.cur, src.w, src.i, also ? also->w : 0u, also ? also->i : 0u, also ? 1u : 0u, gen_src.w, gen_src.i);
    if constexpr (P::is_check || P::is_gen) {
        p.cur = cur_add(p.cur, fp, 1);
        if (cout) *cout = fr_from_mont(x);
    } else gNum2BigEndianBytesF(p, N, x, also, cout);
}
// Bytes2Nibbles(N) :103-121  [out[2N] | in[N] | inDecomposed[N][8]] || Num2Bits(8) x N
template <class P> GD SmRef gBytes2Nibbles(P& p, int N, SmRef src) {
    SmRef o = p.sms(2 * N), in = p.sms(N); BitRef dec = p.bits(8 * N);
    for (int i = 0; i < N; i++) {
        S v = p.put(in + i, p.get(src + i));
        B bv[8];
        gNum2Bits8(p, v, bv);
        S lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            bool b = p.bit(p.put(dec + (8 * i + k), bv[k]));
            if (k < 4) lo |= (S)b << k; else hi |= (S)b << (k - 4);
        }
        p.put(o + 2 * i, hi); p.put(o + 2 * i + 1, lo);
    }
    return o;
}
// Nibbles2Bytes(n) :132-141  [bytes[n] | nibbles[2n]] || AssertBits(4) x 2n
template <class P> GD SmRef gNibbles2Bytes(P& p, int n, SmRef src) {
    SmRef o = p.sms(n), nib = p.sms(2 * n);
    copy_n(p, nib, src, (int)(2 * n));
    for (int i = 0; i < n; i++) {
        S a = p.get(nib + 2 * i), b = p.get(nib + 2 * i + 1);
        gAssertBitsS(p, 4, a); gAssertBitsS(p, 4, b);
        p.put(o + i, a * 16 + b);
    }
    return o;
}

// ============================================================================ circuits/utils/substring_check.circom:24-100
// [out | mainInput[mm], mainLen, subInput[sl] | subInputNum, M[mm+1], exists[k], isLastIndex[k], allowed[k+1], sums[k+1], doesNotExist]
// || AssertByteString(sl), AssertByteString(mm), AssertLessEqThan(16) x2, LittleEndianBytes2Num(sl), {IsEqual, IsEqual} x k, IsZero
// The MPT layer-inclusion check: M[i+1] = mainInput[i]*256^i + M[i]; exists[i] = (sub*256^i == M[i+sl]-M[i]).
// exists[i] needs NO field arithmetic: M[i+sl] - M[i] = 256^i * W_i with W_i = the sl bytes from position i as a little-endian integer (< 2^248 < p
// for bytes, which AssertByteString(mainInput) enforces), so  subNum * 256^i == M[i+sl] - M[i]  <=>  subNum == W_i : a sliding window of sl bytes
// against the canonical subNum.  mainInput[] (a copy), M[] and the IsEqual operands are DERIVED wires (policy.hpp): the emitter rebuilds them.
template <class P> GD Fr sc_window(P& p, SmRef src, uint32_t i, uint32_t sl) {      // bytes [i, i + sl) little-endian, canonical limbs (sl <= 31)
    Fr w = fr_zero();
#pragma unroll
    for (uint32_t k = 0; k < 31; k++) {
        const S b = p.get(src + (i + (k < sl ? k : sl - 1)));
        if (k < sl) w.l[k >> 2] |= ((uint32_t)b & 0xffu) << (8 * (k & 3));
    }
    return w;
}
HD void sc_window_step(Fr& w, uint32_t sl, uint32_t nb) {                          // drop the lowest byte, append nb as byte sl - 1
#pragma unroll
    for (int j = 0; j < 7; j++) w.l[j] = (w.l[j] >> 8) | (w.l[j + 1] << 24);
    w.l[7] >>= 8;
    const uint32_t pos = sl - 1;
#pragma unroll
    for (int j = 0; j < 8; j++) if ((pos >> 2) == (uint32_t)j) w.l[j] |= (nb & 0xffu) << (8 * (pos & 3));
}
template <class P> GD B gSubstringCheck(P& p, int mm, int sl, SmRef mainSrc, S mainLen, SmRef subSrc) {
    const int k = mm - sl + 1;
    BitRef o = p.bits(1); const uint32_t mi_w = p.dvs(mm); SmRef ml = p.sms(1), si = p.sms(sl);
    FrRef num = p.frs(1); const uint32_t M_w = p.dvs(mm + 1); BitRef ex = p.bits(k), isl = p.bits(k), alw = p.bits(k + 1); SmRef sums = p.sms(k + 1); BitRef dne = p.bits(1);
    mainLen = p.put(ml, mainLen);
    copy_n(p, si, subSrc, (int)(sl));
    gAssertByteString(p, sl, si);
    gAssertByteString(p, mm, mainSrc);
    gAssertLessEqThanS(p, 16, mainLen, (S)mm);
    gAssertLessEqThanS(p, 16, (S)sl, mainLen);
    const F subNum = p.put(num, gLittleEndianBytes2NumF(p, sl, si));
    const Fr subC = fr_from_mont(subNum);
    const F c256 = fr_from_i64(256);
    if constexpr (P::is_emit) {   // mainInput[] copies and the M[] prefix sums (:45-49)
        F pw = fr_one_mont(), acc = fr_zero();
        p.derived_fr(M_w, acc);
        for (int i = 0; i < mm; i++) {
            const S b = p.get(mainSrc + i);
            p.derived(mi_w + i, b);
            acc = fr_add(fr_mul(fr_from_i64(b), pw), acc);
            p.derived_fr(M_w + i + 1, acc);
            pw = fr_mul(pw, c256);
        }
    }
    // per i the loop below instantiates IsEqual([i, lastIndex]) and IsEqual(exists): two BIT outputs and four derived operand wires each
    B allowed = p.put(alw, ~(B)0);
    S sum = p.put(sums, 0);
    F pw = fr_one_mont();
    Fr win = sc_window(p, mainSrc, 0, (uint32_t)sl);
    for (int i = 0; i < k; i++) {
        const S nxt = i + sl < mm ? p.get(mainSrc + (i + sl)) : 0;
        B last = p.put(isl + i, gIsEqualS(p, (S)i, (S)(mainLen - sl + 1)));                 // :87
        allowed = p.put(alw + i + 1, allowed & ~last);
        uint32_t w0;
        B e = p.put(ex + i, gIsEqualFz(p, p.ballot(fr_eq(win, subC)), &w0));               // :91
        if constexpr (P::is_emit) { iseqf_derived(p, w0, fr_mul(subNum, pw), fr_mul(fr_to_mont(win), pw)); pw = fr_mul(pw, c256); }
        sum = p.put(sums + i + 1, sum + (S)p.bit(allowed & e));
        sc_window_step(win, (uint32_t)sl, (uint32_t)nxt);
    }
    B none = p.put(dne, gIsZeroS(p, sum));
    return p.put(o, ~none);
}

// ============================================================================ circuits/utils/rlp/integer.circom
// CountBytes(N) :16-49  [len | bytes[N] | isZero[N], stillZero[N]] || IsZero x N
template <class P> GD S gCountBytes(P& p, int N, SmRef src) {
    SmRef o = p.sms(1), by = p.sms(N); BitRef iz = p.bits(N), sz = p.bits(N);
    // one pass, the bytes in batches of 8, isZero[] / stillZero[] from the values (as two loops of single wires every byte was a load behind a store and every
    // isZero[i] was read back right after it had been written: 2 N memory round trips)
    B still = ~(B)0; S lead = 0;
    for (int i0 = 0; i0 < N; i0 += 8) {
        SmRef rr[8]; S vv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) rr[q] = by + (uint32_t)(i0 + q < N ? i0 + q : N - 1);
        const SmLoaded<8> h = sm_load(p, rr);
#pragma unroll
        for (int q = 0; q < 8; q++) vv[q] = p.get(src + (uint32_t)(i0 + q < N ? i0 + q : N - 1));
        sm_commit(p, rr, h, vv);
#pragma unroll
        for (int q = 0; q < 8; q++) if (i0 + q < N) {
            const uint32_t i = (uint32_t)(i0 + q);
            const B z = p.put(iz + i, gIsZeroS(p, P::is_check ? h.s[q] : vv[q]));
            still = p.put(sz + i, still & z);
            lead += (S)p.bit(still);
        }
    }
    return p.put(o, N - lead);
}
// RlpInteger(N) :67-110  [out[N+1], outLen | in | bytes[N], length, bigEndian[N], isSingleByte, isZero, firstRlpByte]
// || Num2BigEndianBytes(N), CountBytes(N), ShiftLeft(N), LessThan(8N), IsZero, Mux1
template <class P> GD SmRef gRlpInteger(P& p, int N, const F& in, S& outLen) {
    SmRef o = p.sms(N + 1), ol = p.sms(1); FrRef i = p.frs(1); SmRef by = p.sms(N), len = p.sms(1), be = p.sms(N);
    BitRef isb = p.bits(1), isz = p.bits(1); SmRef frb = p.sms(1);
    F x = p.put(i, in);
    SmRef r = gNum2BigEndianBytesF(p, N, x);
    copy_n(p, by, r, (int)(N));
    S length = p.put(len, gCountBytes(p, N, by));
    r = gShiftLeft(p, N, by, N - length);
    copy_n(p, be, r, (int)(N));
    B single = p.put(isb, gLessThanF(p, 8 * N, x, fr_from_i64(128)));
    B zero = p.put(isz, gIsZeroFd(p, x));
    S first = p.put(frb, gMux1SF(p, 0x80 + length, x, single));
    bool sb = p.bit(single), zb = p.bit(zero);
    p.put(o, first + (zb ? 0x80 : 0));
    for (int j = 1; j < N + 1; j++) p.put(o + j, sb ? 0 : p.get(be + (j - 1)));
    outLen = p.put(ol, (sb ? 0 : 1) + length + (zb ? 1 : 0));
    return o;
}
// RlpEmptyAccount(mb) rlp/empty_account.circom:20-134
// [out[70+mb], outLen | balance | prefixedNonceAndBalanceRlp[4+mb], prefixedNonceAndBalanceRlpLen, balanceRlp[mb+1], balanceRlpLen,
//  nonceAndBalanceRlpLen, storageAndCodeHashRlp[66]] || RlpInteger(mb), Concat(4+mb, 66)
HD uint8_t empty_account_tail(int i) {   // 0xa0|keccak(rlp(""))|0xa0|keccak("")   (empty_account.circom:9-10)
    const uint8_t t[66] = {0xa0, 0x56, 0xe8, 0x1f, 0x17, 0x1b, 0xcc, 0x55, 0xa6, 0xff, 0x83, 0x45, 0xe6, 0x92, 0xc0, 0xf8, 0x6e, 0x5b, 0x48, 0xe0, 0x1b, 0x99,
                           0x6c, 0xad, 0xc0, 0x01, 0x62, 0x2f, 0xb5, 0xe3, 0x63, 0xb4, 0x21, 0xa0, 0xc5, 0xd2, 0x46, 0x01, 0x86, 0xf7, 0x23, 0x3c, 0x92, 0x7e,
                           0x7d, 0xb2, 0xdc, 0xc7, 0x03, 0xc0, 0xe5, 0x00, 0xb6, 0x53, 0xca, 0x82, 0x27, 0x3b, 0x7b, 0xfa, 0xd8, 0x04, 0x5d, 0x85, 0xa4, 0x70};
    return t[i];
}
template <class P> GD SmRef gRlpEmptyAccount(P& p, int mb, const F& balance, S& outLen) {
    SmRef o = p.sms(70 + mb), ol = p.sms(1); FrRef ib = p.frs(1);
    SmRef pn = p.sms(4 + mb), pnl = p.sms(1), br = p.sms(mb + 1), brl = p.sms(1), nbl = p.sms(1), sc = p.sms(66);
    F bal = p.put(ib, balance);
    p.put(pn + 2, 0x80);
    S blen;
    SmRef r = gRlpInteger(p, mb, bal, blen);
    for (int j = 0; j < mb + 1; j++) p.put(pn + 3 + j, p.put(br + j, p.get(r + j)));
    blen = p.put(brl, blen);
    S nb = p.put(nbl, 1 + blen);
    S pl = p.put(pnl, 2 + nb);
    for (int j = 0; j < 66; j++) p.put(sc + j, (S)empty_account_tail(j));
    p.put(pn, 0xf8);
    p.put(pn + 1, nb + 66);
    S clen;
    SmRef c = gConcat(p, 4 + mb, 66, pn, pl, sc, (S)66, clen);
    copy_n(p, o, c, (int)(70 + mb));
    outLen = p.put(ol, clen);
    return o;
}

// ============================================================================ circuits/utils/rlp/merkle_patricia_trie_leaf.circom
// TruncatedAddressHash(b) :50-90
// [out[b+1], outLen | addressHashNibbles[2b], addressHashNibblesLen | div, rem, shifted[2b], outNibbles[2b+2], temp[2b-1]]
// || AssertLessEqThan(7), Divide(7), ShiftLeft(2b), Mux1 x (2b-1), Nibbles2Bytes(b+1);  temp[] is never assigned (:76) -> 0
template <class P> GD SmRef gTruncatedAddressHash(P& p, int b, SmRef src, S len, S& outLen) {
    const int n2 = 2 * b;
    SmRef o = p.sms(b + 1), ol = p.sms(1), in = p.sms(n2), il = p.sms(1), dv = p.sms(1), rm = p.sms(1), shf = p.sms(n2), on = p.sms(n2 + 2), tmp = p.sms(n2 - 1);
    copy_n(p, in, src, (int)(n2));
    len = p.put(il, len);
    for (int i = 0; i < n2 - 1; i++) p.put(tmp + i, 0);
    gAssertLessEqThanS(p, 7, len, (S)n2);
    S q, r;
    gDivide(p, 7, len, (S)2, q, r);
    q = p.put(dv, q); r = p.put(rm, r);
    SmRef s = gShiftLeft(p, n2, in, n2 - len);
    copy_n(p, shf, s, (int)(n2));
    p.put(on, 2 + r);
    p.put(on + 1, r * p.get(shf));
    B rbit = p.ballot(r & 1);
    for (int i = 0; i < n2; i++) {
        if (i < n2 - 1) p.put(on + i + 2, gMux1S(p, p.get(shf + i), p.get(shf + i + 1), rbit));
        else p.put(on + i + 2, (1 - r) * p.get(shf + i));
    }
    SmRef by = gNibbles2Bytes(p, b + 1, on);
    copy_n(p, o, by, (int)(b + 1));
    outLen = p.put(ol, 1 + q);
    return o;
}
// RlpMerklePatriciaTrieLeaf(ab, bb) :102-189
template <class P> GD SmRef gRlpMptLeaf(P& p, int ab, int bb, SmRef nibSrc, S nibLen, const F& balance, S& outLen) {
    const int maxAcc = 4 + bb + 66, maxVal = 2 + maxAcc, maxKey = 1 + ab, maxPK = 2 + 1 + maxKey, maxOut = maxPK + maxVal;
    SmRef o = p.sms(maxOut), ol = p.sms(1), in = p.sms(2 * ab), inl = p.sms(1); FrRef ib = p.frs(1);
    SmRef key = p.sms(maxKey), keyLen = p.sms(1), acc = p.sms(maxAcc), accLen = p.sms(1), pk = p.sms(maxPK), pkLen = p.sms(1), val = p.sms(maxVal), valLen = p.sms(1);
    copy_n(p, in, nibSrc, (int)(2 * ab));
    nibLen = p.put(inl, nibLen);
    F bal = p.put(ib, balance);
    S kl;
    SmRef r = gTruncatedAddressHash(p, ab, in, nibLen, kl);
    copy_n(p, key, r, (int)(maxKey));
    kl = p.put(keyLen, kl);
    gAssertGreaterEqThanS(p, 16, kl, (S)2);                                   // :151
    S al;
    r = gRlpEmptyAccount(p, bb, bal, al);
    copy_n(p, acc, r, (int)(maxAcc));
    al = p.put(accLen, al);
    p.put(val, 0xb8); p.put(val + 1, al);
    for (int i = 0; i < maxAcc; i++) p.put(val + 2 + i, p.get(acc + i));
    S vl = p.put(valLen, 2 + al);
    p.put(pk, 0xf8); p.put(pk + 1, (kl + 1) + vl); p.put(pk + 2, 0x80 + kl);
    for (int i = 0; i < maxKey; i++) p.put(pk + 3 + i, p.get(key + i));
    S pl = p.put(pkLen, 3 + kl);
    S cl;
    SmRef c = gConcat(p, maxPK, maxVal, pk, pl, val, vl, cl);
    copy_n(p, o, c, (int)(maxOut));
    outLen = p.put(ol, cl);
    return o;
}
// IsInRange(B) :196-207  [out | lower, value, upper | lowerLteValue, valueLteUpper] || AssertBits x3, LessEqThan x2
template <class P> GD B gIsInRange(P& p, int nb, S lo, S v, S hi) {
    BitRef o = p.bits(1); SmRef in = p.sms(3); BitRef mid = p.bits(2);
    lo = p.put(in, lo); v = p.put(in + 1, v); hi = p.put(in + 2, hi);
    gAssertBitsS(p, nb, lo); gAssertBitsS(p, nb, v); gAssertBitsS(p, nb, hi);
    B a = p.put(mid, gLessEqThanS(p, nb, lo, v));
    B b = p.put(mid + 1, gLessEqThanS(p, nb, v, hi));
    return p.put(o, a & b);
}
// LeafDetector(N) :247-294
template <class P> GD B gLeafDetector(P& p, int N, SmRef src, S layerLen) {
    BitRef o = p.bits(1); SmRef layer = p.sms(N), ll = p.sms(1);
    // intermediates in declaration order (:255-287)
    BitRef leafPrefixIsF8 = p.bits(1); SmRef totalLength = p.sms(1); BitRef isConsistentWithLayerLen = p.bits(1); SmRef keyPrefix = p.sms(1);
    BitRef keyPrefixIsValid = p.bits(1), keyIsMultiByte = p.bits(1); SmRef keyExtraLen = p.sms(1), keyLen = p.sms(1), valueWrapperPrefix = p.sms(1);
    BitRef valueWrapperPrefixIsB8 = p.bits(1); SmRef valueWrapperLen = p.sms(1), valuePrefix = p.sms(1); BitRef valuePrefixIsF8 = p.bits(1);
    SmRef valueLen = p.sms(1); BitRef isValueWrapperLenConsistent = p.bits(1), isKeyValueLenEqualWithLayerLen = p.bits(1);
    copy_n(p, layer, src, (int)(N));
    layerLen = p.put(ll, layerLen);
    gAssertLessEqThanS(p, 16, layerLen, (S)N);
    B m[7];
    m[0] = p.put(leafPrefixIsF8, gIsEqualS(p, p.get(layer), (S)0xf8));
    S tl = p.put(totalLength, p.get(layer + 1));
    m[1] = p.put(isConsistentWithLayerLen, gIsEqualS(p, tl + 2, layerLen));
    S kp = p.put(keyPrefix, p.get(layer + 2));
    m[2] = p.put(keyPrefixIsValid, gLessEqThanS(p, 16, kp, (S)0xb7));
    B multi = p.put(keyIsMultiByte, gIsInRange(p, 16, (S)0x81, kp, (S)0xb7));
    S kel = p.put(keyExtraLen, p.bit(multi) ? kp - 0x80 : 0);
    S kl = p.put(keyLen, 1 + kel);
    S vwp = p.put(valueWrapperPrefix, gSelectorS(p, N, layer, 2 + kl + 0));
    m[3] = p.put(valueWrapperPrefixIsB8, gIsEqualS(p, vwp, (S)0xb8));
    S vwl = p.put(valueWrapperLen, gSelectorS(p, N, layer, 2 + kl + 1));
    S vp = p.put(valuePrefix, gSelectorS(p, N, layer, 2 + kl + 2));
    m[5] = p.put(valuePrefixIsF8, gIsEqualS(p, vp, (S)0xf8));
    S vl = p.put(valueLen, gSelectorS(p, N, layer, 2 + kl + 2 + 1));
    m[4] = p.put(isValueWrapperLenConsistent, gIsEqualS(p, vwl, vl + 2));
    m[6] = p.put(isKeyValueLenEqualWithLayerLen, gIsEqualS(p, kl + vl + 6, layerLen));
    return p.put(o, MultiANDg<P, 7>::run(p, m));
}

// ============================================================================ circuits/utils/keccak.circom:412-446
// Pad(maxBlocks, blockSize) as a template of its own (KeccakBytes' instance is cut into the kb_head / kb_range units of circuits.hpp)
// [out[m], numBlocks | in[m], inLen | div, rem, filter[m+1], isEq[m], isLast[m]] || Divide(16), AssertLessEqThan(16), IsEqual([i, inLen]) x m,
// IsEqual([i, numBlocks*blockSize - 1]) x m;  out[i] = in[i]*filter[i+1] + isEq[i] + 0x80*isLast[i]
template <class P> GD SmRef gPad(P& p, int mb, int bs, SmRef src, S inLen, S& numBlocks) {
    const int m = mb * bs;
    SmRef o = p.sms(m), nbr = p.sms(1), in = p.sms(m), il = p.sms(1), dv = p.sms(1), rm = p.sms(1); BitRef flt = p.bits(m + 1), isEq = p.bits(m), isLast = p.bits(m);
    copy_n(p, in, src, m);
    inLen = p.put(il, inLen);
    S q, r;
    gDivide(p, 16, inLen, (S)bs, q, r);
    q = p.put(dv, q); p.put(rm, r);
    const S nb = p.put(nbr, q + 1);
    gAssertLessEqThanS(p, 16, nb, (S)mb);
    B f = p.put(flt, ~(B)0);
    for (int i = 0; i < m; i++) {
        const B e = p.put(isEq + i, gIsEqualS(p, (S)i, inLen));
        f = p.put(flt + i + 1, f & ~e);
    }
    const S last = (S)((uint32_t)nb * (uint32_t)bs - 1u);
    for (int i = 0; i < m; i++) {
        const B l = p.put(isLast + i, gIsEqualS(p, (S)i, last));
        const S v = p.get(in + i);
        p.put(o + i, (p.bit(p.get(flt + i + 1)) ? v : 0) + (S)p.bit(p.get(isEq + i)) + (p.bit(l) ? 0x80 : 0));
    }
    numBlocks = nb;
    return o;
}

// ============================================================================ circuits/utils/burn_address.circom:47-58
// BurnAddress  [addressBytes[20] | burnKey, revealAmount, burnExtraCommitment | hash, hashBytes[32]] || Poseidon(4), Num2BigEndianBytes(32), Fit(32,20)
template <class P> GD SmRef gBurnAddress(P& p, const PosOff& k5, const F& prefix0, const F& bk, const F& ra, const F& bec, Cur fp_n2be32, const PosSrc& psrc, F* hash_canon = nullptr) {
    SmRef o = p.sms(20); FrRef in = p.frs(3), h = p.frs(1); SmRef hb = p.sms(32);
    F pin[4]; pin[0] = prefix0; pin[1] = p.put(in, bk); pin[2] = p.put(in + 1, ra); pin[3] = p.put(in + 2, bec);
    const FrRef pos_out = {p.cur.w, p.cur.f};            // Poseidon.out: the block's first wire
    F hash = p.put(h, gPoseidonU<P, 5>(p, k5, pin, psrc));
    F c;
    gNum2BigEndianBytesFU(p, 32, h, hash, fp_n2be32, pos_out, &hb, &c);   // hashBytes[i] = big-endian byte i = little-endian byte 31 - i, written per witness
    SmRef fo = p.sms(20), fi = p.sms(32);                // Fit(32, 20)  [out[20] | in[32]]
    auto be_byte = [&](int i) { return canon_byte(c, 31 - i); };
    sm_puts<P, 8>(p, fi, 32, be_byte); sm_puts<P, 8>(p, fo, 20, be_byte); sm_puts<P, 8>(p, o, 20, be_byte);
    if (hash_canon) *hash_canon = c;
    return o;
}
