// BN254 Fr for the HIP witness generator: 8 x 32-bit limbs, Montgomery form (x * 2^256 mod p),
// one field element per lane held entirely in VGPRs (lane = witness), so carries stay in registers
// and no cross-lane traffic is needed.  32-bit limbs because CDNA4's integer multiplier is 32-bit
// (v_mad_u64_u32 does the 32x32+64 multiply-accumulate of one CIOS step in one instruction).
// __host__ __device__ so the same code runs in the host-side layout planner.
#pragma once
#include <stdint.h>

#ifndef HD
#ifdef __HIPCC__
#define HD __host__ __device__ __forceinline__
#define HDN __host__ __device__ inline __attribute__((noinline))
#else
#define HD inline
#define HDN inline
#endif
#endif

struct Fr { uint32_t l[8]; };
// opaque use of a VGPR value: a point the compiler cannot move the computation of x across (tests/hostsim builds for x86)
#ifdef POB_HOSTSIM
#define POB_OPAQUE(x) asm volatile("" : "+r"(x))
#define POB_OPAQUE_S(x) asm volatile("" : "+r"(x))
#else
#define POB_OPAQUE(x) asm volatile("" : "+v"(x))
#define POB_OPAQUE_S(x) asm volatile("" : "+s"(x))      // a wavefront-uniform value (scalar registers)
#endif

// p, R = 2^256 mod p (Montgomery 1), R2 = 2^512 mod p, -p^-1 mod 2^32
#define FR_P_LIMBS   {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}
#define FR_R_LIMBS   {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}
#define FR_R2_LIMBS  {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}
#define FR_NINV32 0xefffffffu

HD Fr fr_p() { Fr r = {FR_P_LIMBS}; return r; }
HD Fr fr_one_mont() { Fr r = {FR_R_LIMBS}; return r; }
HD Fr fr_r2() { Fr r = {FR_R2_LIMBS}; return r; }
HD Fr fr_zero() { Fr r = {{0, 0, 0, 0, 0, 0, 0, 0}}; return r; }

HD bool fr_is_zero(const Fr& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i];
    return o == 0;
}
HD bool fr_eq(const Fr& a, const Fr& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}
HD bool fr_geq_p(const Fr& a) {
    const uint32_t P[8] = FR_P_LIMBS;
    bool gt = false, lt = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (!gt && !lt) { gt = a.l[i] > P[i]; lt = a.l[i] < P[i]; }
    }
    return !lt;
}
HD Fr fr_sub_p(const Fr& a) {
    const uint32_t P[8] = FR_P_LIMBS;
    Fr r; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.l[i] - P[i] - br; r.l[i] = (uint32_t)d; br = (d >> 63) & 1; }
    return r;
}
HD Fr fr_add(const Fr& a, const Fr& b) {
    Fr r; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)a.l[i] + b.l[i]; r.l[i] = (uint32_t)c; c >>= 32; }
    // a,b < p < 2^254 so no carry out of 256 bits
    return fr_geq_p(r) ? fr_sub_p(r) : r;
}
HD Fr fr_sub(const Fr& a, const Fr& b) {
    const uint32_t P[8] = FR_P_LIMBS;
    Fr r; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.l[i] - b.l[i] - br; r.l[i] = (uint32_t)d; br = (d >> 63) & 1; }
    if (br) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)r.l[i] + P[i]; r.l[i] = (uint32_t)c; c >>= 32; }
    }
    return r;
}
HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }

// Montgomery product a*b*2^-256 mod p.
// by value: a non-inlined call passes the operands in 16 VGPRs and returns in 8 (by-reference would go through scratch)
#if defined(__HIP_DEVICE_COMPILE__)
// Device: product scanning (FIPS Montgomery).  Column k accumulates sum_{i+j=k} a_i*b_j + m_i*p_j in a 96-bit accumulator:
// v_mad_u64_u32 adds a 32x32 product into the low 64 bits, its carry-out goes into the third word with v_addc_co_u32 -- two
// instructions per product, no 64-bit adds and no operand zero-extension (the row-wise CIOS form compiles to ~600 VALU
// instructions, half of them moves; this form to ~350).
#define FR_MADC(x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc")
#define FR_MADC_S(x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(x), "s"(y) : "vcc")
// (splitting the a*b and m*p products over two accumulators to give a lone wavefront two dependency chains measured no faster)
HDN Fr fr_mul(Fr a, Fr b) {
    const uint32_t P[8] = FR_P_LIMBS;
    uint32_t m[8], t[9];
    uint64_t acc = 0; uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i < k; i++) { FR_MADC(a.l[i], b.l[k - i]); FR_MADC_S(m[i], P[k - i]); }
        FR_MADC(a.l[k], b.l[0]);
        m[k] = (uint32_t)acc * FR_NINV32;
        FR_MADC_S(m[k], P[0]);
        acc = (acc >> 32) | ((uint64_t)hi << 32); hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
        for (int i = k - 7; i < 8; i++) { FR_MADC(a.l[i], b.l[k - i]); FR_MADC_S(m[i], P[k - i]); }
        t[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)hi << 32); hi = 0;
    }
    t[8] = (uint32_t)acc;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return (t[8] || fr_geq_p(r)) ? fr_sub_p(r) : r;
}
#undef FR_MADC
#undef FR_MADC_S
#else
// Host (layout planner, tables): CIOS over 32-bit limbs.
HDN Fr fr_mul(Fr a, Fr b) {
    const uint32_t P[8] = FR_P_LIMBS;
    uint32_t t[10];
    for (int i = 0; i < 10; i++) t[i] = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 8; j++) { c += (uint64_t)a.l[j] * b.l[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
        c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
        uint32_t m = t[0] * FR_NINV32;
        c = (uint64_t)m * P[0] + t[0]; c >>= 32;
        for (int j = 1; j < 8; j++) { c += (uint64_t)m * P[j] + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
        c += t[8]; t[7] = (uint32_t)c; t[8] = t[9] + (uint32_t)(c >> 32);
    }
    Fr r;
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return (t[8] || fr_geq_p(r)) ? fr_sub_p(r) : r;
}
#endif
HD Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }
HD Fr fr_to_mont(const Fr& canon) { return fr_mul(canon, fr_r2()); }
HD Fr fr_from_mont(const Fr& m) { Fr one = {{1, 0, 0, 0, 0, 0, 0, 0}}; return fr_mul(m, one); }
// small signed integer -> Montgomery form (negative k -> p - |k|)
HD Fr fr_from_i64(int64_t k) {
    uint64_t a = k < 0 ? (uint64_t)(-k) : (uint64_t)k;
    Fr c = {{(uint32_t)a, (uint32_t)(a >> 32), 0, 0, 0, 0, 0, 0}};
    Fr m = fr_to_mont(c);
    return k < 0 ? fr_neg(m) : m;
}
// Field inversion, Montgomery in / Montgomery out, 0 -> 0.  Binary extended Euclid on 256-bit integers (shifts, adds,
// compares only -- no multiplier): with x = a*R the loop finds y = x^-1 mod p, and Montgomery-multiplying y by R^3 gives
// a^-1 * R.  Invariants  A*x = u, C*x = v (mod p); every step halves u or v, so <= 2*254 steps; lanes that are done idle
// under predication and the wavefront leaves the loop when all 64 are done.  ~8x cheaper than the a^(p-2) ladder (380
// Montgomery products), which matters for SubstringCheck's per-range batch inversions.
#define FR_R3_LIMBS {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu}
HD void fr256_shr1(uint32_t* a, uint32_t top) {
#pragma unroll
    for (int i = 0; i < 7; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
    a[7] = (a[7] >> 1) | (top << 31);
}
HD void fr256_halfmod(uint32_t* a) {            // a <- a/2 mod p
    const uint32_t P[8] = FR_P_LIMBS;
    uint32_t carry = 0;
    if (a[0] & 1) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)a[i] + P[i]; a[i] = (uint32_t)c; c >>= 32; }
        carry = (uint32_t)c;
    }
    fr256_shr1(a, carry);
}
HD void fr256_submod(uint32_t* a, const uint32_t* b) {   // a <- a - b mod p
    const uint32_t P[8] = FR_P_LIMBS;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a[i] - b[i] - br; a[i] = (uint32_t)d; br = (d >> 63) & 1; }
    if (br) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { c += (uint64_t)a[i] + P[i]; a[i] = (uint32_t)c; c >>= 32; }
    }
}
#if defined(__HIP_DEVICE_COMPILE__)
// Device: Kaliski's almost-inverse with batched shifts, branch-free per step.  Phase 1 works on plain 256-bit integers --
// (u, v) start as (p, x), the coefficients (r, s) are only added and shifted LEFT (no modular halving), k counts the shifts --
// and ends with r = -x^-1 * 2^k mod p, 254 <= k <= 508.  Every step subtracts the smaller of the two odd values from the larger
// and strips ALL trailing zeros of the difference at once (ctz), so ~190 steps of ~120 instructions replace the ~360 steps of
// the classic loop whose four divergent paths a wavefront executes one after the other.  Phase 2 is one Montgomery product with
// 2^(768-k) from a table: (x = a*R)  x^-1 * 2^k * 2^(768-k) * 2^-256 = a^-1 * R.
#include "fr_pow2_table.h"
static __device__ const uint32_t FR_POW2_TAB[FR_POW2_TAB_LEN * 8] = FR_POW2_TAB_INIT;
// fr_inv_inl: the inlined form -- as a called function it needs more VGPRs than the caller-saved set and saves callee-saved ones on the
// stack; fr_inv: the called form.  (Round 4: only the emitter inverts -- every IsZero.inv wire is derived -- and the inversion test kernel.)
HD Fr fr_inv_inl(Fr a) {
    const uint32_t P[8] = FR_P_LIMBS;
    uint32_t u[8], v[8], r[8], s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { u[i] = P[i]; v[i] = a.l[i]; r[i] = 0; s[i] = i == 0; }
    uint32_t k = 0;
    bool active = !fr_is_zero(a);
    for (int it = 0; it < 1100; it++) {
        if (!__any(active)) break;
        const bool eu = active && !(u[0] & 1), ev = active && !(v[0] & 1);
        if (__any(eu || ev)) {              // only before the first step (x even) and after a difference with > 31 trailing zeros
            const bool wu = eu;             // shift u (and s) or v (and r)
            const uint32_t w0 = wu ? u[0] : v[0];
            const uint32_t t = (eu || ev) ? (w0 ? (uint32_t)__builtin_ctz(w0) : 31u) : 0u;     // 0: this lane has nothing to fix
            const uint32_t tt = t > 31u ? 31u : t;
            uint32_t x[8], c[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { x[i] = wu ? u[i] : v[i]; c[i] = wu ? s[i] : r[i]; }
            uint32_t xs[8], cs[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t hi_ = i < 7 ? x[i + 1] : 0u, lo_ = i > 0 ? c[i - 1] : 0u;
                xs[i] = tt ? (x[i] >> tt) | (hi_ << (32 - tt)) : x[i];
                cs[i] = tt ? (c[i] << tt) | (lo_ >> (32 - tt)) : c[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (wu) { u[i] = xs[i]; s[i] = cs[i]; } else { v[i] = xs[i]; r[i] = cs[i]; }
            }
            k += tt;
            continue;
        }
        // both odd: d = |u - v|, sum = r + s
        uint32_t d[8], sum[8];
        uint32_t bw = 0, cy = 0, nzd = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint64_t df = (uint64_t)u[i] - v[i] - bw; d[i] = (uint32_t)df; bw = (uint32_t)(df >> 63);
            const uint64_t sm = (uint64_t)r[i] + s[i] + cy; sum[i] = (uint32_t)sm; cy = (uint32_t)(sm >> 32);
        }
        if (bw) {                            // u < v: d = -d
            uint32_t c1 = 1;
#pragma unroll
            for (int i = 0; i < 8; i++) { const uint64_t ng = (uint64_t)(~d[i]) + c1; d[i] = (uint32_t)ng; c1 = (uint32_t)(ng >> 32); }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) nzd |= d[i];
        const bool gt = !bw && nzd != 0;     // u > v: u <- d >> t, r <- r + s, s <<= t;  otherwise v <- d >> t, s <- r + s, r <<= t
        const uint32_t t = nzd ? (d[0] ? (uint32_t)__builtin_ctz(d[0]) : 31u) : 1u;
        const uint32_t tt = t > 31u ? 31u : t;                   // >= 1: the difference of two odd values is even
        uint32_t dsh[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t hi_ = i < 7 ? d[i + 1] : 0u;
            dsh[i] = (d[i] >> tt) | (hi_ << (32 - tt));
            const uint32_t ci = gt ? s[i] : r[i], cl = i > 0 ? (gt ? s[i - 1] : r[i - 1]) : 0u;
            sh[i] = (ci << tt) | (cl >> (32 - tt));
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (gt) { u[i] = dsh[i]; r[i] = sum[i]; s[i] = sh[i]; } else { v[i] = dsh[i]; s[i] = sum[i]; r[i] = sh[i]; }
            }
            k += tt;
            if (nzd == 0) active = false;    // u == v (== gcd = 1): v is now 0, done
        }
    }
    // r < 2p; almost-inverse = p - (r mod p)
    Fr rr;
#pragma unroll
    for (int i = 0; i < 8; i++) rr.l[i] = r[i];
    if (fr_geq_p(rr)) rr = fr_sub_p(rr);
    Fr res = fr_sub(fr_zero(), rr);
    Fr c;
    const uint32_t kk = k < FR_POW2_TAB_LEN ? k : 0;
#pragma unroll
    for (int i = 0; i < 8; i++) c.l[i] = FR_POW2_TAB[kk * 8 + i];
    Fr y = fr_mul(res, c);
    return fr_is_zero(a) ? fr_zero() : y;
}
HDN Fr fr_inv(Fr a) { return fr_inv_inl(a); }
// a^(p-2) by square-and-multiply over fr_mul calls only (no stack): the emitter's rare path (IsZero operands beyond the table)
HD Fr fr_inv_fermat(const Fr& a) {
    const uint32_t E[8] = {0xefffffffu, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};   // p - 2
    Fr r = fr_one_mont();
    for (int i = 253; i >= 0; i--) {
        r = fr_mul(r, r);
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) if ((i >> 5) == j) w = E[j];
        if ((w >> (i & 31)) & 1) r = fr_mul(r, a);
    }
    return r;
}
#else
HDN Fr fr_inv(Fr a) {
    const uint32_t P[8] = FR_P_LIMBS;
    uint32_t u[8], v[8], A[8], C[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { u[i] = a.l[i]; v[i] = P[i]; A[i] = i == 0; C[i] = 0; }
    for (int it = 0; it < 1024; it++) {
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) nz |= u[i];
#ifdef __HIP_DEVICE_COMPILE__
        if (!__any(nz != 0)) break;
#else
        if (!nz) break;
#endif
        if (nz) {
            if (!(u[0] & 1)) { fr256_shr1(u, 0); fr256_halfmod(A); }
            else if (!(v[0] & 1)) { fr256_shr1(v, 0); fr256_halfmod(C); }
            else {
                bool ge = true, dec = false;          // u >= v ?
#pragma unroll
                for (int i = 7; i >= 0; i--) if (!dec && u[i] != v[i]) { ge = u[i] > v[i]; dec = true; }
                if (ge) {
                    uint64_t br = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)u[i] - v[i] - br; u[i] = (uint32_t)d; br = (d >> 63) & 1; }
                    fr256_shr1(u, 0); fr256_submod(A, C); fr256_halfmod(A);
                } else {
                    uint64_t br = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)v[i] - u[i] - br; v[i] = (uint32_t)d; br = (d >> 63) & 1; }
                    fr256_shr1(v, 0); fr256_submod(C, A); fr256_halfmod(C);
                }
            }
        }
    }
    Fr y, r3 = {FR_R3_LIMBS};
#pragma unroll
    for (int i = 0; i < 8; i++) y.l[i] = C[i];
    return fr_mul(y, r3);          // x = 0: the loop never runs, C = 0 -> 0
}
HD Fr fr_inv_inl(const Fr& a) { return fr_inv(a); }
HD Fr fr_inv_fermat(const Fr& a) { return fr_inv(a); }
#endif
HD uint32_t fr_bit(const Fr& canon, int i) { return (canon.l[i >> 5] >> (i & 31)) & 1; }
