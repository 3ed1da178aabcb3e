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

// Montgomery product a*b*2^-256 mod p, CIOS over 32-bit limbs.
// by value: a non-inlined call passes the operands in 16 VGPRs and returns in 8 (by-reference would go through scratch)
HDN Fr fr_mul(Fr a, Fr b) {
    const uint32_t P[8] = FR_P_LIMBS;
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { c += (uint64_t)a.l[j] * b.l[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
        c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
        uint32_t m = t[0] * FR_NINV32;
        c = (uint64_t)m * P[0] + t[0]; c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) { c += (uint64_t)m * P[j] + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
        c += t[8]; t[7] = (uint32_t)c; t[8] = t[9] + (uint32_t)(c >> 32);
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return (t[8] || fr_geq_p(r)) ? fr_sub_p(r) : r;
}
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
// x^(p-2) by square-and-multiply over the fixed exponent; 0 -> 0.
HDN Fr fr_inv(Fr a) {
    const uint32_t P[8] = FR_P_LIMBS;
    Fr acc = fr_one_mont();
    for (int i = 253; i >= 0; i--) {
        acc = fr_sqr(acc);
        uint32_t w = P[i >> 5];
        if (i < 32) w -= 2;                 // exponent p-2 (low limb 0xf0000001 - 2, no borrow)
        if ((w >> (i & 31)) & 1) acc = fr_mul(acc, a);
    }
    return fr_is_zero(a) ? a : acc;
}
HD uint32_t fr_bit(const Fr& canon, int i) { return (canon.l[i >> 5] >> (i & 31)) & 1; }
