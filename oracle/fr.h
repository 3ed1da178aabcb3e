/* TEST INFRASTRUCTURE -- CPU oracle, not product code.
 *
 * BN254 scalar field Fr arithmetic for the witness oracle.  Restates what the circom-emitted
 * runtime's fr.cpp/fr.asm provide (UNVENDORED): canonical representatives in
 * [0,p), a fast path for "short" values (the emitted FrElement has a 32-bit short form; here any
 * value < 2^64 takes the integer path) and Montgomery multiplication for long values.
 * p = 21888242871839275222246405745257275088548364400416034343698204186575808495617
 * (reference: tests/poseidon.py:1-3).
 */
#ifndef ORACLE_FR_H
#define ORACLE_FR_H
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t l[4]; } fr;
typedef unsigned __int128 u128;

static const fr FR_P   = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static const fr FR_R2  = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}; /* 2^512 mod p */
static const uint64_t FR_NINV = 0xc2e1f593efffffffULL; /* -p^-1 mod 2^64 */

static inline fr fr_u64(uint64_t v) { fr r = {{v, 0, 0, 0}}; return r; }
static inline int fr_is_small(const fr *a) { return (a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fr_is_zero(const fr *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fr_eq(const fr *a, const fr *b) { return memcmp(a, b, sizeof(fr)) == 0; }
static inline int fr_eq_u64(const fr *a, uint64_t v) { return fr_is_small(a) && a->l[0] == v; }

static inline int fr_geq(const fr *a, const fr *b) {
    for (int i = 3; i >= 0; i--) { if (a->l[i] != b->l[i]) return a->l[i] > b->l[i]; }
    return 1;
}
static inline uint64_t fr_raw_add(fr *r, const fr *a, const fr *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t fr_raw_sub(fr *r, const fr *a, const fr *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline fr fr_add(const fr *a, const fr *b) {
    fr r; uint64_t c = fr_raw_add(&r, a, b);
    if (c || fr_geq(&r, &FR_P)) { fr t; fr_raw_sub(&t, &r, &FR_P); return t; }
    return r;
}
static inline fr fr_sub(const fr *a, const fr *b) {
    fr r; if (fr_raw_sub(&r, a, b)) { fr t; fr_raw_add(&t, &r, &FR_P); return t; }
    return r;
}
static inline fr fr_neg(const fr *a) { fr z = {{0, 0, 0, 0}}; return fr_sub(&z, a); }

/* Montgomery product a*b*2^-256 mod p (CIOS) */
static inline fr fr_mont(const fr *a, const fr *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FR_NINV;
        c = (u128)m * FR_P.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * FR_P.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fr r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fr_geq(&r, &FR_P)) { fr s; fr_raw_sub(&s, &r, &FR_P); return s; }
    return r;
}
static inline fr fr_mul(const fr *a, const fr *b) {
    if (fr_is_small(a) && fr_is_small(b)) {
        u128 pr = (u128)a->l[0] * b->l[0];            /* < 2^128 < p */
        fr r = {{(uint64_t)pr, (uint64_t)(pr >> 64), 0, 0}};
        return r;
    }
    fr t = fr_mont(a, b);
    return fr_mont(&t, &FR_R2);
}
static inline fr fr_mul_u64(const fr *a, uint64_t k) { fr kk = fr_u64(k); return fr_mul(a, &kk); }

/* a^(p-2): Fermat inversion; 0 -> 0 (matches circom's `in!=0 ? 1/in : 0` hint in IsZero) */
static inline fr fr_inv(const fr *a) {
    if (fr_is_zero(a)) return *a;
    fr e = FR_P; e.l[0] -= 2;
    fr am = fr_mont(a, &FR_R2);                        /* Montgomery form of a */
    fr one = fr_u64(1);
    fr acc = fr_mont(&one, &FR_R2);                    /* Montgomery 1 */
    for (int i = 253; i >= 0; i--) {
        acc = fr_mont(&acc, &acc);
        if ((e.l[i >> 6] >> (i & 63)) & 1) acc = fr_mont(&acc, &am);
    }
    return fr_mont(&acc, &one);
}
static inline int fr_bit(const fr *a, int i) { return (int)((a->l[i >> 6] >> (i & 63)) & 1); }
/* 2^k for k < 254 */
static inline fr fr_pow2(int k) { fr r = {{0, 0, 0, 0}}; r.l[k >> 6] = 1ULL << (k & 63); return r; }

#endif
