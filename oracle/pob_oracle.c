/* TEST INFRASTRUCTURE -- CPU oracle.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (proof_of_burn_amd/) never does.
 *
 * What it is: a plain-C restatement of the witness calculator that `circom -c ... --O0` emits for
 * the reference's circuits (reference: Makefile:2-5, tests/test.py:32).  The emitted C++ is not in
 * the reference tree and circom is not installable here, so this file re-derives, template by
 * template from the .circom sources, (a) every signal's value and (b) the O0 wire numbering:
 *   - witness[0] = 1; the main component's block starts at 1 (tests/test.py:40-42);
 *   - a component's block = own outputs | own inputs | own intermediates (declaration order,
 *     arrays row-major) followed by its sub-components' blocks in the order in which they are
 *     initialised (for anonymous components `T(p)(args)` = statement order; for explicitly declared
 *     components = the statement that assigns their last input) -- SURVEY.md app. B hypothesis,
 *     "PARITY UNPINNED": the reference holds no golden .wtns/.sym to check the numbering against.
 *     `ORACLE_DECL_ORDER=1` flips the two circomlib templates where declaration order and
 *     initialisation order differ (Num2Bits_strict, MultiAND) so the alternative can be diffed.
 *   - `===`/assert failures do not abort: the first failing site is recorded (template, source
 *     line) and execution continues; the harness treats any failure as "no witness"
 *     (tests/test.py:65-68).
 * Pinned against: all 56 known-answer suites of the reference (tests/golden/suites.json,
 * generated from /root/reference/tests/testcases by tests/golden/make_golden.py): output signal
 * values and must-fail sets.  Each template cites the reference lines it follows.
 *
 * circomlib templates (UNVENDORED submodule, .gitmodules:1-3) are restated from the published
 * circomlib v2.0.5 circuits (bitify, comparators, gates, mux1, aliascheck, compconstant, poseidon).
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include "fr.h"
#include "poseidon_consts.h"

typedef struct {
    fr *w;
    size_t n, cap;
    int failed;
    char msg[160];
    int decl_order;
    fr inv_cache[8193]; /* inverses of -4096..4096 (index k+4096), lazily filled */
    uint8_t inv_have[8193];
    char sites[1536]; size_t sites_len;   /* EVERY failing assert / === site of the run, "Template:line;" each once, in execution order (the emitted calculator stops at the first;
                                            the HIP path reports the LOWEST site code of a witness: the tests compare it with the lowest of these) */
} ctx;

#define W(i) (c->w[(i)])

static size_t A(ctx *c, size_t n) {
    size_t b = c->n;
    c->n += n;
    if (c->n > c->cap) { fprintf(stderr, "oracle: witness capacity exceeded\n"); abort(); }
    return b;
}
static void fail_at(ctx *c, const char *tpl, int line) {
    if (!c->failed) { c->failed = 1; snprintf(c->msg, sizeof c->msg, "Failed assert in template %s line %d", tpl, line); }
    char one[96]; const int k = snprintf(one, sizeof one, "%s:%d;", tpl, line);
    if (k > 0 && !strstr(c->sites, one) && c->sites_len + (size_t)k < sizeof c->sites) { memcpy(c->sites + c->sites_len, one, (size_t)k + 1); c->sites_len += (size_t)k; }
}
#define REQUIRE(cond, tpl, line) do { if (!(cond)) fail_at(c, tpl, line); } while (0)

static void cp(ctx *c, size_t dst, const fr *src, size_t n) { memcpy(&W(dst), src, n * sizeof(fr)); }

static fr inv_cached(ctx *c, const fr *a) {
    /* small |a| (as signed residue) hit a cache: ~97% of IsZero operands are index differences */
    if (fr_is_small(a) && a->l[0] <= 4096) {
        size_t k = 4096 + a->l[0];
        if (!c->inv_have[k]) { c->inv_cache[k] = fr_inv(a); c->inv_have[k] = 1; }
        return c->inv_cache[k];
    }
    fr n = fr_neg(a);
    if (fr_is_small(&n) && n.l[0] <= 4096) {
        size_t k = 4096 - n.l[0];
        if (!c->inv_have[k]) { c->inv_cache[k] = fr_inv(a); c->inv_have[k] = 1; }
        return c->inv_cache[k];
    }
    return fr_inv(a);
}

/* =====================================================================================
 * circomlib (UNVENDORED) -- gates.circom / bitify.circom / comparators.circom / mux1.circom
 * ===================================================================================== */

/* XOR: out <== a + b - 2*a*b   [out | a, b] */
static size_t XOR(ctx *c, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o + 1) = *a; W(o + 2) = *b;
    fr ab = fr_mul(a, b), s = fr_add(a, b), t = fr_add(&ab, &ab);
    W(o) = fr_sub(&s, &t);
    return o;
}
/* AND: out <== a*b */
static size_t AND(ctx *c, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o + 1) = *a; W(o + 2) = *b; W(o) = fr_mul(a, b); return o;
}
/* OR: out <== a + b - a*b */
static size_t OR(ctx *c, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o + 1) = *a; W(o + 2) = *b;
    fr ab = fr_mul(a, b), s = fr_add(a, b); W(o) = fr_sub(&s, &ab);
    return o;
}
/* Num2Bits(n): out[i] <-- (in>>i)&1; out[i]*(out[i]-1)===0; sum 2^i out[i] === in   [out[n] | in] */
static size_t Num2Bits(ctx *c, int n, const fr *in) {
    size_t o = A(c, (size_t)n + 1); W(o + n) = *in;
    int ok = 1;
    for (int i = 0; i < 256; i++) {
        int b = fr_bit(in, i);
        if (i < n) W(o + i) = fr_u64((uint64_t)b); else if (b) ok = 0;
    }
    REQUIRE(ok, "Num2Bits", 38);   /* lc1 === in */
    return o;
}
/* Bits2Num(n): out <== sum in[i]*2^i   [out | in[n]] */
static size_t Bits2Num(ctx *c, int n, const fr *in) {
    size_t o = A(c, (size_t)n + 1); cp(c, o + 1, in, (size_t)n);
    fr acc = fr_u64(0);
    for (int i = 0; i < n; i++) { fr e = fr_pow2(i), t = fr_mul(&in[i], &e); acc = fr_add(&acc, &t); }
    W(o) = acc;
    return o;
}
/* IsZero: inv <-- in!=0 ? 1/in : 0; out <== -in*inv+1; in*out === 0   [out | in | inv] */
static size_t IsZero(ctx *c, const fr *in) {
    size_t o = A(c, 3); W(o + 1) = *in;
    fr iv = inv_cached(c, in); W(o + 2) = iv;
    W(o) = fr_u64(fr_is_zero(in) ? 1 : 0);
    return o;
}
/* IsEqual: isz.in <== in[1]-in[0]   [out | in[2]] || IsZero */
static size_t IsEqual(ctx *c, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o + 1) = *a; W(o + 2) = *b;
    fr d = fr_sub(b, a);
    size_t z = IsZero(c, &d); W(o) = W(z);
    return o;
}
static size_t IsEqualK(ctx *c, uint64_t k, const fr *b) { fr a = fr_u64(k); return IsEqual(c, &a, b); }
/* LessThan(n): n2b.in <== in[0] + (1<<n) - in[1]; out <== 1 - n2b.out[n]   [out | in[2]] || Num2Bits(n+1) */
static size_t LessThan(ctx *c, int n, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o + 1) = *a; W(o + 2) = *b;
    fr e = fr_pow2(n), s = fr_add(a, &e), v = fr_sub(&s, b);
    size_t nb = Num2Bits(c, n + 1, &v);
    W(o) = fr_u64(fr_is_zero(&W(nb + n)) ? 1 : 0);
    return o;
}
/* LessEqThan(n): lt.in <== [in0, in1+1]   [out | in[2]] || LessThan(n) */
static size_t LessEqThan(ctx *c, int n, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o + 1) = *a; W(o + 2) = *b;
    fr one = fr_u64(1), b1 = fr_add(b, &one);
    size_t lt = LessThan(c, n, a, &b1); W(o) = W(lt);
    return o;
}
/* GreaterEqThan(n): lt.in <== [in1, in0+1] */
static size_t GreaterEqThan(ctx *c, int n, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o + 1) = *a; W(o + 2) = *b;
    fr one = fr_u64(1), a1 = fr_add(a, &one);
    size_t lt = LessThan(c, n, b, &a1); W(o) = W(lt);
    return o;
}
/* MultiAND(n)   [out | in[n]]; n==1 copy; n==2 one AND; else and2=AND(), ands[0]=MultiAND(n\2),
 * ands[1]=MultiAND(n-n\2).  Initialisation order: ands[0], ands[1], and2 (and2's inputs are assigned
 * last); declaration order: and2 first. */
static size_t MultiAND(ctx *c, int n, const fr *in) {
    size_t o = A(c, (size_t)n + 1); cp(c, o + 1, in, (size_t)n);
    if (n == 1) { W(o) = in[0]; return o; }
    if (n == 2) { size_t a = AND(c, &in[0], &in[1]); W(o) = W(a); return o; }
    int n1 = n / 2, n2 = n - n / 2;
    if (c->decl_order) {
        size_t a2 = A(c, 3);
        size_t m0 = MultiAND(c, n1, in), m1 = MultiAND(c, n2, in + n1);
        W(a2 + 1) = W(m0); W(a2 + 2) = W(m1); W(a2) = fr_mul(&W(m0), &W(m1)); W(o) = W(a2);
    } else {
        size_t m0 = MultiAND(c, n1, in), m1 = MultiAND(c, n2, in + n1);
        size_t a2 = AND(c, &W(m0), &W(m1)); W(o) = W(a2);
    }
    return o;
}
/* Mux1: out = (c1-c0)*s + c0   [out | c[2], s] || MultiMux1(1) [out[1] | c[1][2], s] */
static size_t Mux1(ctx *c, const fr *c0, const fr *c1, const fr *s) {
    size_t o = A(c, 4); W(o + 1) = *c0; W(o + 2) = *c1; W(o + 3) = *s;
    size_t m = A(c, 4); W(m + 1) = *c0; W(m + 2) = *c1; W(m + 3) = *s;
    fr d = fr_sub(c1, c0), t = fr_mul(&d, s); W(m) = fr_add(&t, c0);
    W(o) = W(m);
    return o;
}
/* CompConstant(ct)   [out | in[254] | parts[127], sout] || Num2Bits(135); out = (in > ct) */
static size_t CompConstant(ctx *c, const fr *ct, const fr *in) {
    size_t o = A(c, 1 + 254 + 127 + 1); cp(c, o + 1, in, 254);
    size_t parts = o + 255, sout = o + 255 + 127;
    fr b = fr_pow2(128), one = fr_u64(1); b = fr_sub(&b, &one);
    fr a = one, e = one, sum = fr_u64(0);
    for (int i = 0; i < 127; i++) {
        int clsb = fr_bit(ct, 2 * i), cmsb = fr_bit(ct, 2 * i + 1);
        const fr *slsb = &in[2 * i], *smsb = &in[2 * i + 1];
        fr ml = fr_mul(smsb, slsb), p;
        if (!cmsb && !clsb) {            /* -b*smsb*slsb + b*smsb + b*slsb */
            fr t0 = fr_mul(&b, &ml), t1 = fr_mul(&b, smsb), t2 = fr_mul(&b, slsb);
            p = fr_add(&t1, &t2); p = fr_sub(&p, &t0);
        } else if (!cmsb && clsb) {      /* a*smsb*slsb - a*slsb + b*smsb - a*smsb + a */
            fr t0 = fr_mul(&a, &ml), t1 = fr_mul(&a, slsb), t2 = fr_mul(&b, smsb), t3 = fr_mul(&a, smsb);
            p = fr_sub(&t0, &t1); p = fr_add(&p, &t2); p = fr_sub(&p, &t3); p = fr_add(&p, &a);
        } else if (cmsb && !clsb) {      /* b*smsb*slsb - a*smsb + a */
            fr t0 = fr_mul(&b, &ml), t1 = fr_mul(&a, smsb);
            p = fr_sub(&t0, &t1); p = fr_add(&p, &a);
        } else {                         /* -a*smsb*slsb + a */
            fr t0 = fr_mul(&a, &ml); p = fr_sub(&a, &t0);
        }
        W(parts + i) = p; sum = fr_add(&sum, &p);
        b = fr_sub(&b, &e); a = fr_add(&a, &e); e = fr_add(&e, &e);
    }
    W(sout) = sum;
    size_t nb = Num2Bits(c, 135, &sum);
    W(o) = W(nb + 127);
    return o;
}
/* AliasCheck   [ | in[254]] || CompConstant(-1); out === 0 */
static size_t AliasCheck(ctx *c, const fr *in) {
    size_t o = A(c, 254); cp(c, o, in, 254);
    fr ct = FR_P; ct.l[0] -= 1;
    size_t cc = CompConstant(c, &ct, in);
    REQUIRE(fr_is_zero(&W(cc)), "AliasCheck", 31);
    return o;
}
/* Num2Bits_strict   [out[254] | in] || n2b=Num2Bits(254) and aliasCheck (see header for the order) */
static size_t Num2Bits_strict(ctx *c, const fr *in) {
    size_t o = A(c, 255); W(o + 254) = *in;
    if (c->decl_order) {
        fr bits[254];
        for (int i = 0; i < 254; i++) bits[i] = fr_u64((uint64_t)fr_bit(in, i));
        AliasCheck(c, bits);
        size_t nb = Num2Bits(c, 254, in); cp(c, o, &W(nb), 254);
    } else {
        size_t nb = Num2Bits(c, 254, in); cp(c, o, &W(nb), 254);
        AliasCheck(c, &W(o));
    }
    return o;
}

/* ------------------------------------------------------------------ poseidon.circom (circomlib)
 * Optimised schedule; constants derived in tools/gen_poseidon.py.  Children, in initialisation order:
 * ark[0]; 3x{t Sigma, ark, mix(M)}; t Sigma, ark[4], mix[3](P); R_P x {sigmaP, mixS}; 3x{t Sigma, ark,
 * mix(M)}; t Sigma; mixLast.  Sigma [out|in|in2,in4]; Ark/Mix/MixS [out[t]|in[t]]; MixLast [out|in[t]]. */
typedef struct { int t, rp; const uint64_t (*C)[4], (*S)[4], (*M)[4], (*P)[4]; } pos_params;
static pos_params pos_get(int t) {
    pos_params p; p.t = t;
    switch (t) {
    case 2: p.rp = POS_RP_2; p.C = POS_C_2; p.S = POS_S_2; p.M = POS_M_2; p.P = POS_P_2; break;
    case 3: p.rp = POS_RP_3; p.C = POS_C_3; p.S = POS_S_3; p.M = POS_M_3; p.P = POS_P_3; break;
    case 4: p.rp = POS_RP_4; p.C = POS_C_4; p.S = POS_S_4; p.M = POS_M_4; p.P = POS_P_4; break;
    default: p.rp = POS_RP_5; p.C = POS_C_5; p.S = POS_S_5; p.M = POS_M_5; p.P = POS_P_5; break;
    }
    return p;
}
static fr K(const uint64_t k[4]) { fr r = {{k[0], k[1], k[2], k[3]}}; return r; }
static size_t Sigma(ctx *c, const fr *in) {
    size_t o = A(c, 4); W(o + 1) = *in;
    W(o + 2) = fr_mul(in, in); W(o + 3) = fr_mul(&W(o + 2), &W(o + 2)); W(o) = fr_mul(&W(o + 3), in);
    return o;
}
static size_t Ark(ctx *c, int t, const uint64_t (*C)[4], int r, const fr *in) {
    size_t o = A(c, 2 * (size_t)t); cp(c, o + t, in, (size_t)t);
    for (int i = 0; i < t; i++) { fr k = K(C[i + r]); W(o + i) = fr_add(&in[i], &k); }
    return o;
}
/* Mix: out[i] = sum_j M[j][i]*in[j] with circomlib's table; ours is A[i][j] row-major, new[i]=sum_j A[i][j] old[j] */
static size_t Mix(ctx *c, int t, const uint64_t (*M)[4], const fr *in) {
    size_t o = A(c, 2 * (size_t)t); cp(c, o + t, in, (size_t)t);
    for (int i = 0; i < t; i++) {
        fr acc = fr_u64(0);
        for (int j = 0; j < t; j++) { fr k = K(M[i * t + j]), p = fr_mul(&k, &in[j]); acc = fr_add(&acc, &p); }
        W(o + i) = acc;
    }
    return o;
}
static size_t MixS(ctx *c, int t, const uint64_t (*S)[4], int r, const fr *in) {
    size_t o = A(c, 2 * (size_t)t); cp(c, o + t, in, (size_t)t);
    int base = (2 * t - 1) * r;
    fr acc = fr_u64(0);
    for (int i = 0; i < t; i++) { fr k = K(S[base + i]), p = fr_mul(&k, &in[i]); acc = fr_add(&acc, &p); }
    W(o) = acc;
    for (int i = 1; i < t; i++) { fr k = K(S[base + t + i - 1]), p = fr_mul(&in[0], &k); W(o + i) = fr_add(&in[i], &p); }
    return o;
}
static size_t MixLast(ctx *c, int t, const uint64_t (*M)[4], int s, const fr *in) {
    size_t o = A(c, 1 + (size_t)t); cp(c, o + 1, in, (size_t)t);
    fr acc = fr_u64(0);
    for (int j = 0; j < t; j++) { fr k = K(M[s * t + j]), p = fr_mul(&k, &in[j]); acc = fr_add(&acc, &p); }
    W(o) = acc;
    return o;
}
static size_t PoseidonEx(ctx *c, int nIn, const fr *inputs, const fr *initialState) {
    pos_params pp = pos_get(nIn + 1);
    int t = pp.t, rp = pp.rp;
    size_t o = A(c, 1 + (size_t)nIn + 1); cp(c, o + 1, inputs, (size_t)nIn); W(o + 1 + nIn) = *initialState;
    fr st[8], tmp[8];
    st[0] = *initialState; for (int j = 1; j < t; j++) st[j] = inputs[j - 1];
    size_t x = Ark(c, t, pp.C, 0, st); for (int j = 0; j < t; j++) st[j] = W(x + j);
    for (int r = 0; r < 3; r++) {
        for (int j = 0; j < t; j++) { size_t s = Sigma(c, &st[j]); tmp[j] = W(s); }
        x = Ark(c, t, pp.C, (r + 1) * t, tmp); for (int j = 0; j < t; j++) tmp[j] = W(x + j);
        x = Mix(c, t, pp.M, tmp); for (int j = 0; j < t; j++) st[j] = W(x + j);
    }
    for (int j = 0; j < t; j++) { size_t s = Sigma(c, &st[j]); tmp[j] = W(s); }
    x = Ark(c, t, pp.C, 4 * t, tmp); for (int j = 0; j < t; j++) tmp[j] = W(x + j);
    x = Mix(c, t, pp.P, tmp); for (int j = 0; j < t; j++) st[j] = W(x + j);
    for (int r = 0; r < rp; r++) {
        size_t s = Sigma(c, &st[0]);
        fr k = K(pp.C[5 * t + r]); tmp[0] = fr_add(&W(s), &k);
        for (int j = 1; j < t; j++) tmp[j] = st[j];
        x = MixS(c, t, pp.S, r, tmp); for (int j = 0; j < t; j++) st[j] = W(x + j);
    }
    for (int r = 0; r < 3; r++) {
        for (int j = 0; j < t; j++) { size_t s = Sigma(c, &st[j]); tmp[j] = W(s); }
        x = Ark(c, t, pp.C, 5 * t + rp + r * t, tmp); for (int j = 0; j < t; j++) tmp[j] = W(x + j);
        x = Mix(c, t, pp.M, tmp); for (int j = 0; j < t; j++) st[j] = W(x + j);
    }
    for (int j = 0; j < t; j++) { size_t s = Sigma(c, &st[j]); tmp[j] = W(s); }
    x = MixLast(c, t, pp.M, 0, tmp);
    W(o) = W(x);
    return o;
}
/* Poseidon(n)   [out | inputs[n]] || PoseidonEx(n,1) [out[1] | inputs[n], initialState] */
static size_t Poseidon(ctx *c, int nIn, const fr *inputs) {
    size_t o = A(c, 1 + (size_t)nIn); cp(c, o + 1, inputs, (size_t)nIn);
    fr z = fr_u64(0);
    size_t e = PoseidonEx(c, nIn, inputs, &z);
    W(o) = W(e);
    return o;
}

/* =====================================================================================
 * circuits/utils/assert.circom
 * ===================================================================================== */
/* AssertBits(B) assert.circom:13-17   [ | in | bits[B]] || Num2Bits(B) */
static size_t AssertBits(ctx *c, int B, const fr *in) {
    size_t o = A(c, 1 + (size_t)B); W(o) = *in;
    size_t nb = Num2Bits(c, B, in); cp(c, o + 1, &W(nb), (size_t)B);
    return o;
}
/* AssertByteString(N) assert.circom:26-31 */
static size_t AssertByteString(ctx *c, int N, const fr *in) {
    size_t o = A(c, (size_t)N); cp(c, o, in, (size_t)N);
    for (int i = 0; i < N; i++) AssertBits(c, 8, &in[i]);
    return o;
}
/* AssertLessThan(B) assert.circom:40-47   [ | a, b | out] || AssertBits, AssertBits, LessThan; out===1 */
static size_t AssertLessThan(ctx *c, int B, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o) = *a; W(o + 1) = *b;
    AssertBits(c, B, a); AssertBits(c, B, b);
    size_t l = LessThan(c, B, a, b); W(o + 2) = W(l);
    REQUIRE(fr_eq_u64(&W(o + 2), 1), "AssertLessThan", 46);
    return o;
}
/* AssertLessEqThan(B) assert.circom:56-63 */
static size_t AssertLessEqThan(ctx *c, int B, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o) = *a; W(o + 1) = *b;
    AssertBits(c, B, a); AssertBits(c, B, b);
    size_t l = LessEqThan(c, B, a, b); W(o + 2) = W(l);
    REQUIRE(fr_eq_u64(&W(o + 2), 1), "AssertLessEqThan", 62);
    return o;
}
/* AssertGreaterEqThan(B) assert.circom:72-79 */
static size_t AssertGreaterEqThan(ctx *c, int B, const fr *a, const fr *b) {
    size_t o = A(c, 3); W(o) = *a; W(o + 1) = *b;
    AssertBits(c, B, a); AssertBits(c, B, b);
    size_t l = GreaterEqThan(c, B, a, b); W(o + 2) = W(l);
    REQUIRE(fr_eq_u64(&W(o + 2), 1), "AssertGreaterEqThan", 78);
    return o;
}

/* =====================================================================================
 * circuits/utils/array.circom
 * ===================================================================================== */
/* Filter(N) array.circom:26-39   [out[N] | in | isEq[N]] || IsEqual x N */
static size_t Filter(ctx *c, int N, const fr *in) {
    size_t o = A(c, 2 * (size_t)N + 1); W(o + N) = *in;
    size_t isEq = o + N + 1;
    fr one = fr_u64(1);
    for (int i = 0; i < N; i++) {
        size_t e = IsEqualK(c, (uint64_t)i, in); W(isEq + i) = W(e);
        fr ne = fr_sub(&one, &W(isEq + i));
        W(o + i) = i > 0 ? fr_mul(&W(o + i - 1), &ne) : ne;
    }
    return o;
}
/* Fit(M,N) array.circom:47-57   [out[N] | in[M]] */
static size_t Fit(ctx *c, int M, int N, const fr *in) {
    size_t o = A(c, (size_t)N + M); cp(c, o + N, in, (size_t)M);
    for (int i = 0; i < N; i++) W(o + i) = i < M ? in[i] : fr_u64(0);
    return o;
}
/* Flatten(M,N) array.circom:64-72, Reshape(M,N) :79-87: identity on row-major data   [out[MN] | in[MN]] */
static size_t Flatten(ctx *c, int M, int N, const fr *in) {
    size_t o = A(c, 2 * (size_t)M * N); cp(c, o, in, (size_t)M * N); cp(c, o + (size_t)M * N, in, (size_t)M * N);
    return o;
}
#define Reshape Flatten
/* Reverse(N) array.circom:94-100 */
static size_t Reverse(ctx *c, int N, const fr *in) {
    size_t o = A(c, 2 * (size_t)N); cp(c, o + N, in, (size_t)N);
    for (int i = 0; i < N; i++) W(o + i) = in[N - 1 - i];
    return o;
}

/* =====================================================================================
 * circuits/utils/divide.circom:17-33   [out, rem | a, b] || AssertLessThan(N)(rem,b), AssertLessEqThan(N)(out,a)
 * out <-- a \ b; rem <-- a % b on canonical representatives; out*b + rem === a
 * ===================================================================================== */
static size_t Divide(ctx *c, int N, const fr *a, const fr *b) {
    size_t o = A(c, 4); W(o + 2) = *a; W(o + 3) = *b;
    fr q = fr_u64(0), r = fr_u64(0);
    if (fr_is_small(a) && fr_is_small(b) && b->l[0]) { q = fr_u64(a->l[0] / b->l[0]); r = fr_u64(a->l[0] % b->l[0]); }
    else if (fr_is_small(b) && b->l[0]) {           /* long / short: schoolbook on 64-bit limbs */
        u128 rem = 0; for (int i = 3; i >= 0; i--) { u128 cur = (rem << 64) | a->l[i]; q.l[i] = (uint64_t)(cur / b->l[0]); rem = cur % b->l[0]; }
        r = fr_u64((uint64_t)rem);
    } else if (!fr_is_zero(b)) {                     /* long / long: a < 2b*2^k shift-subtract */
        fr rr = fr_u64(0);
        for (int i = 255; i >= 0; i--) {
            fr t; fr_raw_add(&t, &rr, &rr); t.l[0] |= (uint64_t)fr_bit(a, i); rr = t;
            if (fr_geq(&rr, b)) { fr u; fr_raw_sub(&u, &rr, b); rr = u; q.l[i >> 6] |= 1ULL << (i & 63); }
        }
        r = rr;
    }
    W(o) = q; W(o + 1) = r;
    AssertLessThan(c, N, &W(o + 1), b);
    AssertLessEqThan(c, N, &W(o), a);
    fr qb = fr_mul(&W(o), b), s = fr_add(&qb, &W(o + 1));
    REQUIRE(fr_eq(&s, a), "Divide", 32);
    return o;
}

/* =====================================================================================
 * circuits/utils/selector.circom
 * ===================================================================================== */
/* Selector(n) selector.circom:21-46   [out | vals[n], select | isEq[n], sum[n+1]] || IsEqual x n; sumIsEq===1 */
static size_t Selector(ctx *c, int n, const fr *vals, const fr *select) {
    size_t o = A(c, 1 + (size_t)n + 1 + n + n + 1); cp(c, o + 1, vals, (size_t)n); W(o + 1 + n) = *select;
    size_t isEq = o + 2 + n, sum = isEq + n;
    fr cnt = fr_u64(0);
    for (int i = 0; i < n; i++) {
        fr k = fr_u64((uint64_t)i);
        size_t e = IsEqual(c, select, &k); W(isEq + i) = W(e);
        cnt = fr_add(&cnt, &W(isEq + i));
        fr p = fr_mul(&W(isEq + i), &vals[i]); W(sum + i + 1) = fr_add(&W(sum + i), &p);
    }
    REQUIRE(fr_eq_u64(&cnt, 1), "Selector", 43);
    W(o) = W(sum + n);
    return o;
}
/* SelectorArray1D(n,p) selector.circom:62-77   [out[p] | arrays[n][p], select | arraysT[p][n]] || Selector(n) x p */
static size_t SelectorArray1D(ctx *c, int n, int p, const fr *arrays, const fr *select) {
    size_t np = (size_t)n * p;
    size_t o = A(c, (size_t)p + np + 1 + np); cp(c, o + p, arrays, np); W(o + p + np) = *select;
    size_t T = o + p + np + 1;
    for (int i = 0; i < n; i++) for (int j = 0; j < p; j++) W(T + (size_t)j * n + i) = arrays[(size_t)i * p + j];
    for (int i = 0; i < p; i++) { size_t s = Selector(c, n, &W(T + (size_t)i * n), select); W(o + i) = W(s); }
    return o;
}
/* SelectorArray2D(n,p,q) selector.circom:91-111 */
static size_t SelectorArray2D(ctx *c, int n, int p, int q, const fr *arrays, const fr *select) {
    size_t pq = (size_t)p * q, npq = pq * n;
    size_t o = A(c, pq + npq + 1 + npq); cp(c, o + pq, arrays, npq); W(o + pq + npq) = *select;
    size_t T = o + pq + npq + 1;
    for (int i = 0; i < n; i++) for (size_t jk = 0; jk < pq; jk++) W(T + jk * n + i) = arrays[(size_t)i * pq + jk];
    for (size_t jk = 0; jk < pq; jk++) { size_t s = Selector(c, n, &W(T + jk * n), select); W(o + jk) = W(s); }
    return o;
}

/* =====================================================================================
 * circuits/utils/shift.circom
 * ===================================================================================== */
/* ShiftLeft(n) shift.circom:17-37   [out[n] | in[n], count | isEq[n][n], temp[n][n]] || AssertLessEqThan(16), IsEqual x n^2 */
static size_t ShiftLeft(ctx *c, int n, const fr *in, const fr *count) {
    size_t nn = (size_t)n * n;
    size_t o = A(c, 2 * (size_t)n + 1 + 2 * nn); cp(c, o + n, in, (size_t)n); W(o + 2 * n) = *count;
    size_t isEq = o + 2 * n + 1, temp = isEq + nn;
    fr fn = fr_u64((uint64_t)n);
    AssertLessEqThan(c, 16, count, &fn);
    for (int i = 0; i < n; i++) {
        fr acc = fr_u64(0);
        for (int j = 0; j < n; j++) {
            fr fi = fr_u64((uint64_t)i), fj = fr_u64((uint64_t)j), jc = fr_sub(&fj, count);
            size_t e = IsEqual(c, &fi, &jc); W(isEq + (size_t)i * n + j) = W(e);
            W(temp + (size_t)i * n + j) = fr_mul(&W(e), &in[j]);
            acc = fr_add(&acc, &W(temp + (size_t)i * n + j));
        }
        W(o + i) = acc;
    }
    return o;
}
/* ShiftRight(n,maxShift) shift.circom:51-75   [out[n+ms] | in[n], count | isEq[ms+1], temps[ms+1][n]] */
static size_t ShiftRight(ctx *c, int n, int ms, const fr *in, const fr *count) {
    size_t o = A(c, (size_t)n + ms + n + 1 + (ms + 1) + (size_t)(ms + 1) * n);
    cp(c, o + n + ms, in, (size_t)n); W(o + n + ms + n) = *count;
    size_t isEq = o + n + ms + n + 1, temps = isEq + ms + 1;
    fr fm = fr_u64((uint64_t)ms);
    AssertLessEqThan(c, 16, count, &fm);
    for (int i = 0; i <= ms; i++) {
        size_t e = IsEqualK(c, (uint64_t)i, count); W(isEq + i) = W(e);
        for (int j = 0; j < n; j++) {
            W(temps + (size_t)i * n + j) = fr_mul(&W(isEq + i), &in[j]);
            W(o + i + j) = fr_add(&W(o + i + j), &W(temps + (size_t)i * n + j));
        }
    }
    return o;
}

/* =====================================================================================
 * circuits/utils/concat.circom
 * ===================================================================================== */
/* Mask(n) concat.circom:18-30   [out[n] | in[n], count | filter[n]] || Filter(n) */
static size_t Mask(ctx *c, int n, const fr *in, const fr *count) {
    size_t o = A(c, 3 * (size_t)n + 1); cp(c, o + n, in, (size_t)n); W(o + 2 * n) = *count;
    size_t f = Filter(c, n, count); cp(c, o + 2 * n + 1, &W(f), (size_t)n);
    for (int i = 0; i < n; i++) W(o + i) = fr_mul(&in[i], &W(o + 2 * n + 1 + i));
    return o;
}
/* Concat(A,B) concat.circom:47-83   [out[A+B], outLen | a[A], aLen, b[B], bLen | maskedA, maskedB, shiftedB] */
static size_t Concat(ctx *c, int La, int Lb, const fr *a, const fr *aLen, const fr *b, const fr *bLen) {
    size_t o = A(c, (size_t)(La + Lb) + 1 + La + 1 + Lb + 1 + La + Lb + La + Lb);
    size_t ia = o + La + Lb + 1, ial = ia + La, ib = ial + 1, ibl = ib + Lb, mA = ibl + 1, mB = mA + La, sB = mB + Lb;
    cp(c, ia, a, (size_t)La); W(ial) = *aLen; cp(c, ib, b, (size_t)Lb); W(ibl) = *bLen;
    fr fa = fr_u64((uint64_t)La), fb = fr_u64((uint64_t)Lb);
    AssertLessEqThan(c, 16, aLen, &fa);
    AssertLessEqThan(c, 16, bLen, &fb);
    size_t m = Mask(c, La, a, aLen); cp(c, mA, &W(m), (size_t)La);
    m = Mask(c, Lb, b, bLen); cp(c, mB, &W(m), (size_t)Lb);
    size_t s = ShiftRight(c, Lb, La, &W(mB), aLen); cp(c, sB, &W(s), (size_t)(La + Lb));
    for (int i = 0; i < La + Lb; i++) W(o + i) = i < La ? fr_add(&W(mA + i), &W(sB + i)) : W(sB + i);
    W(o + La + Lb) = fr_add(aLen, bLen);
    return o;
}

/* =====================================================================================
 * circuits/utils/convert.circom
 * ===================================================================================== */
/* LittleEndianBytes2Num(N) convert.circom:12-26   [out | in[N]] || AssertByteString(N) */
static size_t LittleEndianBytes2Num(ctx *c, int N, const fr *in) {
    size_t o = A(c, 1 + (size_t)N); cp(c, o + 1, in, (size_t)N);
    AssertByteString(c, N, in);
    fr acc = fr_u64(0);
    for (int i = 0; i < N; i++) { fr e = fr_pow2(8 * i), t = fr_mul(&e, &in[i]); acc = fr_add(&acc, &t); }
    W(o) = acc;
    return o;
}
/* BigEndianBytes2Num(N) convert.circom:33-39   [out | in[N] | inReversed[N]] || Reverse, LittleEndianBytes2Num */
static size_t BigEndianBytes2Num(ctx *c, int N, const fr *in) {
    size_t o = A(c, 1 + 2 * (size_t)N); cp(c, o + 1, in, (size_t)N);
    size_t r = Reverse(c, N, in); cp(c, o + 1 + N, &W(r), (size_t)N);
    size_t l = LittleEndianBytes2Num(c, N, &W(o + 1 + N)); W(o) = W(l);
    return o;
}
/* Num2BitsSafe(N) convert.circom:46-56 */
static size_t Num2BitsSafe(ctx *c, int N, const fr *in) {
    if (N >= 254) {
        size_t o = A(c, (size_t)N + 1 + 254); W(o + N) = *in;
        size_t s = Num2Bits_strict(c, in); cp(c, o + N + 1, &W(s), 254);
        size_t f = Fit(c, 254, N, &W(o + N + 1)); cp(c, o, &W(f), (size_t)N);
        return o;
    }
    size_t o = A(c, (size_t)N + 1); W(o + N) = *in;
    size_t nb = Num2Bits(c, N, in); cp(c, o, &W(nb), (size_t)N);
    return o;
}
/* Num2LittleEndianBytes(N) convert.circom:69-83   [out[N] | in | bits[8N], byteArrays[N][8]] */
static size_t Num2LittleEndianBytes(ctx *c, int N, const fr *in) {
    size_t o = A(c, (size_t)N + 1 + 16 * (size_t)N); W(o + N) = *in;
    size_t bits = o + N + 1, ba = bits + 8 * (size_t)N;
    size_t s = Num2BitsSafe(c, 8 * N, in); cp(c, bits, &W(s), 8 * (size_t)N);
    size_t r = Reshape(c, N, 8, &W(bits)); cp(c, ba, &W(r), 8 * (size_t)N);
    for (int i = 0; i < N; i++) { size_t b = Bits2Num(c, 8, &W(ba + 8 * (size_t)i)); W(o + i) = W(b); }
    return o;
}
/* Num2BigEndianBytes(N) convert.circom:90-96   [out[N] | in | littleEndian[N]] */
static size_t Num2BigEndianBytes(ctx *c, int N, const fr *in) {
    size_t o = A(c, 2 * (size_t)N + 1); W(o + N) = *in;
    size_t l = Num2LittleEndianBytes(c, N, in); cp(c, o + N + 1, &W(l), (size_t)N);
    size_t r = Reverse(c, N, &W(o + N + 1)); cp(c, o, &W(r), (size_t)N);
    return o;
}
/* Bytes2Nibbles(N) convert.circom:103-121   [out[2N] | in[N] | inDecomposed[N][8]] || Num2Bits(8) x N */
static size_t Bytes2Nibbles(ctx *c, int N, const fr *in) {
    size_t o = A(c, 2 * (size_t)N + N + 8 * (size_t)N); cp(c, o + 2 * N, in, (size_t)N);
    size_t dec = o + 3 * (size_t)N;
    for (int i = 0; i < N; i++) {
        size_t nb = Num2Bits(c, 8, &in[i]); cp(c, dec + 8 * (size_t)i, &W(nb), 8);
        uint64_t lo = 0, hi = 0;
        for (int j = 0; j < 4; j++) { lo += W(nb + j).l[0] << j; hi += W(nb + j + 4).l[0] << j; }
        W(o + 2 * i) = fr_u64(hi); W(o + 2 * i + 1) = fr_u64(lo);
    }
    return o;
}
/* Nibbles2Bytes(n) convert.circom:132-141   [bytes[n] | nibbles[2n]] || AssertBits(4) x 2n */
static size_t Nibbles2Bytes(ctx *c, int n, const fr *nib) {
    size_t o = A(c, 3 * (size_t)n); cp(c, o + n, nib, 2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        AssertBits(c, 4, &nib[2 * i]); AssertBits(c, 4, &nib[2 * i + 1]);
        fr t = fr_mul_u64(&nib[2 * i], 16); W(o + i) = fr_add(&t, &nib[2 * i + 1]);
    }
    return o;
}

/* =====================================================================================
 * circuits/utils/keccak.circom
 * ===================================================================================== */
/* ShR(n,r) keccak.circom:19-30: out[i] = in[i+r] or 0   [out[n] | in[n]] */
static size_t ShR(ctx *c, int n, int r, const fr *in) {
    size_t o = A(c, 2 * (size_t)n); cp(c, o + n, in, (size_t)n);
    for (int i = 0; i < n; i++) if (i + r < n) W(o + i) = in[i + r];
    return o;
}
/* ShL(n,r) keccak.circom:40-51: out[i] = in[i-r] or 0 */
static size_t ShL(ctx *c, int n, int r, const fr *in) {
    size_t o = A(c, 2 * (size_t)n); cp(c, o + n, in, (size_t)n);
    for (int i = r; i < n; i++) W(o + i) = in[i - r];
    return o;
}
/* XorArray(n) keccak.circom:77-85   [out[n] | a[n], b[n]] || XOR x n; OrArray :105-113; AndArray :120-128 */
static size_t GateArray(ctx *c, int n, const fr *a, const fr *b, size_t (*g)(ctx *, const fr *, const fr *)) {
    size_t o = A(c, 3 * (size_t)n); cp(c, o + n, a, (size_t)n); cp(c, o + 2 * n, b, (size_t)n);
    for (int i = 0; i < n; i++) { size_t x = g(c, &a[i], &b[i]); W(o + i) = W(x); }
    return o;
}
#define XorArray(c, n, a, b) GateArray(c, n, a, b, XOR)
#define OrArray(c, n, a, b) GateArray(c, n, a, b, OR)
#define AndArray(c, n, a, b) GateArray(c, n, a, b, AND)
/* NotArray(n) keccak.circom:92-98: out = 1 - a   [out[n] | a[n]] */
static size_t NotArray(ctx *c, int n, const fr *a) {
    size_t o = A(c, 2 * (size_t)n); cp(c, o + n, a, (size_t)n);
    fr one = fr_u64(1);
    for (int i = 0; i < n; i++) W(o + i) = fr_sub(&one, &a[i]);
    return o;
}
/* Xor5(n) keccak.circom:58-70   [out | a,b,c,d,e | xor_ab, xor_abc, xor_abcd] || XorArray x 4 */
static size_t Xor5(ctx *c, int n, const fr *a, const fr *b, const fr *cc, const fr *d, const fr *e) {
    size_t N = (size_t)n, o = A(c, 9 * N);
    cp(c, o + N, a, N); cp(c, o + 2 * N, b, N); cp(c, o + 3 * N, cc, N); cp(c, o + 4 * N, d, N); cp(c, o + 5 * N, e, N);
    size_t x = XorArray(c, n, a, b); cp(c, o + 6 * N, &W(x), N);
    x = XorArray(c, n, &W(o + 6 * N), cc); cp(c, o + 7 * N, &W(x), N);
    x = XorArray(c, n, &W(o + 7 * N), d); cp(c, o + 8 * N, &W(x), N);
    x = XorArray(c, n, &W(o + 8 * N), e); cp(c, o, &W(x), N);
    return o;
}
/* D keccak.circom:135-144: out = b ^ (a<<1 | a>>63)   [out | a, b | aux0, aux1, aux2] || ShL(64,1), ShR(64,63), OrArray, XorArray */
static size_t Dt(ctx *c, const fr *a, const fr *b) {
    size_t o = A(c, 6 * 64); cp(c, o + 64, a, 64); cp(c, o + 128, b, 64);
    size_t x = ShL(c, 64, 1, a); cp(c, o + 192, &W(x), 64);
    x = ShR(c, 64, 63, a); cp(c, o + 256, &W(x), 64);
    x = OrArray(c, 64, &W(o + 192), &W(o + 256)); cp(c, o + 320, &W(x), 64);
    x = XorArray(c, 64, b, &W(o + 320)); cp(c, o, &W(x), 64);
    return o;
}
/* Theta keccak.circom:151-170   [out[25][64] | in[25][64] | c[5][64], d[5][64]] || Xor5 x5, D x5, XorArray x25 (i outer, j inner) */
static size_t Theta(ctx *c, const fr *in) {
    size_t o = A(c, 1600 + 1600 + 320 + 320), I = o + 1600, C = o + 3200, Dd = o + 3520;
    cp(c, I, in, 1600);
    for (int i = 0; i < 5; i++) {
        size_t x = Xor5(c, 64, &W(I + 64 * i), &W(I + 64 * (5 + i)), &W(I + 64 * (10 + i)), &W(I + 64 * (15 + i)), &W(I + 64 * (20 + i)));
        cp(c, C + 64 * i, &W(x), 64);
    }
    for (int i = 0; i < 5; i++) {
        size_t x = Dt(c, &W(C + 64 * ((i + 1) % 5)), &W(C + 64 * ((i + 4) % 5)));
        cp(c, Dd + 64 * i, &W(x), 64);
    }
    for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) {
        size_t x = XorArray(c, 64, &W(I + 64 * (i + j * 5)), &W(Dd + 64 * i));
        cp(c, o + 64 * (i + j * 5), &W(x), 64);
    }
    return o;
}
/* stepRhoPi(shl,shr) keccak.circom:177-184   [out | a | aux0, aux1] || ShR(64,shr), ShL(64,shl), OrArray */
static size_t stepRhoPi(ctx *c, int shl, int shr, const fr *a) {
    size_t o = A(c, 256); cp(c, o + 64, a, 64);
    size_t x = ShR(c, 64, shr, a); cp(c, o + 128, &W(x), 64);
    x = ShL(c, 64, shl, a); cp(c, o + 192, &W(x), 64);
    x = OrArray(c, 64, &W(o + 128), &W(o + 192)); cp(c, o, &W(x), 64);
    return o;
}
/* RhoPi keccak.circom:191-204 */
static size_t RhoPi(ctx *c, const fr *in) {
    static const int rot[25] = {1, 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    size_t o = A(c, 3200), I = o + 1600; cp(c, I, in, 1600);
    cp(c, o, &W(I), 64);
    for (int i = 0; i < 24; i++) {
        int shl = ((i + 1) * (i + 2) / 2) % 64;
        size_t x = stepRhoPi(c, shl, 64 - shl, &W(I + 64 * rot[i]));
        cp(c, o + 64 * rot[i + 1], &W(x), 64);
    }
    return o;
}
/* stepChi keccak.circom:212-221: out = a ^ (~b & c)   [out | a,b,c | bXor, bc] || NotArray, AndArray, XorArray */
static size_t stepChi(ctx *c, const fr *a, const fr *b, const fr *cc) {
    size_t o = A(c, 384); cp(c, o + 64, a, 64); cp(c, o + 128, b, 64); cp(c, o + 192, cc, 64);
    size_t x = NotArray(c, 64, b); cp(c, o + 256, &W(x), 64);
    x = AndArray(c, 64, &W(o + 256), cc); cp(c, o + 320, &W(x), 64);
    x = XorArray(c, 64, a, &W(o + 320)); cp(c, o, &W(x), 64);
    return o;
}
/* Chi keccak.circom:228-241 */
static size_t Chi(ctx *c, const fr *in) {
    size_t o = A(c, 3200), I = o + 1600; cp(c, I, in, 1600);
    for (int i = 0; i < 25; i++) {
        size_t x;
        if (i % 5 == 3) x = stepChi(c, &W(I + 64 * i), &W(I + 64 * (i + 1)), &W(I + 64 * (i - 3)));
        else if (i % 5 == 4) x = stepChi(c, &W(I + 64 * i), &W(I + 64 * (i - 4)), &W(I + 64 * (i - 3)));
        else x = stepChi(c, &W(I + 64 * i), &W(I + 64 * (i + 1)), &W(I + 64 * (i + 2)));
        cp(c, o + 64 * i, &W(x), 64);
    }
    return o;
}
static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
/* RoundConstants(r) keccak.circom:248-266   [out[64]] */
static size_t RoundConstants(ctx *c, int r) {
    size_t o = A(c, 64);
    for (int i = 0; i < 64; i++) W(o + i) = fr_u64((KECCAK_RC[r] >> i) & 1);
    return o;
}
/* Iota(r) keccak.circom:273-283   [out | in | roundConstants[64]] || RoundConstants(r), XorArray */
static size_t Iota(ctx *c, int r, const fr *in) {
    size_t o = A(c, 3264), I = o + 1600, RC = o + 3200; cp(c, I, in, 1600);
    size_t x = RoundConstants(c, r); cp(c, RC, &W(x), 64);
    x = XorArray(c, 64, &W(I), &W(RC)); cp(c, o, &W(x), 64);
    cp(c, o + 64, &W(I + 64), 1536);
    return o;
}
/* KeccakfRound(r) keccak.circom:290-297   [out | in | theta, rhopi, chi] */
static size_t KeccakfRound(ctx *c, int r, const fr *in) {
    size_t o = A(c, 8000); cp(c, o + 1600, in, 1600);
    size_t x = Theta(c, in); cp(c, o + 3200, &W(x), 1600);
    x = RhoPi(c, &W(o + 3200)); cp(c, o + 4800, &W(x), 1600);
    x = Chi(c, &W(o + 4800)); cp(c, o + 6400, &W(x), 1600);
    x = Iota(c, r, &W(o + 6400)); cp(c, o, &W(x), 1600);
    return o;
}
/* Keccakf keccak.circom:356-367   [out | in | midRound[25][25][64]] */
static size_t Keccakf(ctx *c, const fr *in) {
    size_t o = A(c, 3200 + 25 * 1600), mid = o + 3200; cp(c, o + 1600, in, 1600);
    cp(c, mid, in, 1600);
    for (int i = 0; i < 24; i++) { size_t x = KeccakfRound(c, i, &W(mid + 1600 * (size_t)i)); cp(c, mid + 1600 * (size_t)(i + 1), &W(x), 1600); }
    cp(c, o, &W(mid + 1600 * 24), 1600);
    return o;
}
/* Absorb keccak.circom:304-323   [out[25][64] | s[25][64], block[17][64] | aux[25][64]] || XorArray x17, Keccakf */
static size_t Absorb(ctx *c, const fr *s, const fr *block) {
    size_t o = A(c, 1600 + 1600 + 1088 + 1600), aux = o + 4288;
    cp(c, o + 1600, s, 1600); cp(c, o + 3200, block, 1088);
    for (int i = 0; i < 25; i++) {
        if (i < 17) { size_t x = XorArray(c, 64, &s[64 * i], &block[64 * i]); cp(c, aux + 64 * i, &W(x), 64); }
        else cp(c, aux + 64 * i, &s[64 * i], 64);
    }
    size_t k = Keccakf(c, &W(aux)); cp(c, o, &W(k), 1600);
    return o;
}
/* Final(n) keccak.circom:330-349   [out[25][64] | in[n][17][64], blocks | s[n+1][25][64]] || Absorb x n, SelectorArray2D(n+1,25,64) */
static size_t Final(ctx *c, int n, const fr *in, const fr *blocks) {
    size_t o = A(c, 1600 + 1088 * (size_t)n + 1 + 1600 * (size_t)(n + 1)), S = o + 1600 + 1088 * (size_t)n + 1;
    cp(c, o + 1600, in, 1088 * (size_t)n); W(o + 1600 + 1088 * (size_t)n) = *blocks;
    for (int b = 0; b < n; b++) { size_t x = Absorb(c, &W(S + 1600 * (size_t)b), &in[1088 * (size_t)b]); cp(c, S + 1600 * (size_t)(b + 1), &W(x), 1600); }
    size_t x = SelectorArray2D(c, n + 1, 25, 64, &W(S), blocks); cp(c, o, &W(x), 1600);
    return o;
}
/* Keccak(n) keccak.circom:374-385   [out[256] | in[n][17][64], blocks | finalState[25][64]] */
static size_t Keccak(ctx *c, int n, const fr *in, const fr *blocks) {
    size_t o = A(c, 256 + 1088 * (size_t)n + 1 + 1600), fs = o + 256 + 1088 * (size_t)n + 1;
    cp(c, o + 256, in, 1088 * (size_t)n); W(o + 256 + 1088 * (size_t)n) = *blocks;
    size_t f = Final(c, n, in, blocks); cp(c, fs, &W(f), 1600);
    cp(c, o, &W(fs), 256);
    return o;
}
/* Pad(maxBlocks, blockSize) keccak.circom:412-446
 * [out[m], numBlocks | in[m], inLen | div, rem, filter[m+1], isEq[m], isLast[m]] || Divide(16), AssertLessEqThan(16), IsEqual x m, IsEqual x m */
static size_t Pad(ctx *c, int mb, int bs, const fr *in, const fr *inLen) {
    size_t m = (size_t)mb * bs;
    size_t o = A(c, m + 1 + m + 1 + 2 + (m + 1) + m + m);
    size_t I = o + m + 1, IL = I + m, dv = IL + 1, rm = dv + 1, flt = rm + 1, isEq = flt + m + 1, isLast = isEq + m;
    cp(c, I, in, m); W(IL) = *inLen;
    fr fbs = fr_u64((uint64_t)bs), one = fr_u64(1), fmb = fr_u64((uint64_t)mb);
    size_t d = Divide(c, 16, inLen, &fbs); W(dv) = W(d); W(rm) = W(d + 1);
    W(o + m) = fr_add(&W(dv), &one);
    AssertLessEqThan(c, 16, &W(o + m), &fmb);
    W(flt) = one;
    for (size_t i = 0; i < m; i++) {
        size_t e = IsEqualK(c, i, inLen); W(isEq + i) = W(e);
        fr ne = fr_sub(&one, &W(isEq + i)); W(flt + i + 1) = fr_mul(&W(flt + i), &ne);
    }
    fr last = fr_mul(&W(o + m), &fbs); last = fr_sub(&last, &one);
    for (size_t i = 0; i < m; i++) {
        size_t e = IsEqualK(c, i, &last); W(isLast + i) = W(e);
        fr t = fr_mul(&in[i], &W(flt + i + 1)), u = fr_mul_u64(&W(isLast + i), 0x80);
        t = fr_add(&t, &W(isEq + i)); W(o + i) = fr_add(&t, &u);
    }
    return o;
}
/* KeccakBytes(maxBlocks) keccak.circom:454-489
 * [out[32] | in[m], inLen | padded[m], numBlocks, inBitsArray[m][8], inBits[8m], inBlocks[mb][17][64], outBits[256], outBytes[32][8]]
 * || AssertLessThan(16), Pad, Num2Bits(8) x m, Flatten(m,8), Keccak(mb), Reshape(32,8), Bits2Num(8) x 32 */
static size_t KeccakBytes(ctx *c, int mb, const fr *in, const fr *inLen) {
    size_t m = 136 * (size_t)mb;
    size_t o = A(c, 32 + m + 1 + m + 1 + 8 * m + 8 * m + 1088 * (size_t)mb + 256 + 256);
    size_t I = o + 32, IL = I + m, pad = IL + 1, nbk = pad + m, bitsA = nbk + 1, bits = bitsA + 8 * m, blk = bits + 8 * m,
           ob = blk + 1088 * (size_t)mb, oby = ob + 256;
    cp(c, I, in, m); W(IL) = *inLen;
    fr fm = fr_u64(m);
    AssertLessThan(c, 16, inLen, &fm);
    size_t p = Pad(c, mb, 136, in, inLen); cp(c, pad, &W(p), m); W(nbk) = W(p + m);
    for (size_t i = 0; i < m; i++) { size_t nb = Num2Bits(c, 8, &W(pad + i)); cp(c, bitsA + 8 * i, &W(nb), 8); }
    size_t f = Flatten(c, (int)m, 8, &W(bitsA)); cp(c, bits, &W(f), 8 * m);
    cp(c, blk, &W(bits), 8 * m);
    size_t k = Keccak(c, mb, &W(blk), &W(nbk)); cp(c, ob, &W(k), 256);
    size_t r = Reshape(c, 32, 8, &W(ob)); cp(c, oby, &W(r), 256);
    for (int i = 0; i < 32; i++) { size_t b = Bits2Num(c, 8, &W(oby + 8 * i)); W(o + i) = W(b); }
    return o;
}

/* =====================================================================================
 * circuits/utils/substring_check.circom:24-100
 * [out | mainInput[mm], mainLen, subInput[sl] | subInputNum, M[mm+1], exists[k], isLastIndex[k], allowed[k+1], sums[k+1], doesNotExist]
 * || AssertByteString(sl), AssertByteString(mm), AssertLessEqThan(16) x2, LittleEndianBytes2Num(sl), {IsEqual, IsEqual} x k, IsZero
 * ===================================================================================== */
static size_t SubstringCheck(ctx *c, int mm, int sl, const fr *mainInput, const fr *mainLen, const fr *subInput) {
    size_t k = (size_t)(mm - sl + 1);
    size_t o = A(c, 1 + (size_t)mm + 1 + sl + 1 + (mm + 1) + k + k + (k + 1) + (k + 1) + 1);
    size_t MI = o + 1, ML = MI + mm, SI = ML + 1, num = SI + sl, Mm = num + 1, ex = Mm + mm + 1, isl = ex + k, alw = isl + k,
           sums = alw + k + 1, dne = sums + k + 1;
    cp(c, MI, mainInput, (size_t)mm); W(ML) = *mainLen; cp(c, SI, subInput, (size_t)sl);
    AssertByteString(c, sl, subInput);
    AssertByteString(c, mm, mainInput);
    fr fmm = fr_u64((uint64_t)mm), fsl = fr_u64((uint64_t)sl), one = fr_u64(1);
    AssertLessEqThan(c, 16, mainLen, &fmm);
    AssertLessEqThan(c, 16, &fsl, mainLen);
    size_t l = LittleEndianBytes2Num(c, sl, subInput); W(num) = W(l);
    fr pw = one, f256 = fr_u64(256);
    fr *pows = (fr *)malloc(sizeof(fr) * (size_t)(mm + 1));
    for (int i = 0; i < mm; i++) {                     /* M[i+1] <== mainInput[i]*256^i + M[i]  (:45-49) */
        pows[i] = pw;
        fr t = fr_mul(&mainInput[i], &pw); W(Mm + i + 1) = fr_add(&t, &W(Mm + i));
        pw = fr_mul(&pw, &f256);
    }
    W(alw) = one;
    fr lastIdx = fr_sub(mainLen, &fsl); lastIdx = fr_add(&lastIdx, &one);
    for (size_t i = 0; i < k; i++) {
        size_t e = IsEqualK(c, i, &lastIdx); W(isl + i) = W(e);                /* :87 */
        fr ne = fr_sub(&one, &W(isl + i)); W(alw + i + 1) = fr_mul(&W(alw + i), &ne);
        fr lhs = fr_mul(&W(num), &pows[i]), rhs = fr_sub(&W(Mm + i + sl), &W(Mm + i));
        e = IsEqual(c, &lhs, &rhs); W(ex + i) = W(e);                          /* :91 */
        fr t = fr_mul(&W(alw + i + 1), &W(ex + i)); W(sums + i + 1) = fr_add(&W(sums + i), &t);
    }
    free(pows);
    size_t z = IsZero(c, &W(sums + k)); W(dne) = W(z);
    W(o) = fr_sub(&one, &W(dne));
    return o;
}

/* =====================================================================================
 * circuits/utils/rlp/integer.circom
 * ===================================================================================== */
/* CountBytes(N) integer.circom:16-49   [len | bytes[N] | isZero[N], stillZero[N]] || IsZero x N */
static size_t CountBytes(ctx *c, int N, const fr *bytes) {
    size_t o = A(c, 1 + 3 * (size_t)N), iz = o + 1 + N, sz = iz + N; cp(c, o + 1, bytes, (size_t)N);
    for (int i = 0; i < N; i++) { size_t z = IsZero(c, &bytes[i]); W(iz + i) = W(z); }
    fr lead = fr_u64(0);
    for (int i = 0; i < N; i++) {
        W(sz + i) = i == 0 ? W(iz) : fr_mul(&W(iz + i), &W(sz + i - 1));
        lead = fr_add(&lead, &W(sz + i));
    }
    fr fn = fr_u64((uint64_t)N); W(o) = fr_sub(&fn, &lead);
    return o;
}
/* RlpInteger(N) integer.circom:67-110
 * [out[N+1], outLen | in | bytes[N], length, bigEndian[N], isSingleByte, isZero, firstRlpByte]
 * || Num2BigEndianBytes(N), CountBytes(N), ShiftLeft(N), LessThan(8N), IsZero, Mux1 */
static size_t RlpInteger(ctx *c, int N, const fr *in) {
    size_t o = A(c, (size_t)N + 1 + 1 + 1 + N + 1 + N + 3);
    size_t OL = o + N + 1, IN = OL + 1, by = IN + 1, len = by + N, be = len + 1, isb = be + N, isz = isb + 1, frb = isz + 1;
    W(IN) = *in;
    size_t x = Num2BigEndianBytes(c, N, in); cp(c, by, &W(x), (size_t)N);
    x = CountBytes(c, N, &W(by)); W(len) = W(x);
    fr fn = fr_u64((uint64_t)N), sh = fr_sub(&fn, &W(len));
    x = ShiftLeft(c, N, &W(by), &sh); cp(c, be, &W(x), (size_t)N);
    fr f128 = fr_u64(128), one = fr_u64(1);
    x = LessThan(c, 8 * N, in, &f128); W(isb) = W(x);
    x = IsZero(c, in); W(isz) = W(x);
    fr c0 = fr_add(&f128, &W(len));
    x = Mux1(c, &c0, in, &W(isb)); W(frb) = W(x);
    fr t = fr_mul_u64(&W(isz), 0x80); W(o) = fr_add(&W(frb), &t);
    fr nsb = fr_sub(&one, &W(isb));
    for (int i = 1; i < N + 1; i++) W(o + i) = fr_mul(&nsb, &W(be + i - 1));
    t = fr_add(&nsb, &W(len)); W(OL) = fr_add(&t, &W(isz));
    return o;
}
/* RlpEmptyAccount(maxBalanceBytes) empty_account.circom:20-134
 * [out[70+mb], outLen | balance | prefixedNonceAndBalanceRlp[4+mb], prefixedNonceAndBalanceRlpLen, balanceRlp[mb+1], balanceRlpLen,
 *  nonceAndBalanceRlpLen, storageAndCodeHashRlp[66]] || RlpInteger(mb), Concat(4+mb, 66) */
static const uint8_t EMPTY_STORAGE_CODE_RLP[66] = { /* 0xa0 | keccak(rlp("")) | 0xa0 | keccak("")  (empty_account.circom:9-10,55-120) */
    0xa0, 0x56, 0xe8, 0x1f, 0x17, 0x1b, 0xcc, 0x55, 0xa6, 0xff, 0x83, 0x45, 0xe6, 0x92, 0xc0, 0xf8, 0x6e, 0x5b, 0x48, 0xe0, 0x1b, 0x99,
    0x6c, 0xad, 0xc0, 0x01, 0x62, 0x2f, 0xb5, 0xe3, 0x63, 0xb4, 0x21,
    0xa0, 0xc5, 0xd2, 0x46, 0x01, 0x86, 0xf7, 0x23, 0x3c, 0x92, 0x7e, 0x7d, 0xb2, 0xdc, 0xc7, 0x03, 0xc0, 0xe5, 0x00, 0xb6, 0x53, 0xca,
    0x82, 0x27, 0x3b, 0x7b, 0xfa, 0xd8, 0x04, 0x5d, 0x85, 0xa4, 0x70};
static size_t RlpEmptyAccount(ctx *c, int mb, const fr *balance) {
    size_t o = A(c, (size_t)(70 + mb) + 1 + 1 + (4 + mb) + 1 + (mb + 1) + 1 + 1 + 66);
    size_t OL = o + 70 + mb, IN = OL + 1, pn = IN + 1, pnl = pn + 4 + mb, br = pnl + 1, brl = br + mb + 1, nbl = brl + 1, sc = nbl + 1;
    W(IN) = *balance;
    W(pn + 2) = fr_u64(0x80);
    size_t x = RlpInteger(c, mb, balance); cp(c, br, &W(x), (size_t)mb + 1); W(brl) = W(x + mb + 1);
    for (int i = 0; i < mb + 1; i++) W(pn + 3 + i) = W(br + i);
    fr one = fr_u64(1), two = fr_u64(2);
    W(nbl) = fr_add(&one, &W(brl));
    W(pnl) = fr_add(&two, &W(nbl));
    for (int i = 0; i < 66; i++) W(sc + i) = fr_u64(EMPTY_STORAGE_CODE_RLP[i]);
    W(pn) = fr_u64(0xf8);
    fr f66 = fr_u64(66); W(pn + 1) = fr_add(&W(nbl), &f66);
    x = Concat(c, 4 + mb, 66, &W(pn), &W(pnl), &W(sc), &f66);
    cp(c, o, &W(x), (size_t)(70 + mb)); W(OL) = W(x + 70 + mb);
    return o;
}

/* =====================================================================================
 * circuits/utils/rlp/merkle_patricia_trie_leaf.circom
 * ===================================================================================== */
/* TruncatedAddressHash(b) :50-90
 * [out[b+1], outLen | addressHashNibbles[2b], addressHashNibblesLen | div, rem, shifted[2b], outNibbles[2b+2], temp[2b-1]]
 * || AssertLessEqThan(7), Divide(7), ShiftLeft(2b), Mux1 x (2b-1), Nibbles2Bytes(b+1).  temp[] (:76) is never assigned: stays 0. */
static size_t TruncatedAddressHash(ctx *c, int b, const fr *nib, const fr *len) {
    int n2 = 2 * b;
    size_t o = A(c, (size_t)(b + 1) + 1 + n2 + 1 + 2 + n2 + (n2 + 2) + (n2 - 1));
    size_t OL = o + b + 1, IN = OL + 1, ILN = IN + n2, dv = ILN + 1, rm = dv + 1, shf = rm + 1, on = shf + n2;
    cp(c, IN, nib, (size_t)n2); W(ILN) = *len;
    fr fn2 = fr_u64((uint64_t)n2), two = fr_u64(2), one = fr_u64(1);
    AssertLessEqThan(c, 7, len, &fn2);
    size_t x = Divide(c, 7, len, &two); W(dv) = W(x); W(rm) = W(x + 1);
    fr cnt = fr_sub(&fn2, len);
    x = ShiftLeft(c, n2, nib, &cnt); cp(c, shf, &W(x), (size_t)n2);
    W(on) = fr_add(&two, &W(rm));
    W(on + 1) = fr_mul(&W(rm), &W(shf));
    for (int i = 0; i < n2; i++) {
        if (i < n2 - 1) { x = Mux1(c, &W(shf + i), &W(shf + i + 1), &W(rm)); W(on + i + 2) = W(x); }
        else { fr nr = fr_sub(&one, &W(rm)); W(on + i + 2) = fr_mul(&nr, &W(shf + i)); }
    }
    x = Nibbles2Bytes(c, b + 1, &W(on)); cp(c, o, &W(x), (size_t)b + 1);
    W(OL) = fr_add(&one, &W(dv));
    return o;
}
/* RlpMerklePatriciaTrieLeaf(ab, bb) :102-189 */
static size_t RlpMerklePatriciaTrieLeaf(ctx *c, int ab, int bb, const fr *nib, const fr *nibLen, const fr *balance) {
    int maxAcc = 4 + bb + 66, maxVal = 2 + maxAcc, maxKey = 1 + ab, maxKeyRlp = 1 + maxKey, maxPK = 2 + maxKeyRlp, maxOut = maxPK + maxVal;
    size_t o = A(c, (size_t)maxOut + 1 + 2 * ab + 1 + 1 + maxKey + 1 + maxAcc + 1 + maxPK + 1 + maxVal + 1);
    size_t OL = o + maxOut, IN = OL + 1, INL = IN + 2 * ab, IB = INL + 1, key = IB + 1, keyLen = key + maxKey, acc = keyLen + 1,
           accLen = acc + maxAcc, pk = accLen + 1, pkLen = pk + maxPK, val = pkLen + 1, valLen = val + maxVal;
    cp(c, IN, nib, 2 * (size_t)ab); W(INL) = *nibLen; W(IB) = *balance;
    size_t x = TruncatedAddressHash(c, ab, nib, nibLen); cp(c, key, &W(x), (size_t)maxKey); W(keyLen) = W(x + maxKey);
    fr two = fr_u64(2), one = fr_u64(1), three = fr_u64(3);
    AssertGreaterEqThan(c, 16, &W(keyLen), &two);
    x = RlpEmptyAccount(c, bb, balance); cp(c, acc, &W(x), (size_t)maxAcc); W(accLen) = W(x + maxAcc);
    W(val) = fr_u64(0xb8); W(val + 1) = W(accLen);
    for (int i = 0; i < maxAcc; i++) W(val + 2 + i) = W(acc + i);
    W(valLen) = fr_add(&two, &W(accLen));
    W(pk) = fr_u64(0xf8);
    fr t = fr_add(&W(keyLen), &one); W(pk + 1) = fr_add(&t, &W(valLen));
    fr f80 = fr_u64(0x80); W(pk + 2) = fr_add(&f80, &W(keyLen));
    for (int i = 0; i < maxKey; i++) W(pk + 3 + i) = W(key + i);
    W(pkLen) = fr_add(&three, &W(keyLen));
    x = Concat(c, maxPK, maxVal, &W(pk), &W(pkLen), &W(val), &W(valLen));
    cp(c, o, &W(x), (size_t)maxOut); W(OL) = W(x + maxOut);
    return o;
}
/* IsInRange(B) :196-207   [out | lower, value, upper | lowerLteValue, valueLteUpper] || AssertBits x3, LessEqThan x2 */
static size_t IsInRange(ctx *c, int B, const fr *lo, const fr *v, const fr *hi) {
    size_t o = A(c, 6); W(o + 1) = *lo; W(o + 2) = *v; W(o + 3) = *hi;
    AssertBits(c, B, lo); AssertBits(c, B, v); AssertBits(c, B, hi);
    size_t x = LessEqThan(c, B, lo, v); W(o + 4) = W(x);
    x = LessEqThan(c, B, v, hi); W(o + 5) = W(x);
    W(o) = fr_mul(&W(o + 4), &W(o + 5));
    return o;
}
/* LeafDetector(N) :247-294 */
static size_t LeafDetector(ctx *c, int N, const fr *layer, const fr *layerLen) {
    size_t o = A(c, 1 + (size_t)N + 1 + 16), L = o + 1, LL = L + N, m = LL + 1;
    enum { leafPrefixIsF8, totalLength, isConsistentWithLayerLen, keyPrefix, keyPrefixIsValid, keyIsMultiByte, keyExtraLen, keyLen,
           valueWrapperPrefix, valueWrapperPrefixIsB8, valueWrapperLen, valuePrefix, valuePrefixIsF8, valueLen,
           isValueWrapperLenConsistent, isKeyValueLenEqualWithLayerLen };
    cp(c, L, layer, (size_t)N); W(LL) = *layerLen;
    fr fN = fr_u64((uint64_t)N), fF8 = fr_u64(0xf8), fB8 = fr_u64(0xb8), fB7 = fr_u64(0xb7), f81 = fr_u64(0x81), f80 = fr_u64(0x80);
    fr one = fr_u64(1), two = fr_u64(2), three = fr_u64(3), six = fr_u64(6);
    AssertLessEqThan(c, 16, layerLen, &fN);
    size_t x = IsEqual(c, &layer[0], &fF8); W(m + leafPrefixIsF8) = W(x);
    W(m + totalLength) = layer[1];
    fr t = fr_add(&W(m + totalLength), &two);
    x = IsEqual(c, &t, layerLen); W(m + isConsistentWithLayerLen) = W(x);
    W(m + keyPrefix) = layer[2];
    x = LessEqThan(c, 16, &W(m + keyPrefix), &fB7); W(m + keyPrefixIsValid) = W(x);
    x = IsInRange(c, 16, &f81, &W(m + keyPrefix), &fB7); W(m + keyIsMultiByte) = W(x);
    t = fr_sub(&W(m + keyPrefix), &f80); W(m + keyExtraLen) = fr_mul(&W(m + keyIsMultiByte), &t);
    W(m + keyLen) = fr_add(&one, &W(m + keyExtraLen));
    fr base = fr_add(&two, &W(m + keyLen));
    x = Selector(c, N, layer, &base); W(m + valueWrapperPrefix) = W(x);
    x = IsEqual(c, &W(m + valueWrapperPrefix), &fB8); W(m + valueWrapperPrefixIsB8) = W(x);
    t = fr_add(&base, &one); x = Selector(c, N, layer, &t); W(m + valueWrapperLen) = W(x);
    t = fr_add(&base, &two); x = Selector(c, N, layer, &t); W(m + valuePrefix) = W(x);
    x = IsEqual(c, &W(m + valuePrefix), &fF8); W(m + valuePrefixIsF8) = W(x);
    t = fr_add(&base, &three); x = Selector(c, N, layer, &t); W(m + valueLen) = W(x);
    t = fr_add(&W(m + valueLen), &two);
    x = IsEqual(c, &W(m + valueWrapperLen), &t); W(m + isValueWrapperLenConsistent) = W(x);
    t = fr_add(&W(m + keyLen), &W(m + valueLen)); t = fr_add(&t, &six);
    x = IsEqual(c, &t, layerLen); W(m + isKeyValueLenEqualWithLayerLen) = W(x);
    fr ins[7] = {W(m + leafPrefixIsF8), W(m + isConsistentWithLayerLen), W(m + keyPrefixIsValid), W(m + valueWrapperPrefixIsB8),
                 W(m + isValueWrapperLenConsistent), W(m + valuePrefixIsF8), W(m + isKeyValueLenEqualWithLayerLen)};
    x = MultiAND(c, 7, ins); W(o) = W(x);
    return o;
}

/* =====================================================================================
 * circuits/utils/constants.circom:3-14, burn_address.circom, proof_of_work.circom, public_commitment.circom
 * ===================================================================================== */
static fr POSEIDON_PREFIX(uint64_t add) {
    /* keccak("EIP-7503") mod p = 5265656504298861414514317065875120428884240036965045859626767452974705356670 */
    fr r = {{0xf0363f983d892f7eULL, 0xd115b780980a6b46ULL, 0x007d2482cd46cec2ULL, 0x0ba44186ee7876b8ULL}};
    fr a = fr_u64(add); return fr_add(&r, &a);
}
/* BurnAddress burn_address.circom:47-58   [addressBytes[20] | burnKey, revealAmount, burnExtraCommitment | hash, hashBytes[32]] */
static size_t BurnAddress(ctx *c, const fr *bk, const fr *ra, const fr *bec) {
    size_t o = A(c, 20 + 3 + 1 + 32); W(o + 20) = *bk; W(o + 21) = *ra; W(o + 22) = *bec;
    fr in[4] = {POSEIDON_PREFIX(0), *bk, *ra, *bec};
    size_t x = Poseidon(c, 4, in); W(o + 23) = W(x);
    x = Num2BigEndianBytes(c, 32, &W(o + 23)); cp(c, o + 24, &W(x), 32);
    x = Fit(c, 32, 20, &W(o + 24)); cp(c, o, &W(x), 20);
    return o;
}
/* BurnAddressHash burn_address.circom:67-83   [addressHashNibbles[64] | 3 inputs | addressBytes[20], addressBytesBlock[136], addressHash[32]] */
static size_t BurnAddressHash(ctx *c, const fr *bk, const fr *ra, const fr *bec) {
    size_t o = A(c, 64 + 3 + 20 + 136 + 32); W(o + 64) = *bk; W(o + 65) = *ra; W(o + 66) = *bec;
    size_t ab = o + 67, blk = ab + 20, ah = blk + 136;
    size_t x = BurnAddress(c, bk, ra, bec); cp(c, ab, &W(x), 20);
    x = Fit(c, 20, 136, &W(ab)); cp(c, blk, &W(x), 136);
    fr f20 = fr_u64(20);
    x = KeccakBytes(c, 1, &W(blk), &f20); cp(c, ah, &W(x), 32);
    x = Bytes2Nibbles(c, 32, &W(ah)); cp(c, o, &W(x), 64);
    return o;
}
/* EIP7503 proof_of_work.circom:11-21 */
static size_t EIP7503(ctx *c) {
    size_t o = A(c, 8); const char *s = "EIP-7503";
    for (int i = 0; i < 8; i++) W(o + i) = fr_u64((uint8_t)s[i]);
    return o;
}
/* ConcatFixed4 proof_of_work.circom:28-48   [out | a,b,c,d] */
static size_t ConcatFixed4(ctx *c, int a, int b, int cc, int d, const fr *pa, const fr *pb, const fr *pc, const fr *pd) {
    size_t n = (size_t)(a + b + cc + d), o = A(c, 2 * n);
    cp(c, o + n, pa, (size_t)a); cp(c, o + n + a, pb, (size_t)b); cp(c, o + n + a + b, pc, (size_t)cc); cp(c, o + n + a + b + cc, pd, (size_t)d);
    cp(c, o, &W(o + n), n);
    return o;
}
/* ProofOfWorkChecker proof_of_work.circom:54-81 */
static size_t ProofOfWorkChecker(ctx *c, const fr *bk, const fr *ra, const fr *bec, const fr *mzb) {
    size_t o = A(c, 4 + 32 * 3 + 8 + 104 + 136 + 32 + 32);
    size_t kb = o + 4, rb = kb + 32, eb = rb + 32, eip = eb + 32, hin = eip + 8, blk = hin + 104, kk = blk + 136, sbz = kk + 32;
    W(o) = *bk; W(o + 1) = *ra; W(o + 2) = *bec; W(o + 3) = *mzb;
    size_t x = Num2BigEndianBytes(c, 32, bk); cp(c, kb, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, ra); cp(c, rb, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, bec); cp(c, eb, &W(x), 32);
    x = EIP7503(c); cp(c, eip, &W(x), 8);
    x = ConcatFixed4(c, 32, 32, 32, 8, &W(kb), &W(rb), &W(eb), &W(eip)); cp(c, hin, &W(x), 104);
    x = Fit(c, 104, 136, &W(hin)); cp(c, blk, &W(x), 136);
    fr f104 = fr_u64(104);
    x = KeccakBytes(c, 1, &W(blk), &f104); cp(c, kk, &W(x), 32);
    x = Filter(c, 32, mzb); cp(c, sbz, &W(x), 32);
    for (int i = 0; i < 32; i++) { fr p = fr_mul(&W(kk + i), &W(sbz + i)); REQUIRE(fr_is_zero(&p), "ProofOfWorkChecker", 79); }
    return o;
}
/* PublicCommitment(N) public_commitment.circom:18-42   [out | in[N][32] | flattenIn[32N], block[136nb], hash[32], reducedHash[31]] */
static size_t PublicCommitment(ctx *c, int N, const fr *in) {
    int nb = N * 32 / 136 + ((N * 32) % 136 != 0);
    size_t n32 = 32 * (size_t)N, bl = 136 * (size_t)nb;
    size_t o = A(c, 1 + n32 + n32 + bl + 32 + 31), fl = o + 1 + n32, blk = fl + n32, h = blk + bl, rh = h + 32;
    cp(c, o + 1, in, n32);
    for (int i = 0; i < N; i++) AssertByteString(c, 32, &in[32 * i]);
    size_t x = Flatten(c, N, 32, in); cp(c, fl, &W(x), n32);
    x = Fit(c, (int)n32, (int)bl, &W(fl)); cp(c, blk, &W(x), bl);
    fr fl32 = fr_u64(n32);
    x = KeccakBytes(c, nb, &W(blk), &fl32); cp(c, h, &W(x), 32);
    x = Fit(c, 32, 31, &W(h)); cp(c, rh, &W(x), 31);
    x = BigEndianBytes2Num(c, 31, &W(rh)); W(o) = W(x);
    return o;
}

/* =====================================================================================
 * circuits/spend.circom:32-53
 * [commitment | burnKey, balance, withdrawnBalance, extraCommitment | coin, remainingCoin, coinBytes[32], withdrawnBalanceBytes[32],
 *  remainingCoinBytes[32], extraCommmitmentBytes[32]] || AssertGreaterEqThan(8*mab), Poseidon(3) x2, Num2BigEndianBytes(32) x4, PublicCommitment(4)
 * ===================================================================================== */
static size_t Spend(ctx *c, int mab, const fr *bk, const fr *bal, const fr *wd, const fr *ec) {
    size_t o = A(c, 1 + 4 + 2 + 128), coin = o + 5, rc = o + 6, cb = o + 7;
    W(o + 1) = *bk; W(o + 2) = *bal; W(o + 3) = *wd; W(o + 4) = *ec;
    AssertGreaterEqThan(c, 8 * mab, bal, wd);
    fr in[3] = {POSEIDON_PREFIX(2), *bk, *bal};
    size_t x = Poseidon(c, 3, in); W(coin) = W(x);
    in[2] = fr_sub(bal, wd);
    x = Poseidon(c, 3, in); W(rc) = W(x);
    x = Num2BigEndianBytes(c, 32, &W(coin)); cp(c, cb, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, wd); cp(c, cb + 32, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, &W(rc)); cp(c, cb + 64, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, ec); cp(c, cb + 96, &W(x), 32);
    x = PublicCommitment(c, 4, &W(cb)); W(o) = W(x);
    return o;
}

/* =====================================================================================
 * circuits/proof_of_burn.circom:34-212
 * ===================================================================================== */
typedef struct { int L, NB, HB, minNib, amountBytes, powZero; fr maxIntended, maxActual; } pob_params;
static size_t ProofOfBurn(ctx *c, const pob_params *P, const fr *in) {
    int L = P->L, LB = 136 * P->NB, HBy = 136 * P->HB, AB = P->amountBytes;
    size_t nin = 6 + (size_t)L * LB + L + 1 + HBy + 3;
    size_t nmid = 2 + 64 + 32 * 7 + LB + 1 + L + (L - 1) + 32 * (size_t)L + 31 * (size_t)L + L + 1 + 139 + 1;
    size_t o = A(c, 1 + nin + nmid);
    /* inputs, declaration order proof_of_burn.circom:43-72 */
    size_t burnKey = o + 1, actualBalance = o + 2, intendedBalance = o + 3, revealAmount = o + 4, burnExtraCommitment = o + 5,
           numLeafAddressNibbles = o + 6, layers = o + 7, layerLens = layers + (size_t)L * LB, numLayers = layerLens + L,
           blockHeader = numLayers + 1, blockHeaderLen = blockHeader + HBy, byteSecurityRelax = blockHeaderLen + 1,
           proofExtraCommitment = byteSecurityRelax + 1;
    cp(c, o + 1, in, nin);
    /* intermediates, declaration order :113-200 */
    size_t remainingCoin = proofExtraCommitment + 1, nullifier = remainingCoin + 1, addressHashNibbles = nullifier + 1,
           blockRoot = addressHashNibbles + 64, stateRoot = blockRoot + 32, nullifierBytes = stateRoot + 32,
           remainingCoinBytes = nullifierBytes + 32, revealAmountBytes = remainingCoinBytes + 32,
           burnExtraCommitmentBytes = revealAmountBytes + 32, extraCommitmentBytes = burnExtraCommitmentBytes + 32,
           lastLayer = extraCommitmentBytes + 32, lastLayerLen = lastLayer + LB, layerExists = lastLayerLen + 1,
           substringCheckers = layerExists + L, layerKeccaks = substringCheckers + (L - 1), reducedLayerKeccaks = layerKeccaks + 32 * (size_t)L,
           isLeaf = reducedLayerKeccaks + 31 * (size_t)L, isLastLayerLeaf = isLeaf + L, leaf = isLastLayerLeaf + 1, leafLen = leaf + 139;
    fr one = fr_u64(1), two = fr_u64(2);

    AssertLessEqThan(c, AB * 8, &W(intendedBalance), &P->maxIntended);                 /* :84 */
    AssertLessEqThan(c, AB * 8, &W(actualBalance), &P->maxActual);                     /* :85 */
    AssertLessEqThan(c, AB * 8, &W(intendedBalance), &W(actualBalance));               /* :86 */
    fr relax2 = fr_mul(&W(byteSecurityRelax), &two), fmin = fr_u64((uint64_t)P->minNib);
    AssertLessEqThan(c, 16, &relax2, &fmin);                                           /* :90 */
    fr lim = fr_sub(&fmin, &relax2);
    AssertGreaterEqThan(c, 16, &W(numLeafAddressNibbles), &lim);                       /* :91 */
    AssertBits(c, AB * 8, &W(revealAmount));                                           /* :96 */
    AssertLessEqThan(c, AB * 8, &W(revealAmount), &W(intendedBalance));                /* :97 */
    fr maxLayerBits = fr_u64((uint64_t)LB * 8), maxHdrBits = fr_u64((uint64_t)HBy * 8);
    for (int i = 0; i < L; i++) {                                                      /* :99-103 */
        AssertLessThan(c, 16, &W(layerLens + i), &maxLayerBits);
        AssertByteString(c, LB, &W(layers + (size_t)i * LB));
    }
    AssertLessThan(c, 16, &W(blockHeaderLen), &maxHdrBits);                            /* :105 */
    AssertByteString(c, HBy, &W(blockHeader));                                         /* :106 */

    fr pin[3] = {POSEIDON_PREFIX(2), W(burnKey), fr_sub(&W(intendedBalance), &W(revealAmount))};
    size_t x = Poseidon(c, 3, pin); W(remainingCoin) = W(x);                           /* :113 */
    fr pin2[2] = {POSEIDON_PREFIX(1), W(burnKey)};
    x = Poseidon(c, 2, pin2); W(nullifier) = W(x);                                     /* :116 */
    x = BurnAddressHash(c, &W(burnKey), &W(revealAmount), &W(burnExtraCommitment)); cp(c, addressHashNibbles, &W(x), 64); /* :119 */
    x = KeccakBytes(c, P->HB, &W(blockHeader), &W(blockHeaderLen)); cp(c, blockRoot, &W(x), 32);                          /* :122 */
    for (int i = 0; i < 32; i++) W(stateRoot + i) = W(blockHeader + 91 + i);          /* :125-129 */
    x = Num2BigEndianBytes(c, 32, &W(nullifier)); cp(c, nullifierBytes, &W(x), 32);    /* :132-136 */
    x = Num2BigEndianBytes(c, 32, &W(remainingCoin)); cp(c, remainingCoinBytes, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, &W(revealAmount)); cp(c, revealAmountBytes, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, &W(burnExtraCommitment)); cp(c, burnExtraCommitmentBytes, &W(x), 32);
    x = Num2BigEndianBytes(c, 32, &W(proofExtraCommitment)); cp(c, extraCommitmentBytes, &W(x), 32);
    {
        fr *pc = (fr *)malloc(sizeof(fr) * 192);
        memcpy(pc, &W(blockRoot), 32 * sizeof(fr)); memcpy(pc + 32, &W(nullifierBytes), 160 * sizeof(fr));
        x = PublicCommitment(c, 6, pc); W(o) = W(x);                                   /* :137-139 */
        free(pc);
    }
    fr nl1 = fr_sub(&W(numLayers), &one);
    x = SelectorArray1D(c, L, LB, &W(layers), &nl1); cp(c, lastLayer, &W(x), (size_t)LB);  /* :142-143 */
    x = Selector(c, L, &W(layerLens), &nl1); W(lastLayerLen) = W(x);                   /* :146 */
    x = Filter(c, L, &W(numLayers)); cp(c, layerExists, &W(x), (size_t)L);             /* :150 */
    fr numDetected = fr_u64(0);
    for (int i = 0; i < L; i++) {                                                      /* :157-181 */
        x = LeafDetector(c, LB, &W(layers + (size_t)i * LB), &W(layerLens + i)); W(isLeaf + i) = W(x);
        numDetected = fr_add(&numDetected, &W(isLeaf + i));
        x = KeccakBytes(c, P->NB, &W(layers + (size_t)i * LB), &W(layerLens + i)); cp(c, layerKeccaks + 32 * (size_t)i, &W(x), 32);
        x = Fit(c, 32, 31, &W(layerKeccaks + 32 * (size_t)i)); cp(c, reducedLayerKeccaks + 31 * (size_t)i, &W(x), 31);
        if (i > 0) {
            x = SubstringCheck(c, LB, 31, &W(layers + (size_t)(i - 1) * LB), &W(layerLens + i - 1), &W(reducedLayerKeccaks + 31 * (size_t)i));
            W(substringCheckers + i - 1) = W(x);
            fr t = fr_sub(&one, &W(substringCheckers + i - 1)); t = fr_mul(&t, &W(layerExists + i));
            REQUIRE(fr_is_zero(&t), "ProofOfBurn", 179);
        }
    }
    REQUIRE(fr_eq_u64(&numDetected, 1), "ProofOfBurn", 186);
    x = LeafDetector(c, LB, &W(lastLayer), &W(lastLayerLen)); W(isLastLayerLeaf) = W(x); /* :187 */
    REQUIRE(fr_eq_u64(&W(isLastLayerLeaf), 1), "ProofOfBurn", 188);
    for (int i = 0; i < 32; i++) REQUIRE(fr_eq(&W(layerKeccaks + i), &W(stateRoot + i)), "ProofOfBurn", 192);
    x = RlpMerklePatriciaTrieLeaf(c, 32, AB, &W(addressHashNibbles), &W(numLeafAddressNibbles), &W(actualBalance)); /* :198 */
    cp(c, leaf, &W(x), 139); W(leafLen) = W(x + 139);
    for (int i = 0; i < 139; i++) REQUIRE(fr_eq(&W(leaf + i), &W(lastLayer + i)), "ProofOfBurn", 204);
    REQUIRE(fr_eq(&W(leafLen), &W(lastLayerLen)), "ProofOfBurn", 206);
    fr mz = fr_u64((uint64_t)P->powZero); mz = fr_add(&mz, &W(byteSecurityRelax));
    ProofOfWorkChecker(c, &W(burnKey), &W(revealAmount), &W(burnExtraCommitment), &mz); /* :211 */
    return o;
}

/* =====================================================================================
 * Entry points (ctypes).  `inputs` = main's input signals, declaration order, flattened row-major,
 * each 4x64 LE limbs canonical.  Mirrors `./<circuit> input.json witness.wtns` (Makefile:4-5) minus JSON.
 * ===================================================================================== */
typedef struct {
    uint64_t *witness;   /* nwitness x 4 limbs; witness[0] = 1 */
    uint64_t nwitness, noutputs, ninputs_expected;
    int32_t failed, unknown;
    char msg[160];
} oracle_result;

static ctx *g_ctx;

static uint64_t pu(const uint64_t *params, int i) { return params[4 * i]; }
static fr pf(const uint64_t *params, int i) { fr r = {{params[4 * i], params[4 * i + 1], params[4 * i + 2], params[4 * i + 3]}}; return r; }

/* the failing sites of the last run ("Template:line;..."), execution order, each once */
const char *oracle_fail_sites(void) { return g_ctx ? g_ctx->sites : ""; }

void oracle_free(void) {
    if (g_ctx) { if (g_ctx->w) munmap(g_ctx->w, g_ctx->cap * sizeof(fr)); free(g_ctx); g_ctx = NULL; }
}

int oracle_run(const char *tpl, const uint64_t *params, int nparams, const uint64_t *inputs, uint64_t ninputs,
               uint64_t capacity, oracle_result *res) {
    oracle_free();
    ctx *c = (ctx *)calloc(1, sizeof(ctx)); g_ctx = c;
    c->cap = capacity ? capacity : ((size_t)1 << 22);
    c->w = (fr *)mmap(NULL, c->cap * sizeof(fr), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (c->w == MAP_FAILED) { c->w = NULL; return -1; }
    const char *e = getenv("ORACLE_DECL_ORDER"); c->decl_order = e && e[0] == '1';
    W(0) = fr_u64(1); c->n = 1;
    const fr *in = (const fr *)inputs;
    memset(res, 0, sizeof *res);
    size_t need = 0, nout = 0;
#define P(i) ((int)pu(params, i))
#define CASE(name, np) else if (!strcmp(tpl, name) && nparams == (np))
#define NEED(n_in, n_out) need = (n_in); nout = (n_out); if (ninputs != need) goto bad_inputs;
    if (0) {}
    CASE("ProofOfBurn", 8) {
        pob_params pp = {P(0), P(1), P(2), P(3), P(4), P(5), pf(params, 6), pf(params, 7)};
        NEED(6 + (size_t)pp.L * 136 * pp.NB + pp.L + 1 + 136 * (size_t)pp.HB + 3, 1) ProofOfBurn(c, &pp, in);
    }
    CASE("Spend", 1) { NEED(4, 1) Spend(c, P(0), &in[0], &in[1], &in[2], &in[3]); }
    CASE("EIP7503", 0) { NEED(0, 8) EIP7503(c); }
    CASE("ConcatFixed4", 4) { int a = P(0), b = P(1), cc = P(2), d = P(3); NEED((size_t)(a + b + cc + d), (size_t)(a + b + cc + d)) ConcatFixed4(c, a, b, cc, d, in, in + a, in + a + b, in + a + b + cc); }
    CASE("ProofOfWorkChecker", 0) { NEED(4, 0) ProofOfWorkChecker(c, &in[0], &in[1], &in[2], &in[3]); }
    CASE("PublicCommitment", 1) { NEED(32 * (size_t)P(0), 1) PublicCommitment(c, P(0), in); }
    CASE("Poseidon", 1) { NEED((size_t)P(0), 1) Poseidon(c, P(0), in); }
    CASE("Divide", 1) { NEED(2, 2) Divide(c, P(0), &in[0], &in[1]); }
    CASE("SubstringCheck", 2) { NEED((size_t)P(0) + 1 + P(1), 1) SubstringCheck(c, P(0), P(1), in, &in[P(0)], &in[P(0) + 1]); }
    CASE("ShiftLeft", 1) { NEED((size_t)P(0) + 1, (size_t)P(0)) ShiftLeft(c, P(0), in, &in[P(0)]); }
    CASE("ShiftRight", 2) { NEED((size_t)P(0) + 1, (size_t)P(0) + P(1)) ShiftRight(c, P(0), P(1), in, &in[P(0)]); }
    CASE("Mask", 1) { NEED((size_t)P(0) + 1, (size_t)P(0)) Mask(c, P(0), in, &in[P(0)]); }
    CASE("Concat", 2) { int a = P(0), b = P(1); NEED((size_t)a + b + 2, (size_t)a + b + 1) Concat(c, a, b, in, &in[a], &in[a + 1], &in[a + 1 + b]); }
    CASE("Selector", 1) { NEED((size_t)P(0) + 1, 1) Selector(c, P(0), in, &in[P(0)]); }
    CASE("SelectorArray1D", 2) { size_t np = (size_t)P(0) * P(1); NEED(np + 1, (size_t)P(1)) SelectorArray1D(c, P(0), P(1), in, &in[np]); }
    CASE("SelectorArray2D", 3) { size_t npq = (size_t)P(0) * P(1) * P(2); NEED(npq + 1, (size_t)P(1) * P(2)) SelectorArray2D(c, P(0), P(1), P(2), in, &in[npq]); }
    CASE("BigEndianBytes2Num", 1) { NEED((size_t)P(0), 1) BigEndianBytes2Num(c, P(0), in); }
    CASE("LittleEndianBytes2Num", 1) { NEED((size_t)P(0), 1) LittleEndianBytes2Num(c, P(0), in); }
    CASE("Bytes2Nibbles", 1) { NEED((size_t)P(0), 2 * (size_t)P(0)) Bytes2Nibbles(c, P(0), in); }
    CASE("Nibbles2Bytes", 1) { NEED(2 * (size_t)P(0), (size_t)P(0)) Nibbles2Bytes(c, P(0), in); }
    CASE("Num2BigEndianBytes", 1) { NEED(1, (size_t)P(0)) Num2BigEndianBytes(c, P(0), in); }
    CASE("Num2LittleEndianBytes", 1) { NEED(1, (size_t)P(0)) Num2LittleEndianBytes(c, P(0), in); }
    CASE("Num2BitsSafe", 1) { NEED(1, (size_t)P(0)) Num2BitsSafe(c, P(0), in); }
    CASE("Pad", 2) { size_t m = (size_t)P(0) * P(1); NEED(m + 1, m + 1) Pad(c, P(0), P(1), in, &in[m]); }
    CASE("KeccakBytes", 1) { size_t m = 136 * (size_t)P(0); NEED(m + 1, 32) KeccakBytes(c, P(0), in, &in[m]); }
    CASE("BurnAddress", 0) { NEED(3, 20) BurnAddress(c, &in[0], &in[1], &in[2]); }
    CASE("BurnAddressHash", 0) { NEED(3, 64) BurnAddressHash(c, &in[0], &in[1], &in[2]); }
    CASE("AssertBits", 1) { NEED(1, 0) AssertBits(c, P(0), in); }
    CASE("AssertByteString", 1) { NEED((size_t)P(0), 0) AssertByteString(c, P(0), in); }
    CASE("AssertLessEqThan", 1) { NEED(2, 0) AssertLessEqThan(c, P(0), &in[0], &in[1]); }
    CASE("AssertLessThan", 1) { NEED(2, 0) AssertLessThan(c, P(0), &in[0], &in[1]); }
    CASE("AssertGreaterEqThan", 1) { NEED(2, 0) AssertGreaterEqThan(c, P(0), &in[0], &in[1]); }
    CASE("Filter", 1) { NEED(1, (size_t)P(0)) Filter(c, P(0), in); }
    CASE("Fit", 2) { NEED((size_t)P(0), (size_t)P(1)) Fit(c, P(0), P(1), in); }
    CASE("Reverse", 1) { NEED((size_t)P(0), (size_t)P(0)) Reverse(c, P(0), in); }
    CASE("Flatten", 2) { size_t n = (size_t)P(0) * P(1); NEED(n, n) Flatten(c, P(0), P(1), in); }
    CASE("Reshape", 2) { size_t n = (size_t)P(0) * P(1); NEED(n, n) Reshape(c, P(0), P(1), in); }
    CASE("RlpInteger", 1) { NEED(1, (size_t)P(0) + 2) RlpInteger(c, P(0), in); }
    CASE("CountBytes", 1) { NEED((size_t)P(0), 1) CountBytes(c, P(0), in); }
    CASE("RlpEmptyAccount", 1) { NEED(1, (size_t)P(0) + 71) RlpEmptyAccount(c, P(0), in); }
    CASE("TruncatedAddressHash", 1) { NEED(2 * (size_t)P(0) + 1, (size_t)P(0) + 2) TruncatedAddressHash(c, P(0), in, &in[2 * P(0)]); }
    CASE("IsInRange", 1) { NEED(3, 1) IsInRange(c, P(0), &in[0], &in[1], &in[2]); }
    CASE("LeafDetector", 1) { NEED((size_t)P(0) + 1, 1) LeafDetector(c, P(0), in, &in[P(0)]); }
    CASE("RlpMerklePatriciaTrieLeaf", 2) { int ab = P(0), bb = P(1); NEED(2 * (size_t)ab + 2, (size_t)(ab + bb + 76) + 1) RlpMerklePatriciaTrieLeaf(c, ab, bb, in, &in[2 * ab], &in[2 * ab + 1]); }
    CASE("Keccakf", 0) { NEED(1600, 1600) Keccakf(c, in); }
    else { res->unknown = 1; snprintf(res->msg, sizeof res->msg, "unknown template %s/%d", tpl, nparams); return -2; }
    res->witness = (uint64_t *)c->w; res->nwitness = c->n; res->noutputs = nout; res->ninputs_expected = need;
    res->failed = c->failed; memcpy(res->msg, c->msg, sizeof res->msg);
    return 0;
bad_inputs:
    res->ninputs_expected = need; res->failed = 1;
    snprintf(res->msg, sizeof res->msg, "Not all inputs have been set. Only %llu out of %llu", (unsigned long long)ninputs, (unsigned long long)need);
    return -3;
}

/* .wtns writer -- restates the emitted runtime's writeBinWitness (patch point tests/test.py:36):
 * "wtns" | u32 2 | u32 2 | u32 1 | u64 40 | u32 32 | prime[32] | u32 nWitness | u32 2 | u64 32*nWitness | values LE */
static void put32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static void put64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
uint64_t oracle_wtns_size(void) { return g_ctx ? 12 + 12 + 40 + 12 + 32 * (uint64_t)g_ctx->n : 0; }
int oracle_wtns_into(uint8_t *dst, uint64_t cap) {
    if (!g_ctx || cap < oracle_wtns_size()) return -1;
    uint8_t *p = dst;
    memcpy(p, "wtns", 4); put32(p + 4, 2); put32(p + 8, 2); p += 12;
    put32(p, 1); put64(p + 4, 40); p += 12;
    put32(p, 32); memcpy(p + 4, &FR_P, 32); put32(p + 36, (uint32_t)g_ctx->n); p += 40;
    put32(p, 2); put64(p + 4, 32 * (uint64_t)g_ctx->n); p += 12;
    memcpy(p, g_ctx->w, 32 * g_ctx->n);
    return 0;
}
int oracle_wtns_write(const char *path) {
    if (!g_ctx) return -1;
    FILE *f = fopen(path, "wb"); if (!f) return -2;
    uint8_t hdr[76];
    memcpy(hdr, "wtns", 4); put32(hdr + 4, 2); put32(hdr + 8, 2);
    put32(hdr + 12, 1); put64(hdr + 16, 40);
    put32(hdr + 24, 32); memcpy(hdr + 28, &FR_P, 32); put32(hdr + 60, (uint32_t)g_ctx->n);
    put32(hdr + 64, 2); put64(hdr + 68, 32 * (uint64_t)g_ctx->n);
    fwrite(hdr, 1, 76, f); fwrite(g_ctx->w, 32, g_ctx->n, f);
    fclose(f);
    return 0;
}

/* standalone Keccak-256 on the oracle's own lane arithmetic, for fixture sanity checks */
static uint64_t rotl64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
void oracle_keccak_f(uint64_t a[25]) {
    static const int rot[25] = {1, 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; r++) {
        uint64_t cc[5], d[5], b[25];
        for (int x = 0; x < 5; x++) cc[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = cc[(x + 4) % 5] ^ rotl64(cc[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        b[0] = a[0];
        for (int i = 0; i < 24; i++) b[rot[i + 1]] = rotl64(a[rot[i]], ((i + 1) * (i + 2) / 2) % 64);
        for (int i = 0; i < 25; i++) { int y = i / 5 * 5; a[i] = b[i] ^ (~b[y + (i + 1) % 5] & b[y + (i + 2) % 5]); }
        a[0] ^= KECCAK_RC[r];
    }
}
void oracle_keccak256(const uint8_t *msg, uint64_t len, uint8_t out[32]) {
    uint64_t st[25]; memset(st, 0, sizeof st);
    uint8_t blk[136];
    uint64_t off = 0;
    for (;;) {
        uint64_t n = len - off < 136 ? len - off : 136;
        memset(blk, 0, 136); memcpy(blk, msg + off, n);
        int last = n < 136;
        if (last) { blk[n] ^= 0x01; blk[135] ^= 0x80; }
        for (int i = 0; i < 17; i++) { uint64_t v; memcpy(&v, blk + 8 * i, 8); st[i] ^= v; }
        oracle_keccak_f(st);
        off += n;
        if (last) break;
    }
    memcpy(out, st, 32);
}
