// Native input loader: input.json text -> the packed row of pob_upload_inputs.
//
// Replaces the calculator's emitted loadJson (reference: tests/test.py:57-59 writes input.json, tests/main.py:160-178 is the producer of
// the two circuits' files).  Same acceptance as the Python host's WitnessCalculator.pack (witness.py), which the tests hold it to
// bit for bit:
//   * the object's keys must be exactly the circuit's `signal input` names (proof_of_burn.circom:43-72, spend.circom:33-36);
//   * a scalar may come wrapped in one-element arrays (tests/testcases/divide.py:4), array inputs may be nested (layers[L][136 NB]) and are
//     flattened row-major; the element count must match the circuit's;
//   * a value is a JSON integer of any size and sign, true / false, or a string holding a decimal or 0x-hex integer; it is reduced mod p
//     (tests/testcases/convert.py:36 passes p - 1 as a string); fractions / exponents / null are refused;
//   * the byte-sized inputs are range-constrained in-circuit (AssertByteString, AssertBits(16) ...): a value that does not fit the int32
//     row (>= 2^31 after reduction, e.g. a negative number) marks the witness as failed up front (FAIL_INPUT_RANGE) and is stored as
//     0x7FFFFFFF (tests/testcases/rlp/integer.py:51-53, assertion.py:87 feed out-of-range values that must fail, not wrap).
// Host code only (no kernel in this translation unit); pob_pack_json_batch* spread the texts over the threads of a PERSISTENT pool (round 5: a
// batch of 1 024 texts is 0.5 ms of parsing on the cores of a GPU box -- starting and joining threads per call cost more than that) and write straight
// into the caller's (pinned) arrays.  Default width: the CPUs this process may run on (its affinity mask: a rank pinned to its GPU's NUMA node
// gets that node's cores) divided by LOCAL_WORLD_SIZE when a launcher set it, so that the ranks of a node do not each start a full-width loader.
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>
#include <new>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pob_hip.h"

namespace {
typedef unsigned __int128 u128;
const uint64_t P64[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
const uint32_t FAIL_INPUT_RANGE = (11u << 12) | 1u;          // witness.py FAIL_INPUT_RANGE

struct U256 { uint64_t l[4]; };
bool geq_p(const uint64_t* v5) {                              // v (5 limbs) >= p ?
    if (v5[4]) return true;
    for (int i = 3; i >= 0; i--) if (v5[i] != P64[i]) return v5[i] > P64[i];
    return true;
}
void sub_p(uint64_t* v5) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { const u128 d = (u128)v5[i] - P64[i] - br; v5[i] = (uint64_t)d; br = (d >> 127) & 1; }
    v5[4] -= (uint64_t)br;
}
// t (5 limbs, any value) <- t mod p: subtract p << k for k = 66 .. 0 (p > 2^253), then at most a few times p itself (rare path: literals of 78+ digits)
void reduce5(uint64_t* t) {
    for (int k = 66; k >= 0; k--) {
        uint64_t ps[6] = {0, 0, 0, 0, 0, 0};                   // p << k
        const int lo = k / 64, sh = k % 64;
        for (int i = 0; i < 4; i++) { if (i + lo < 5) ps[i + lo] |= P64[i] << sh; if (sh && i + lo + 1 < 5) ps[i + lo + 1] |= P64[i] >> (64 - sh); }
        if (lo == 1 && (P64[3] >> (64 - sh)) && sh) {}           // (bits shifted past limb 4 cannot occur: p << 66 < 2^320)
        bool ge = true;
        for (int i = 4; i >= 0; i--) if (t[i] != ps[i]) { ge = t[i] > ps[i]; break; }
        if (ge) { u128 br = 0; for (int i = 0; i < 5; i++) { const u128 d = (u128)t[i] - ps[i] - br; t[i] = (uint64_t)d; br = (d >> 127) & 1; } }
    }
    while (geq_p(t)) sub_p(t);
}
// t (5 limbs, t[4] == 0 on entry) <- t * mul + add, reduced mod p only when it leaves 256 bits (a 77-digit decimal literal -- every field element -- never does:
// round 4 reduced after every digit, 25 us per 77-digit value and 50 of the 57 us a production input.json took to load)
void muladd5(uint64_t* t, uint64_t mul, uint64_t add) {
    u128 c = add;
    for (int i = 0; i < 4; i++) { c += (u128)t[i] * mul; t[i] = (uint64_t)c; c >>= 64; }
    t[4] = (uint64_t)c;
    if (t[4]) reduce5(t);
}
void neg_mod(U256& v) {                                       // v <- (-v) mod p
    if (!(v.l[0] | v.l[1] | v.l[2] | v.l[3])) return;
    u128 br = 0;
    for (int i = 0; i < 4; i++) { const u128 d = (u128)P64[i] - v.l[i] - br; v.l[i] = (uint64_t)d; br = (d >> 127) & 1; }
}


struct Parser {
    const char* s; const char* e; std::string err;
    void ws() { while (s < e && (*s == ' ' || *s == '\n' || *s == '\t' || *s == '\r')) s++; }
    bool fail(const std::string& m) { if (err.empty()) err = m; return false; }
    // an integer literal (digits [s, t), optional sign consumed by the caller), base 10 or 16, reduced mod p
    bool digits(const char* a, const char* b, int base, bool neg, U256& out) {
        if (a == b) return fail("empty number");
        out = U256{{0, 0, 0, 0}};
        if (base == 10 && b - a <= 18) {                       // the common case: bytes and lengths
            uint64_t v = 0;
            for (const char* q = a; q < b; q++) { if (*q < '0' || *q > '9') return fail("bad digit in number"); v = v * 10 + (uint64_t)(*q - '0'); }
            out.l[0] = v;
        } else {
            // chunks of 18 decimal / 15 hex digits: one 256 x 64-bit multiply-add per chunk
            uint64_t t[5] = {0, 0, 0, 0, 0};
            const int per = base == 10 ? 18 : 15;
            for (const char* q = a; q < b;) {
                uint64_t c = 0, m = 1;
                for (int k = 0; k < per && q < b; k++, q++) {
                    int d;
                    if (*q >= '0' && *q <= '9') d = *q - '0';
                    else if (base == 16 && *q >= 'a' && *q <= 'f') d = *q - 'a' + 10;
                    else if (base == 16 && *q >= 'A' && *q <= 'F') d = *q - 'A' + 10;
                    else return fail("bad digit in number");
                    c = c * (uint64_t)base + (uint64_t)d; m *= (uint64_t)base;
                }
                muladd5(t, m, c);
            }
            while (geq_p(t)) sub_p(t);
            for (int i = 0; i < 4; i++) out.l[i] = t[i];
        }
        if (neg) neg_mod(out);
        return true;
    }
    // one scalar JSON value (number / string / true / false) -> field element
    bool scalar(U256& out) {
        ws();
        if (s >= e) return fail("unexpected end of input");
        if (*s == '"') {
            const char* a = ++s;
            while (s < e && *s != '"') { if (*s == '\\') return fail("escape in a numeric string"); s++; }
            if (s >= e) return fail("unterminated string");
            const char* b = s++;
            // witness.to_field: base 16 iff the string itself starts with 0x (Python's int(v, 16)), otherwise base 10 with optional sign;
            // int() strips surrounding whitespace (underscore digit separators are refused here)
            const bool hex = b - a >= 2 && a[0] == '0' && (a[1] == 'x' || a[1] == 'X');
            while (a < b && (*a == ' ' || *a == '\t' || *a == '\n')) a++;
            while (b > a && (b[-1] == ' ' || b[-1] == '\t' || b[-1] == '\n')) b--;
            if (hex) return digits(a + 2, b, 16, false, out);
            bool neg = false;
            if (a < b && (*a == '-' || *a == '+')) { neg = *a == '-'; a++; }
            return digits(a, b, 10, neg, out);
        }
        if (*s == 't' && e - s >= 4 && !memcmp(s, "true", 4)) { s += 4; out = U256{{1, 0, 0, 0}}; return true; }
        if (*s == 'f' && e - s >= 5 && !memcmp(s, "false", 5)) { s += 5; out = U256{{0, 0, 0, 0}}; return true; }
        bool neg = false;
        if (*s == '-') { neg = true; s++; }
        const char* a = s;
        while (s < e && *s >= '0' && *s <= '9') s++;
        if (s < e && (*s == '.' || *s == 'e' || *s == 'E')) return fail("a JSON number with a fraction or exponent is not an integer input");
        if (a == s) return fail("unsupported input value");
        if (s - a > 1 && *a == '0') return fail("a JSON number must not have leading zeros");      // (json.loads refuses 007; the string "007" is fine)
        return digits(a, s, 10, neg, out);
    }
    // a scalar, possibly wrapped in one-element arrays
    bool wrapped_scalar(U256& out) {
        ws();
        int depth = 0;
        while (s < e && *s == '[') { s++; depth++; ws(); if (depth > MAX_NEST) return fail("arrays nested too deeply"); }
        if (!scalar(out)) return false;
        for (; depth; depth--) { ws(); if (s >= e || *s != ']') return fail("a scalar input must be a value or a one-element array"); s++; }
        return true;
    }
    // a (nested) array flattened row-major into int32 slots; returns the element count through n
    // (the recursion is bounded: a hostile text of nothing but '[' must not overflow the stack of a loader thread)
    enum { MAX_NEST = 16 };
    bool flat(int32_t* dst, uint32_t cap, uint32_t& n, bool& big, int depth = 0) {
        ws();
        if (s < e && *s == '[') {
            if (depth >= MAX_NEST) return fail("arrays nested too deeply");
            s++; ws();
            if (s < e && *s == ']') { s++; return true; }
            // the bulk of an input.json is arrays of short non-negative decimal numbers (bytes, lengths) as json.dumps / JSON.stringify write them: "12, 0, 255, ..."
            // -- one tight loop per run of such elements (1-9 digits, no leading zero, then ',' + optional blanks or ']'); anything else falls to the general path below
            // at the element where it was met (10 900 elements per production witness: the loop is the loader's rate)
            while (e - s > 16 && n < cap) {
#if defined(__SSE2__)
                // zero padding (a proof shorter than the circuit's maximum is padded with zeros, tests/main.py:65-178: 44 % of a 10-layer production input): sixteen
                // "0, " (json.dumps) or eight "0," (JSON.stringify) per step
                while (*s == '0' && s[1] == ',' && e - s > 64 && n + 16 <= cap) {
                    const __m128i x0 = _mm_loadu_si128((const __m128i*)s);
                    if (s[2] == ' ') {
                        const __m128i x1 = _mm_loadu_si128((const __m128i*)(s + 16)), x2 = _mm_loadu_si128((const __m128i*)(s + 32));
                        const __m128i p0 = _mm_setr_epi8('0', ',', ' ', '0', ',', ' ', '0', ',', ' ', '0', ',', ' ', '0', ',', ' ', '0');
                        const __m128i p1 = _mm_setr_epi8(',', ' ', '0', ',', ' ', '0', ',', ' ', '0', ',', ' ', '0', ',', ' ', '0', ',');
                        const __m128i p2 = _mm_setr_epi8(' ', '0', ',', ' ', '0', ',', ' ', '0', ',', ' ', '0', ',', ' ', '0', ',', ' ');
                        const __m128i eq = _mm_and_si128(_mm_and_si128(_mm_cmpeq_epi8(x0, p0), _mm_cmpeq_epi8(x1, p1)), _mm_cmpeq_epi8(x2, p2));
                        if (_mm_movemask_epi8(eq) != 0xFFFF || (uint32_t)(uint8_t)s[48] - '0' > 9) break;
                        memset(dst + n, 0, 64); n += 16; s += 48;
                    } else {
                        const __m128i pc = _mm_setr_epi8('0', ',', '0', ',', '0', ',', '0', ',', '0', ',', '0', ',', '0', ',', '0', ',');
                        if (_mm_movemask_epi8(_mm_cmpeq_epi8(x0, pc)) != 0xFFFF || (uint32_t)(uint8_t)s[16] - '0' > 9) break;
                        memset(dst + n, 0, 32); n += 8; s += 16;
                    }
                }
                // 64 bytes of text at a time: bit masks of the digits, commas and blanks; every comma below the first other character closes an element whose start
                // follows from the comma before it, so the elements' positions come from mask arithmetic (a cycle or two each) and their values are computed
                // independently of one another -- the one-element loop below walks a chain of load -> classify -> advance, 15 cycles per element
                if (e - s > 80 && n + 33 <= cap && (uint32_t)(uint8_t)*s - '0' <= 9) {
                    uint64_t D = 0, C = 0, S = 0, Z = 0;
                    for (int k = 0; k < 4; k++) {
                        const __m128i x = _mm_loadu_si128((const __m128i*)(s + 16 * k));
                        const __m128i dg = _mm_sub_epi8(x, _mm_set1_epi8('0'));
                        D |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(dg, _mm_set1_epi8(9)), dg)) << (16 * k);
                        C |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, _mm_set1_epi8(','))) << (16 * k);
                        S |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, _mm_set1_epi8(' '))) << (16 * k);
                        Z |= (uint64_t)(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(dg, _mm_setzero_si128())) << (16 * k);
                    }
                    const uint64_t O = ~(D | C | S);
                    // Whether an element may be taken is decided for the whole block, by mask arithmetic (round 6; round 5 tested every element: its digits, its length, its leading
                    // zero -- half of the 28 operations an element cost).  An element is 1-3 digits, no leading zero, closed by a comma, at most one blank behind the comma:
                    //   a blank anywhere else | a run of four or more digits | a comma not behind a digit | a zero that starts an element and is followed by a digit
                    // and every comma below the first such place (and below the first other character) closes an element that needs no further look.
                    const uint64_t A = C << 1;                                              // positions right behind a comma
                    const uint64_t Es = (A & ~S) | ((A & S) << 1) | 1ull;                   // element starts (the block begins at one)
                    const uint64_t V = (S & ~A) | (D & (D >> 1) & (D >> 2) & (D >> 3)) | (C & ~(D << 1)) | (Z & Es & (D >> 1));
                    const uint64_t upto = O | V;
                    uint64_t Cm = upto ? C & ((1ULL << __builtin_ctzll(upto)) - 1) : C;
                    const bool stop = V != 0 && (!O || __builtin_ctzll(V) < __builtin_ctzll(O));      // the slow path looks at the element that is not plain
                    unsigned start = 0;
                    if (Cm) {
                        const unsigned plast = 63u - (unsigned)__builtin_clzll(Cm);
                        start = plast + 1 + (unsigned)(((S >> plast) >> 1) & 1);                        // where the text goes on behind the last element taken
                        uint64_t Em = Es;                                                                // the i-th start belongs to the i-th comma
                        do {
                            const unsigned p = (unsigned)__builtin_ctzll(Cm), st = (unsigned)__builtin_ctzll(Em), len = p - st;   // 1..3
                            uint32_t w; memcpy(&w, s + st, 4);
                            const uint32_t y = ((w ^ 0x30303030u) << (8 * (3 - len))) & 0xFFFFFFu;           // hundreds | tens | units in bytes 0 | 1 | 2 (what follows the digits is shifted out)
                            dst[n++] = (int32_t)((y & 0xFF) * 100 + ((y >> 8) & 0xFF) * 10 + (y >> 16));
                            Cm &= Cm - 1; Em &= Em - 1;
                        } while (Cm);
                    }
                    s += start;
                    if (start) {                                    // at the next element (or at whatever follows the last comma taken: the loops below decide)
                        if (s < e && *s == ' ') s++;
                        if (s < e && (uint32_t)(uint8_t)*s - '0' <= 9) { if (!stop) continue; }
                        else { ws(); if (s < e && ((uint32_t)(uint8_t)*s - '0' > 9)) goto general; }
                    }
                }
#endif
                const char* a = s;
                // (the zero padding without SSE2: eight "0, " at a time)
                if (*s == '0' && e - s > 40 && n + 8 <= cap) {
                    uint64_t w0, w1, w2; memcpy(&w0, s, 8); memcpy(&w1, s + 8, 8); memcpy(&w2, s + 16, 8);
                    // "0, 0, 0," | " 0, 0, 0" | ", 0, 0, " as little-endian words
                    if (w0 == 0x2C30202C30202C30ULL && w1 == 0x30202C30202C3020ULL && w2 == 0x202C30202C30202CULL && (uint32_t)(uint8_t)s[24] - '0' <= 9) {
                        for (int z = 0; z < 8; z++) dst[n + z] = 0;
                        n += 8; s += 24;
                        continue;
                    }
                }
                uint32_t d = (uint32_t)(uint8_t)*s - '0';
                if (d > 9) break;
                uint32_t v = d; s++;
                while ((d = (uint32_t)(uint8_t)*s - '0') <= 9 && s - a < 10) { v = v * 10 + d; s++; }
                if (s - a > 9 || (s - a > 1 && *a == '0')) { s = a; break; }
                char c = *s;
                if (c == ' ') { const char* q = s; while (e - q > 1 && (*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) q++; c = *q; if (c == ',' || c == ']') s = q; }
                if (c == ',') { dst[n++] = (int32_t)v; s++; if (s < e && *s == ' ') s++; if (s >= e || (uint32_t)(uint8_t)*s - '0' > 9) { ws(); if (s < e && ((uint32_t)(uint8_t)*s - '0' > 9)) goto general; } continue; }
                if (c == ']') { dst[n++] = (int32_t)v; s++; return true; }
                s = a; break;                                  // a fraction, an exponent, a longer number: the general path decides
            }
            for (;;) {
            general:
                if (!flat(dst, cap, n, big, depth + 1)) return false;
                ws();
                if (s < e && *s == ',') { s++; continue; }
                if (s < e && *s == ']') { s++; return true; }
                return fail("expected , or ] in array");
            }
        }
        if (s < e && *s >= '0' && *s <= '9') {                  // the bulk of an input: short non-negative decimal numbers (bytes, lengths)
            const char* a = s; uint64_t v = 0;
            while (s < e && *s >= '0' && *s <= '9' && s - a < 10) { v = v * 10 + (uint64_t)(*s - '0'); s++; }
            if (s - a > 1 && *a == '0') { s = a; }                // (leading zeros: the general path refuses them)
            else if (s < e && (*s == ',' || *s == ']' || *s == ' ' || *s == '\n')) {
                if (n < cap) { if (v >= (1ull << 31)) { big = true; dst[n] = 0x7FFFFFFF; } else dst[n] = (int32_t)v; }
                n++;
                return true;
            }
            s = a;                                                 // longer, or a fraction / exponent follows: the general path decides
        }
        U256 v;
        if (!scalar(v)) return false;
        if (n < cap) {
            if (v.l[1] | v.l[2] | v.l[3] || v.l[0] >= (1ull << 31)) { big = true; dst[n] = 0x7FFFFFFF; } else dst[n] = (int32_t)v.l[0];
        }
        n++;
        return true;
    }
    bool skip_value() {                                          // (an unexpected key: skipped, then reported)
        ws();
        if (s >= e) return fail("unexpected end of input");
        if (*s == '[' || *s == '{') {
            const char open = *s, close = open == '[' ? ']' : '}'; int d = 0;
            for (; s < e; s++) { if (*s == '"') { s++; while (s < e && *s != '"') s += (*s == '\\') ? 2 : 1; } else if (*s == open) d++; else if (*s == close && --d == 0) { s++; return true; } }
            return fail("unterminated value");
        }
        if (*s == '"') { s++; while (s < e && *s != '"') s += (*s == '\\') ? 2 : 1; if (s < e) s++; return true; }
        while (s < e && *s != ',' && *s != '}' && *s != ']') s++;
        return true;
    }
};

struct Shape { int circuit; uint32_t nfr, nsm; uint32_t L, LB, HBy; };
const char* const POB_FR[6] = {"burnKey", "actualBalance", "intendedBalance", "revealAmount", "burnExtraCommitment", "_proofExtraCommitment"};
const char* const SPEND_FR[4] = {"burnKey", "balance", "withdrawnBalance", "extraCommitment"};
const char* const POB_SM[7] = {"numLeafAddressNibbles", "layers", "layerLens", "numLayers", "blockHeader", "blockHeaderLen", "byteSecurityRelax"};

bool make_shape(int circuit, const uint64_t* params, int nparams, Shape& sh, std::string& err) {
    sh.circuit = circuit;
    if (circuit == POB_CIRCUIT_PROOF_OF_BURN) {
        if (nparams != 8) { err = "ProofOfBurn takes 8 template parameters"; return false; }
        for (int k = 0; k < 3; k++) if (params[4 * k + 1] | params[4 * k + 2] | params[4 * k + 3] || params[4 * k] == 0 || params[4 * k] > 64) { err = "unsupported ProofOfBurn parameters"; return false; }
        sh.L = (uint32_t)params[0]; sh.LB = 136u * (uint32_t)params[4]; sh.HBy = 136u * (uint32_t)params[8];
        sh.nfr = 6; sh.nsm = 1 + sh.L * sh.LB + sh.L + 1 + sh.HBy + 2;
        return true;
    }
    if (circuit == POB_CIRCUIT_SPEND) { sh.L = sh.LB = sh.HBy = 0; sh.nfr = 4; sh.nsm = 0; return true; }
    err = "unknown circuit";
    return false;
}

bool pack_one(const Shape& sh, const char* json, uint64_t len, uint8_t* fr_row, int32_t* sm_row, uint32_t* forced, std::string& err) {
    Parser p{json, json + len, {}};
    const int nfr = (int)sh.nfr, nsmn = sh.circuit == POB_CIRCUIT_PROOF_OF_BURN ? 7 : 0;
    const char* const* frn = sh.circuit == POB_CIRCUIT_PROOF_OF_BURN ? POB_FR : SPEND_FR;
    // slot of every small input in the int32 row (declaration order) and its element count
    uint32_t off[7] = {0, 0, 0, 0, 0, 0, 0}, cnt[7] = {1, 1, 1, 1, 1, 1, 1};
    if (nsmn) { cnt[1] = sh.L * sh.LB; cnt[2] = sh.L; cnt[4] = sh.HBy; uint32_t o = 0; for (int k = 0; k < 7; k++) { off[k] = o; o += cnt[k]; } }
    uint32_t seen_fr = 0, seen_sm = 0, big_sm = 0;          // big_sm: bit k = small input k holds a value that does not fit its int32 slot (of its LAST occurrence: a duplicate key replaces the earlier value, like json.loads)
    std::string unexpected;
    p.ws();
    if (p.s >= p.e || *p.s != '{') { err = "input.json must be an object"; return false; }
    p.s++; p.ws();
    if (p.s < p.e && *p.s == '}') p.s++;
    else for (;;) {
        p.ws();
        if (p.s >= p.e || *p.s != '"') { err = "expected a key"; return false; }
        const char* a = ++p.s;
        while (p.s < p.e && *p.s != '"') p.s++;
        if (p.s >= p.e) { err = "unterminated key"; return false; }
        const size_t kl = (size_t)(p.s - a); p.s++;
        p.ws();
        if (p.s >= p.e || *p.s != ':') { err = "expected :"; return false; }
        p.s++;
        int hit = -1;
        for (int k = 0; k < nfr && hit < 0; k++) if (strlen(frn[k]) == kl && !memcmp(frn[k], a, kl)) hit = k;
        if (hit >= 0) {
            U256 v;
            if (!p.wrapped_scalar(v)) { err = std::string(frn[hit]) + ": " + p.err; return false; }
            memcpy(fr_row + 32 * hit, v.l, 32);
            seen_fr |= 1u << hit;
        } else {
            for (int k = 0; k < nsmn && hit < 0; k++) if (strlen(POB_SM[k]) == kl && !memcmp(POB_SM[k], a, kl)) hit = k;
            if (hit >= 0) {
                bool big = false;
                if (cnt[hit] == 1 && hit != 2) {               // scalar small input (layerLens is an array even when maxNumLayers = 1)
                    U256 v;
                    if (!p.wrapped_scalar(v)) { err = std::string(POB_SM[hit]) + ": " + p.err; return false; }
                    if (v.l[1] | v.l[2] | v.l[3] || v.l[0] >= (1ull << 31)) { big = true; sm_row[off[hit]] = 0x7FFFFFFF; } else sm_row[off[hit]] = (int32_t)v.l[0];
                } else {
                    uint32_t n = 0;
                    if (!p.flat(sm_row + off[hit], cnt[hit], n, big)) { err = std::string(POB_SM[hit]) + ": " + p.err; return false; }
                    if (n != cnt[hit]) { err = std::string(POB_SM[hit]) + " has " + std::to_string(n) + " elements, circuit expects " + std::to_string(cnt[hit]); return false; }
                }
                seen_sm |= 1u << hit;
                big_sm = big ? (big_sm | (1u << hit)) : (big_sm & ~(1u << hit));
            } else {
                if (!unexpected.empty()) unexpected += ", ";
                unexpected.append(a, kl);
                if (!p.skip_value()) { err = p.err; return false; }
            }
        }
        p.ws();
        if (p.s < p.e && *p.s == ',') { p.s++; continue; }
        if (p.s < p.e && *p.s == '}') { p.s++; break; }
        err = "expected , or } in object"; return false;
    }
    p.ws();
    if (p.s != p.e) { err = "trailing characters after the object"; return false; }
    std::string missing;
    for (int k = 0; k < nfr; k++) if (!((seen_fr >> k) & 1)) { if (!missing.empty()) missing += ", "; missing += frn[k]; }
    for (int k = 0; k < nsmn; k++) if (!((seen_sm >> k) & 1)) { if (!missing.empty()) missing += ", "; missing += POB_SM[k]; }
    if (!missing.empty() || !unexpected.empty()) { err = "missing [" + missing + "] unexpected [" + unexpected + "]"; return false; }
    *forced = big_sm ? FAIL_INPUT_RANGE : 0;
    return true;
}

// ---- the loader's thread pool: workers sleep on a condition variable between batches; a job is "call fn(worker) on `want` workers"; the caller works too
class LoaderPool {
    std::mutex mu; std::condition_variable cv_work, cv_done;
    std::vector<std::thread> th;
    const std::function<void(uint32_t)>* job = nullptr; uint64_t gen = 0; uint32_t want = 0, taken = 0, done = 0; bool stop = false;
    void loop() {
        // a pool thread yields to the thread that drives the GPU: when the loader fills every CPU the process may use (a service loop parsing batch k + 2 beside the
        // device's batch k), the enqueueing thread must not wait for a time slice -- the device would idle for it.  (Linux: nice is per thread)
        setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), 10);
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(uint32_t)>* f = nullptr; uint32_t slot = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (gen != seen && taken < want); });
                if (stop) return;
                seen = gen;
                f = job; slot = ++taken;                     // (slot 0 is the caller)
            }
            (*f)(slot);
            { std::lock_guard<std::mutex> lk(mu); if (++done == want) cv_done.notify_one(); }
        }
    }
public:
    std::mutex call_mu;                                      // one batch at a time through the pool (callers on other threads queue up)
    ~LoaderPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv_work.notify_all(); for (std::thread& t : th) t.join(); }
    // fn(0) on the calling thread and fn(1 .. helpers) on pool threads; returns when all are done
    void run(uint32_t helpers, const std::function<void(uint32_t)>& fn) {
        if (helpers) {
            std::lock_guard<std::mutex> lk(mu);
            while (th.size() < helpers) th.emplace_back([this] { loop(); });
            job = &fn; want = helpers; taken = done = 0; gen++;
        }
        if (helpers) cv_work.notify_all();
        fn(0);
        if (helpers) { std::unique_lock<std::mutex> lk(mu); cv_done.wait(lk, [&] { return done == want; }); want = 0; job = nullptr; }
    }
};
// (leaked on purpose: no join of sleeping threads at exit; a forked child has none of the parent's threads and starts a pool of its own)
LoaderPool* g_loader_pool = nullptr;
std::mutex g_loader_pool_mu;
LoaderPool& loader_pool() {
    std::lock_guard<std::mutex> lk(g_loader_pool_mu);
    if (!g_loader_pool) {
        static bool hooked = false;
        if (!hooked) { pthread_atfork(nullptr, nullptr, [] { g_loader_pool = nullptr; new (&g_loader_pool_mu) std::mutex(); }); hooked = true; }
        g_loader_pool = new LoaderPool();
    }
    return *g_loader_pool;
}

uint32_t default_threads() {
    uint32_t n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (uint32_t)CPU_COUNT(&set);
    const bool pinned = n && n < std::thread::hardware_concurrency();      // an affinity mask narrower than the machine: somebody placed this process
    bool quota_cut = false;
    if (!n) n = std::thread::hardware_concurrency();
    if (!n) n = 1;
    // a container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota_us): more runnable threads than the quota are throttled, not faster (the GPU boxes of this project
    // show 256 CPUs and grant 16: 64 threads took 10 ms for a batch that 32 parsed in 2.9 ms, round 4)
    {
        long quota = -1, period = 100000;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[32]; if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max")) quota = atol(q); fclose(f); }
        else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g);
            if (FILE* h2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h2, "%ld", &period) != 1) period = 100000; fclose(h2); } }
        // the pool is 1.5 x the quota wide: in a service loop its threads work in bursts (a batch of 1 024 is 11.5 ms of CPU time per 1.05 ms step: 11 CPUs on average), and a burst
        // finishes sooner on more threads -- 16 / 20 / 24 / 32 threads under a quota of 16: 0.92-1.07 / 0.77-0.94 / 0.72-0.76 / 0.64-0.96 ms per batch inside the loop, the last with
        // throttled steps (profiles/round6_experiments.txt 19)
        if (quota > 0 && period > 0) { const uint32_t q = (uint32_t)((quota + period - 1) / period), w = q + q / 2; if (q && w < n) { n = w; quota_cut = true; } }      // (a quota is the container's: its ranks share it)
    }
    if (const char* e = getenv("POB_LOADER_THREADS")) { const int v = atoi(e); if (v > 0) return (uint32_t)v; }
    // the ranks of a node share its cores: divide by LOCAL_WORLD_SIZE -- unless this process has been pinned to its share already (distributed.bind_rank_to_gpu_numa)
    // -- a mask narrower than the machine is not enough for that: a job-wide cpuset / taskset that all ranks share looks the same.  The rank holds its share when
    // bind_rank_to_gpu_numa said so (POB_RANK_BOUND=1) or when its mask is no wider than the machine divided by the ranks.
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) {
        const int v = atoi(e);
        const char* b = getenv("POB_RANK_BOUND");
        const uint32_t hw = std::thread::hardware_concurrency();
        const bool has_share = pinned && ((b && b[0] == '1') || (v > 0 && hw && n <= hw / (uint32_t)v));
        if (v > 1 && (!has_share || quota_cut)) n = n / (uint32_t)v ? n / (uint32_t)v : 1;
    }
    return n;
}

// one int32 row -> bytes + exception slots; false: more than POB_EXC_CAP values outside 0..255
bool narrow_row(const int32_t* row, uint32_t nsm, uint8_t* b, pob_sm_exc_t* ex) {
    uint32_t ne = 0, k0 = 0;
#if defined(__SSE2__)
    for (; k0 + 16 <= nsm; k0 += 16) {                        // sixteen values at a time while all of them are bytes (all but a handful of a witness' 10 900 are)
        const __m128i a = _mm_loadu_si128((const __m128i*)(row + k0)), c = _mm_loadu_si128((const __m128i*)(row + k0 + 4));
        const __m128i d = _mm_loadu_si128((const __m128i*)(row + k0 + 8)), f = _mm_loadu_si128((const __m128i*)(row + k0 + 12));
        const __m128i hi = _mm_andnot_si128(_mm_set1_epi32(255), _mm_or_si128(_mm_or_si128(a, c), _mm_or_si128(d, f)));
        if (_mm_movemask_epi8(_mm_cmpeq_epi32(hi, _mm_setzero_si128())) == 0xFFFF) {
            _mm_storeu_si128((__m128i*)(b + k0), _mm_packus_epi16(_mm_packs_epi32(a, c), _mm_packs_epi32(d, f)));
            continue;
        }
        for (uint32_t k = k0; k < k0 + 16; k++) {
            const int32_t v = row[k];
            if ((uint32_t)v < 256u) b[k] = (uint8_t)v;
            else { b[k] = 0; if (ne < POB_EXC_CAP) { ex[ne].k = k; ex[ne].v = v; } ne++; }
        }
    }
#endif
    for (uint32_t k = k0; k < nsm; k++) {
        const int32_t v = row[k];
        if ((uint32_t)v < 256u) b[k] = (uint8_t)v;
        else { b[k] = 0; if (ne < POB_EXC_CAP) { ex[ne].k = k; ex[ne].v = v; } ne++; }
    }
    for (uint32_t e = ne; e < POB_EXC_CAP; e++) { ex[e].k = POB_EXC_NONE; ex[e].v = 0; }
    return ne <= POB_EXC_CAP;
}

// n texts over the pool; sm8 / exc: the byte form (each worker parses into a row of its own and narrows it), else sm: the int32 form
int pack_batch(const Shape& sh, const char* const* json, const uint64_t* len, uint32_t n, int threads, uint8_t* fr, int32_t* sm, uint8_t* sm8, pob_sm_exc_t* exc,
               uint32_t* forced, char* err, uint32_t errcap);
void put_err(char* dst, uint32_t cap, const std::string& m) { if (dst && cap) { snprintf(dst, cap, "%s", m.c_str()); } }
int pack_batch(const Shape& sh, const char* const* json, const uint64_t* len, uint32_t n, int threads, uint8_t* fr, int32_t* sm, uint8_t* sm8, pob_sm_exc_t* exc,
               uint32_t* forced, char* err, uint32_t errcap) {
    uint32_t nt = threads > 0 ? (uint32_t)threads : default_threads();
    if (nt > (n + 3) / 4) nt = (n + 3) / 4;                        // at least four texts (~0.5 ms of parsing) per thread: waking a sleeping worker costs tens of microseconds
    if (nt == 0) nt = 1;
    std::atomic<uint32_t> next(0), bad(0xFFFFFFFFu), overflow(0), oom(0);
    std::vector<std::string> errs(nt);
    const uint32_t smw = sh.nsm ? sh.nsm : 1;                      // (witness.py keeps one dummy column for circuits without small inputs)
    const bool narrow = sm8 != nullptr;
    std::function<void(uint32_t)> work = [&](uint32_t t) {
        uint32_t i = 0;
        try {                                                        // (an exception must not leave a pool thread: std::terminate would take the caller's process with it)
            std::vector<int32_t> row(narrow ? smw : 0);            // byte form: the int32 row of the text being parsed (43 KB: stays in the core's cache)
            for (;;) {
                i = next.fetch_add(1);
                if (i >= n) return;
                std::string em;
                int32_t* dst = narrow ? row.data() : (sm ? sm + (uint64_t)i * smw : nullptr);
                if (!pack_one(sh, json[i], len[i], fr + (uint64_t)i * sh.nfr * 32, sh.nsm ? dst : nullptr, forced + i, em)) {
                    uint32_t cur = bad.load();
                    while (i < cur && !bad.compare_exchange_weak(cur, i)) {}
                    if (errs[t].empty()) errs[t] = "input " + std::to_string(i) + ": " + em;
                } else if (narrow && sh.nsm) {
                    if (!narrow_row(row.data(), sh.nsm, sm8 + (uint64_t)i * sh.nsm, exc + (uint64_t)i * POB_EXC_CAP)) overflow.store(1);
                }
            }
        } catch (...) {                                              // out of memory while parsing: the batch is refused
            const uint32_t at = i < n ? i : 0;
            uint32_t cur = bad.load();
            while (at < cur && !bad.compare_exchange_weak(cur, at)) {}
            oom.store(1);
        }
    };
    {
        LoaderPool& P = loader_pool();
        std::lock_guard<std::mutex> lk(P.call_mu);
        P.run(nt - 1, work);
    }
    if (oom.load()) { put_err(err, errcap, "out of memory in the loader"); return POB_E_NOMEM; }      // (not POB_E_ARG: the input may be fine)
    if (bad.load() != 0xFFFFFFFFu) { for (const std::string& m : errs) if (!m.empty()) { put_err(err, errcap, m); break; } return POB_E_ARG; }
    if (overflow.load()) { put_err(err, errcap, "a witness has more than POB_EXC_CAP small inputs outside 0..255: use the int32 form for this batch"); return POB_E_RANGE; }
    return POB_OK;
}
}  // namespace

extern "C" {

int pob_pack_json(int circuit, const uint64_t* params, int nparams, const char* json, uint64_t len, uint8_t* fr_row, int32_t* sm_row, uint32_t* forced, char* err, uint32_t errcap) {
    if (!params || !json || !fr_row || !forced) return POB_E_ARG;
    Shape sh; std::string e;
    if (!make_shape(circuit, params, nparams, sh, e)) { put_err(err, errcap, e); return POB_E_ARG; }
    if (sh.nsm && !sm_row) return POB_E_ARG;
    if (!pack_one(sh, json, len, fr_row, sm_row, forced, e)) { put_err(err, errcap, e); return POB_E_ARG; }
    return POB_OK;
}

int pob_pack_json_batch(int circuit, const uint64_t* params, int nparams, const char* const* json, const uint64_t* len, uint32_t n, int threads,
                        uint8_t* fr, int32_t* sm, uint32_t* forced, char* err, uint32_t errcap) {
    if (!params || !json || !len || !fr || !forced || n == 0) return POB_E_ARG;
    Shape sh; std::string e;
    if (!make_shape(circuit, params, nparams, sh, e)) { put_err(err, errcap, e); return POB_E_ARG; }
    if (sh.nsm && !sm) return POB_E_ARG;
    return pack_batch(sh, json, len, n, threads, fr, sm, nullptr, nullptr, forced, err, errcap);
}

int pob_pack_json_batch8(int circuit, const uint64_t* params, int nparams, const char* const* json, const uint64_t* len, uint32_t n, int threads,
                         uint8_t* fr, uint8_t* sm8, pob_sm_exc_t* exc, uint32_t* forced, char* err, uint32_t errcap) {
    if (!params || !json || !len || !fr || !forced || n == 0) return POB_E_ARG;
    Shape sh; std::string e;
    if (!make_shape(circuit, params, nparams, sh, e)) { put_err(err, errcap, e); return POB_E_ARG; }
    if (sh.nsm && (!sm8 || !exc)) return POB_E_ARG;
    return pack_batch(sh, json, len, n, threads, fr, nullptr, sm8, exc, forced, err, errcap);
}

int pob_narrow_inputs(const int32_t* sm, uint32_t n, uint32_t nsm, uint8_t* sm8, pob_sm_exc_t* exc) {
    if (!sm || !sm8 || !exc || n == 0 || nsm == 0) return POB_E_ARG;
    bool ok = true;
    for (uint32_t i = 0; i < n; i++) ok &= narrow_row(sm + (uint64_t)i * nsm, nsm, sm8 + (uint64_t)i * nsm, exc + (uint64_t)i * POB_EXC_CAP);
    return ok ? POB_OK : POB_E_RANGE;
}

}  // extern "C"
