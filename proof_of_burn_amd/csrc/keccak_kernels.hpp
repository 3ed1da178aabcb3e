// Bit-sliced Keccak-f[1600] witness expansion / constraint evaluation (keccak.circom:19-367).
//
// Layout: one wavefront = one (64-witness group, sponge/permutation, round); LANE = BIT POSITION k of the
// 64-bit Keccak lane, and each 64-bit register word holds bit k of that Keccak lane for the 64 witnesses
// of the group.  theta/chi/iota are plain 64-bit ALU ops on those words, the rho rotations are cross-lane
// permutes (ds_bpermute), and every `signal x[64]` array of the circuit is ONE coalesced 512-byte
// store/load (lane k <-> wire x[k]).  The O0 wire offsets below restate the component tree of
// KeccakfRound: own signals, then children in instantiation order.
//
// One templated walker (`round_walk`) visits all 102 656 wires of a round; `GenIO` stores them, `CheckIO`
// loads them and ORs every mismatch into a per-witness mask: generator and constraint evaluator cannot
// disagree about the layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_common.hpp"
#ifdef POB_HOSTSIM
#define POB_WAVES_PER_SIMD(n)
#else
#define POB_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

static __device__ __constant__ u64 KECCAK_RC_DEV[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// RhoPi's lane walk (keccak.circom:195): out[rot[i+1]] = rotl(in[rot[i]], ((i+1)(i+2)/2) % 64)
__host__ __device__ constexpr int krot_(int i) {
    constexpr int t[25] = {1, 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    return t[i];
}
#define KROT(i) krot_(i)

// ---- what a KeccakfRound block stores (round 4; circuits.hpp KR_STORED).  A wire of the block is named by a 14-bit code
//      (slot << 7) | (bit position << 1) | negated:  slot KS_IN + i  = midRound[r][i]        (Keccakf's own wires, written by k_chain)
//                                                    slot KS_ST + s  = stored array s of the round block (the 76 gate outputs below)
//                                                    slot KS_OUT + i = midRound[r+1][i]
//                                                    KS_ZERO (negated: the constant 1), KS_RC: bit `position` of the round constant
enum : uint32_t { KS_IN = 0, KS_ST = 25, KS_OUT = 101, KS_ZERO = 126, KS_RC = 127 };
// stored arrays: Xor5 x: ab, abc, abcd, out = 4x .. 4x+3 | D x: out = 20 + x | Theta out[idx] = 25 + idx | stepChi i: AndArray out = 50 + i | stepChi 0 out = 75
#define KSL_X5(x, k) (4u * (x) + (k))
#define KSL_D(x) (20u + (x))
#define KSL_TH(i) (25u + (i))
#define KSL_AND(i) (50u + (i))
#define KSL_CHI0 75u

// value held by lane (lane - r) mod 64  (rotl by r in bit-position space)
__device__ __forceinline__ u64 lane_rotl(u64 v, int r, uint32_t lane) {
    int src = (int)((lane - (uint32_t)r) & 63u);
    uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t lane_rotl(uint32_t v, int r, uint32_t lane) { return __shfl(v, (int)((lane - (uint32_t)r) & 63u), 64); }

// Device side of the walker: V = the 64-witness mask of bit position `lane`.  arr / gate (the alias wires) are nothing here.
struct DevIOBase {
    typedef u64 V;
    uint32_t lane, lane4;
    // rotation addresses from an opaque copy of the lane's byte address (refresh(): once per round): otherwise the 25 loop-invariant ds_bpermute addresses
    // of a round are hoisted out of the kernel's item / round loop and held -- or spilled -- across it
    __device__ __forceinline__ void refresh() { lane4 = lane * 4u; POB_OPAQUE(lane4); }
    __device__ __forceinline__ V rotl(V v, int r) const {
        const int a = (int)((lane4 + 4u * (64u - (uint32_t)r)) & 252u);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)(uint32_t)(v >> 32));
        return ((u64)hi << 32) | lo;
    }
    __device__ __forceinline__ V keep_ge(V v, int s) const { return (int)lane >= s ? v : 0; }
    __device__ __forceinline__ V keep_lt(V v, int s) const { return (int)lane < s ? v : 0; }
    __device__ __forceinline__ V or_disjoint(V, V, V whole) const { return whole; }       // (a | b with disjoint supports = the unmasked rotation)
    __device__ __forceinline__ V not_(V v) const { return ~v; }
    __device__ __forceinline__ V rc(int r) const { return ((KECCAK_RC_DEV[r] >> lane) & 1) ? ~0ULL : 0ULL; }
    __device__ __forceinline__ void arr(uint32_t, V) const {}
    __device__ __forceinline__ void gate(uint32_t, V, V, V) const {}
    __device__ __forceinline__ void out(int, V) const {}
};
// generation: the 76 gate outputs are stored; the round's output state is midRound[r+1], which k_chain wrote -- and which the wavefront has just
// computed: it stays in registers (s[]) as the next round's input, so a wavefront that expands several consecutive rounds reads midRound[r0] only
struct GenIO : DevIOBase {
    u64* st; V s[25];
    __device__ __forceinline__ V in(int i) const { return s[i]; }
    // (non-temporal stores for the expansion's 1.26 GB per batch: the kernel alone and the loop with 8 / 12 calculators in flight within 0.5 %, profiles/round6_experiments.txt)
    __device__ __forceinline__ void stw(u64* p, V v) const { *p = v; }
    __device__ __forceinline__ V gx(uint32_t sl, V a, V b) const { const V v = a ^ b; stw(st + 64 * sl + lane, v); return v; }
    __device__ __forceinline__ V ga(uint32_t sl, V a, V b) const { const V v = a & b; stw(st + 64 * sl + lane, v); return v; }
    __device__ __forceinline__ V gxo(int, V a, V b) const { return a ^ b; }
    __device__ __forceinline__ void out(int i, V v) { s[i] = v; }
};
// The 101 loads of a round come in the walk's fixed order (kchk_seq); the compiler, held to 128 VGPRs, issues ONE load per wait whatever it is told, so the
// loads in flight are written into the source: a ring of DP prefetched arrays -- the gate that consumes load q issues load q + DP, and a scheduling barrier
// after every gate keeps each load where it is written.
//   q = 0..24: stored slot q (Xor5 partials, D) | 25..49: theta out, i outer / j inner | 50 + 2i: AND(i), 51 + 2i: chi out i (i = 0: the stored stepChi 0 out) | 100: iota out 0
HD constexpr int kchk_seq(int q) {        // >= 0: word offset in the round's stored arrays; < 0: -(word offset in midRound[r+1]) - 1
    if (q < 25) return 64 * q;
    if (q < 50) return 64 * (25 + (q - 25) / 5 + 5 * ((q - 25) % 5));
    if (q == 100) return -1;
    if ((q & 1) == 0) return 64 * (50 + (q - 50) / 2);
    return q == 51 ? 64 * 75 : -(64 * ((q - 51) / 2)) - 1;
}
// ... and its inverse: the position of an array in the sequence (a constant wherever the walk is unrolled: the ring is indexed by constants)
HD constexpr int kchk_pos(int want) {
    if (want < 0) { const int i = (-want - 1) / 64; return i == 0 ? 100 : 51 + 2 * i; }
    const int sl = want / 64;
    if (sl < 25) return sl;
    if (sl < 50) return 25 + (sl - 25) % 5 * 5 + (sl - 25) / 5;
    return sl == 75 ? 51 : 50 + 2 * (sl - 50);
}
#define KCHK_LOADS 101
template <bool NT, int DP = 0> struct CheckIOT : DevIOBase {
    const u64* st; const u64* out_; u64 bad; V s[25];
    V ring[DP ? DP : 1];
    __device__ __forceinline__ u64 ldw(const u64* p) const { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
    __device__ __forceinline__ V ldq(int k) const { const int o = kchk_seq(k); return o >= 0 ? ldw(st + o + lane) : ldw(out_ + (-o - 1) + lane); }
    __device__ __forceinline__ void begin_round() {
        refresh();
        if constexpr (DP > 0) {
#pragma unroll
            for (int k = 0; k < DP; k++) ring[k] = ldq(k);
        }
    }
    // the array at `want` (kchk_seq's coding): it was requested DP gates ago; its slot of the ring takes the array DP positions further on
    __device__ __forceinline__ V next(int want) {
        V v;
        if constexpr (DP > 0) {
            const int q = kchk_pos(want);
            v = ring[q % DP];
            if (q + DP < KCHK_LOADS) ring[q % DP] = ldq(q + DP);
            __builtin_amdgcn_sched_barrier(0);
        } else v = want >= 0 ? ldw(st + want + lane) : ldw(out_ + (-want - 1) + lane);
        return v;
    }
    __device__ __forceinline__ V in(int i) const { return s[i]; }
    __device__ __forceinline__ V gx(uint32_t sl, V a, V b) { const V v = next(64 * (int)sl); bad |= v ^ a ^ b; return v; }
    __device__ __forceinline__ V ga(uint32_t sl, V a, V b) { const V v = next(64 * (int)sl); bad |= v ^ (a & b); return v; }
    __device__ __forceinline__ V gxo(int i, V a, V b) { const V v = next(-(64 * i) - 1); bad |= v ^ a ^ b; return v; }
    __device__ __forceinline__ void out(int i, V v) { s[i] = v; }
};
// host side: V = the 64 codes of an array; the walk fills the alias table of the block (one per library, shared by every round: the only
// round-dependent wires are the round constants, coded KS_RC)
struct SymV { uint16_t e[64]; };
struct SymIO {
    typedef SymV V;
    uint16_t* tab; bool ok;
    static uint16_t enc(uint32_t slot, uint32_t k, uint32_t neg = 0) { return (uint16_t)((slot << 7) | (k << 1) | neg); }
    static V all(uint32_t slot) { V v; for (uint32_t k = 0; k < 64; k++) v.e[k] = enc(slot, k); return v; }
    V in(int i) { return all(KS_IN + (uint32_t)i); }
    V gx(uint32_t s, const V&, const V&) { return all(KS_ST + s); }
    V ga(uint32_t s, const V&, const V&) { return all(KS_ST + s); }
    V gxo(int i, const V&, const V&) { return all(KS_OUT + (uint32_t)i); }
    V rotl(const V& v, int r) { V o; for (int k = 0; k < 64; k++) o.e[k] = v.e[(k - r) & 63]; return o; }
    V keep_ge(const V& v, int s) { V o; for (int k = 0; k < 64; k++) o.e[k] = k >= s ? v.e[k] : enc(KS_ZERO, 0); return o; }
    V keep_lt(const V& v, int s) { V o; for (int k = 0; k < 64; k++) o.e[k] = k < s ? v.e[k] : enc(KS_ZERO, 0); return o; }
    V or_disjoint(const V& a, const V& b, const V& whole) {
        V o;
        for (int k = 0; k < 64; k++) {
            const uint16_t z = enc(KS_ZERO, 0);
            if (a.e[k] != z && b.e[k] != z) ok = false;               // an OrArray whose operands overlap would be a gate of its own
            o.e[k] = a.e[k] != z ? a.e[k] : b.e[k];
            if (o.e[k] != whole.e[k]) ok = false;
        }
        return o;
    }
    V not_(const V& v) { V o; for (int k = 0; k < 64; k++) o.e[k] = v.e[k] ^ 1; return o; }
    V rc(int) { V v; for (uint32_t k = 0; k < 64; k++) v.e[k] = enc(KS_RC, k); return v; }
    void arr(uint32_t off, const V& v) { for (uint32_t k = 0; k < 64; k++) set(off + k, v.e[k]); }
    void gate(uint32_t off, const V& o, const V& a, const V& b) { for (uint32_t k = 0; k < 64; k++) { set(off + 3 * k, o.e[k]); set(off + 3 * k + 1, a.e[k]); set(off + 3 * k + 2, b.e[k]); } }
    void out(int, const V&) {}
    void set(uint32_t w, uint16_t e) { if (w >= KECCAKF_ROUND_WIRES || tab[w] != 0xFFFFu) ok = false; else tab[w] = e; }     // every wire exactly once
};

// XorArray/OrArray/AndArray(64) block: [out | a | b | 64 x (o,a,b)]
template <class IO> HD void garr(IO& io, uint32_t off, const typename IO::V& o, const typename IO::V& a, const typename IO::V& b) {
    io.arr(off, o); io.arr(off + 64, a); io.arr(off + 128, b); io.gate(off + 192, o, a, b);
}

// the bare permutation round on the bit-sliced state (no wires); W = u64: bit k of a Keccak lane for the 64 witnesses of a group, W = uint32_t: for 32 of them
template <class W> __device__ __forceinline__ void round_native(W* a, int r, uint32_t lane) {
    W c[5], d[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ lane_rotl(c[(x + 1) % 5], 1, lane);
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    b[0] = a[0];
#pragma unroll
    for (int i = 0; i < 24; i++) b[KROT(i + 1)] = lane_rotl(a[KROT(i)], ((i + 1) * (i + 2) / 2) % 64, lane);
#pragma unroll
    for (int i = 0; i < 25; i++) { const int y = i / 5 * 5; a[i] = b[i] ^ (~b[y + (i + 1) % 5] & b[y + (i + 2) % 5]); }
    a[0] ^= ((KECCAK_RC_DEV[r] >> lane) & 1) ? ~(W)0 : (W)0;
}

// KeccakfRound(r) keccak.circom:290-297: every wire of the block, in O0 order, from the round input midRound[r].  The GATES (io.gx / io.ga /
// io.gxo: XorArray / AndArray outputs) are what generation stores and the evaluator checks; every io.arr / io.gate names an alias wire
// (offset within the block, value) and is compiled to nothing on the device.
template <class IO> HD void round_walk(IO& io, int r) {
    typedef typename IO::V V;
    V in[25], th[25], rp[25], ch[25], out[25];
#pragma unroll
    for (int i = 0; i < 25; i++) in[i] = io.in(i);
    // ---- own: out@0 in@1600 theta@3200 rhopi@4800 chi@6400
#pragma unroll
    for (int i = 0; i < 25; i++) io.arr(1600 + 64 * i, in[i]);
    // ---- Theta @8000 (:151-170): out@0 in@1600 c@3200 d@3520 | Xor5 x5 @3840 (2112 each) | D x5 @14400 (1408 each) | XorArray x25 @21440
    {
        const uint32_t T = 8000;
        V C[5], Dv[5];
#pragma unroll
        for (int i = 0; i < 25; i++) io.arr(T + 1600 + 64 * i, in[i]);
#pragma unroll
        for (int x = 0; x < 5; x++) {   // Xor5(64) :58-70  [out | a,b,c,d,e | xor_ab, xor_abc, xor_abcd] || XorArray x4
            const uint32_t X = T + 3840 + 2112 * x;
            const V a = in[x], b = in[5 + x], c = in[10 + x], d = in[15 + x], e = in[20 + x];
            const V ab = io.gx(KSL_X5(x, 0), a, b), abc = io.gx(KSL_X5(x, 1), ab, c), abcd = io.gx(KSL_X5(x, 2), abc, d), o = io.gx(KSL_X5(x, 3), abcd, e);
            io.arr(X, o); io.arr(X + 64, a); io.arr(X + 128, b); io.arr(X + 192, c); io.arr(X + 256, d); io.arr(X + 320, e);
            io.arr(X + 384, ab); io.arr(X + 448, abc); io.arr(X + 512, abcd);
            garr(io, X + 576, ab, a, b); garr(io, X + 960, abc, ab, c); garr(io, X + 1344, abcd, abc, d); garr(io, X + 1728, o, abcd, e);
            C[x] = o; io.arr(T + 3200 + 64 * x, o);
        }
#pragma unroll
        for (int x = 0; x < 5; x++) {   // D :135-144  [out | a, b | aux0, aux1, aux2] || ShL(64,1), ShR(64,63), OrArray, XorArray
            const uint32_t Dd = T + 14400 + 1408 * x;
            const V a = C[(x + 1) % 5], b = C[(x + 4) % 5];
            const V rot = io.rotl(a, 1);
            const V aux0 = io.keep_ge(rot, 1), aux1 = io.keep_lt(rot, 1), aux2 = io.or_disjoint(aux0, aux1, rot);
            const V o = io.gx(KSL_D(x), b, aux2);
            io.arr(Dd, o); io.arr(Dd + 64, a); io.arr(Dd + 128, b); io.arr(Dd + 192, aux0); io.arr(Dd + 256, aux1); io.arr(Dd + 320, aux2);
            io.arr(Dd + 384, aux0); io.arr(Dd + 448, a);          // ShL(64,1)  [out | in]
            io.arr(Dd + 512, aux1); io.arr(Dd + 576, a);          // ShR(64,63)
            garr(io, Dd + 640, aux2, aux0, aux1); garr(io, Dd + 1024, o, b, aux2);
            Dv[x] = o; io.arr(T + 3520 + 64 * x, o);
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
#pragma unroll
            for (int j = 0; j < 5; j++) {   // :165-169, i outer, j inner
                const int idx = i + 5 * j;
                const V o = io.gx(KSL_TH(idx), in[idx], Dv[i]);
                garr(io, T + 21440 + 384 * (i * 5 + j), o, in[idx], Dv[i]);
                th[idx] = o; io.arr(T + 64 * idx, o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 25; i++) io.arr(3200 + 64 * i, th[i]);
    // ---- RhoPi @39040 (:191-204): out@0 in@1600 | stepRhoPi x24 @3200 (896 each): [out | a | aux0, aux1] || ShR, ShL, OrArray
    {
        const uint32_t Rb = 39040;
#pragma unroll
        for (int i = 0; i < 25; i++) io.arr(Rb + 1600 + 64 * i, th[i]);
        rp[0] = th[0]; io.arr(Rb, th[0]);
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const uint32_t S_ = Rb + 3200 + 896 * i;
            const int shl = ((i + 1) * (i + 2) / 2) % 64;
            const V a = th[KROT(i)];
            const V rot = io.rotl(a, shl);
            const V aux0 = io.keep_lt(rot, shl);                 // ShR(64, 64-shl): out[k] = a[k + 64 - shl]
            const V aux1 = io.keep_ge(rot, shl);                 // ShL(64, shl):    out[k] = a[k - shl]
            const V o = io.or_disjoint(aux0, aux1, rot);
            io.arr(S_, o); io.arr(S_ + 64, a); io.arr(S_ + 128, aux0); io.arr(S_ + 192, aux1);
            io.arr(S_ + 256, aux0); io.arr(S_ + 320, a);
            io.arr(S_ + 384, aux1); io.arr(S_ + 448, a);
            garr(io, S_ + 512, o, aux0, aux1);
            rp[KROT(i + 1)] = o; io.arr(Rb + 64 * KROT(i + 1), o);
        }
    }
#pragma unroll
    for (int i = 0; i < 25; i++) io.arr(4800 + 64 * i, rp[i]);
    // ---- Chi @63744 (:228-241): out@0 in@1600 | stepChi x25 @3200 (1280 each): [out | a,b,c | bXor, bc] || NotArray, AndArray, XorArray
    {
        const uint32_t Cb = 63744;
#pragma unroll
        for (int i = 0; i < 25; i++) io.arr(Cb + 1600 + 64 * i, rp[i]);
#pragma unroll
        for (int i = 0; i < 25; i++) {
            const uint32_t C_ = Cb + 3200 + 1280 * i;
            const int y = i / 5 * 5;
            const V a = rp[i], b = rp[y + (i + 1) % 5], c = rp[y + (i + 2) % 5];
            const V bx = io.not_(b), bc = io.ga(KSL_AND(i), bx, c);
            const V o = i == 0 ? io.gx(KSL_CHI0, a, bc) : io.gxo(i, a, bc);        // chi.out[i], i > 0, IS the round's output = midRound[r+1][i]
            io.arr(C_, o); io.arr(C_ + 64, a); io.arr(C_ + 128, b); io.arr(C_ + 192, c); io.arr(C_ + 256, bx); io.arr(C_ + 320, bc);
            io.arr(C_ + 384, bx); io.arr(C_ + 448, b);           // NotArray [out | a]
            garr(io, C_ + 512, bc, bx, c); garr(io, C_ + 896, o, a, bc);
            ch[i] = o; io.arr(Cb + 64 * i, o);
        }
    }
#pragma unroll
    for (int i = 0; i < 25; i++) io.arr(6400 + 64 * i, ch[i]);
    // ---- Iota(r) @98944 (:273-283): out@0 in@1600 roundConstants@3200 | RoundConstants @3264 | XorArray @3328
    {
        const uint32_t Ib = 98944;
        const V rc = io.rc(r);
#pragma unroll
        for (int i = 0; i < 25; i++) io.arr(Ib + 1600 + 64 * i, ch[i]);
        io.arr(Ib + 3200, rc); io.arr(Ib + 3264, rc);
        const V o = io.gxo(0, ch[0], rc);
        garr(io, Ib + 3328, o, ch[0], rc);
        out[0] = o;
#pragma unroll
        for (int i = 1; i < 25; i++) out[i] = ch[i];
#pragma unroll
        for (int i = 0; i < 25; i++) io.arr(Ib + 64 * i, out[i]);
    }
#pragma unroll
    for (int i = 0; i < 25; i++) { io.arr(64 * i, out[i]); io.out(i, out[i]); }
}
// the alias table of a KeccakfRound block (KECCAKF_ROUND_WIRES codes): false if the walk did not name every wire exactly once
static inline bool keccak_round_alias_table(uint16_t* tab) {
    for (uint32_t i = 0; i < KECCAKF_ROUND_WIRES; i++) tab[i] = 0xFFFFu;
    SymIO io; io.tab = tab; io.ok = true;
    round_walk(io, 0);
    for (uint32_t i = 0; i < KECCAKF_ROUND_WIRES; i++) if (tab[i] == 0xFFFFu) io.ok = false;
    return io.ok;
}

// Keccakf block offsets (:356-367): out@0 in@1600 midRound[25]@3200 | KeccakfRound(r) @43200 + r*102656
#define KF_MID 3200u
#define KF_ROUNDS 43200u
// Absorb block offsets (:304-323): out@0 s@1600 block@3200 aux@4288 | XorArray x17 @5888 | Keccakf @12416
#define AB_KECCAKF (ABSORB_OWN + 17u * 384u)


// Sponge chain (Final/Absorb, keccak.circom:304-349): everything of Keccak(n)/Final(n)/Absorb x n EXCEPT the 24 round
// blocks and the output selector: Keccak.in, Final.in, Final.s[0..n], Absorb own wires + 17 XorArrays, Keccakf in/out/midRound.
// (<= 128 VGPRs: the small sponges of a side track must fit the slot a k_rounds wave frees, see g_gen_heavy_small.hip)
// (every kernel of this file is a BODY -- a device function of the launch arguments and the (item, group) of its wavefront -- plus a __global__ wrapper, so that a launch
//  can carry the wavefronts of several independent kernels: fused launches, g_*.hip)
// (Round 6 let the chain's evaluation ride with it like k_rounds_gc's -- every array loaded back behind its store and compared: the compiler requests a block's 400 arrays at once
//  (961 spilled VGPRs); with the compares pinned per group of ~25 loads the serial chain waits 14 more round trips per block: a lone generation 1.83 -> 3.38 ms, the loop with 12
//  in flight 1.11 -> 1.16 ms per step.  Not kept: k_chain_check stays a launch of the evaluation; profiles/round6_experiments.txt 16.)
__device__ __forceinline__ void chain_body(const KArgs A, uint32_t bx, uint32_t by) {
    const uint32_t lane = threadIdx.x & 63u;              // (a wavefront of a wider workgroup in the fused launch, g_gen_poswide.hip)
    const SpongeDesc sp = A.sponges[A.first + bx];
    u64* G = A.bits + (uint64_t)by * A.group_stride;
    u64 st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    auto put = [&](uint32_t idx, u64 v) { G[idx + lane] = v; };
    for (uint32_t b = 0; b < sp.n; b++) {
        const uint32_t Ab = sp.abs_b + b * ABSORB_BITS;
        u64 blk[17], aux[25];
#pragma unroll
        for (int i = 0; i < 17; i++) blk[i] = G[sp.src_b + b * 1088 + 64 * i + lane];
#pragma unroll
        for (int i = 0; i < 25; i++) { put(sp.fs_b + b * 1600 + 64 * i, st[i]); put(Ab + 1600 + 64 * i, st[i]); }
#pragma unroll
        for (int i = 0; i < 17; i++) {
            put(sp.kin_b + b * 1088 + 64 * i, blk[i]); put(sp.fin_b + b * 1088 + 64 * i, blk[i]); put(Ab + 3200 + 64 * i, blk[i]);
            const u64 o = st[i] ^ blk[i];
            const uint32_t X = Ab + ABSORB_OWN + 384 * i;
            put(X, o); put(X + 64, st[i]); put(X + 128, blk[i]);
            { u64* q = G + X + 192 + 3 * lane; q[0] = o; q[1] = st[i]; q[2] = blk[i]; }
            aux[i] = o;
        }
#pragma unroll
        for (int i = 17; i < 25; i++) aux[i] = st[i];
        const uint32_t Kf = Ab + AB_KECCAKF;
#pragma unroll
        for (int i = 0; i < 25; i++) { put(Ab + 4288 + 64 * i, aux[i]); put(Kf + 1600 + 64 * i, aux[i]); put(Kf + KF_MID + 64 * i, aux[i]); st[i] = aux[i]; }
        for (int r = 0; r < 24; r++) {
            round_native(st, r, lane);
#pragma unroll
            for (int i = 0; i < 25; i++) put(Kf + KF_MID + 1600 * (r + 1) + 64 * i, st[i]);
        }
#pragma unroll
        for (int i = 0; i < 25; i++) { put(Kf + 64 * i, st[i]); put(Ab + 64 * i, st[i]); }
    }
#pragma unroll
    for (int i = 0; i < 25; i++) put(sp.fs_b + sp.n * 1600 + 64 * i, st[i]);
}
#ifdef POB_KECCAK_TU          // (the non-template kernels are defined in ONE translation unit, k_keccak.hip; the bodies and the template kernels wherever they are used)
__global__ void __launch_bounds__(64, 3) k_chain(KArgs A) { chain_body(A, blockIdx.x, blockIdx.y); }
#endif

// (round 5 tried TWO wavefronts per (group, sponge), each on the 32-bit half of every word -- 32-bit logic, one ds_bpermute per rotation: the header's 17-block chain
//  0.433 -> 0.336 ms alone, but the step went from 1.57-1.60 / 1.39 to 1.67 / 1.52-1.53 ms with 4 / 8 in flight: twice the wavefronts storing 4-byte halves of every
//  8-byte word is twice the store transactions on a write path the round expansion already saturates; profiles/round5_experiments.txt 8.  Not kept.)
// Constraint evaluation of the same wires, LOCAL per permutation (every relation of Absorb/Final/Keccakf's own wires is
// between stored wires, so no permutation needs to be recomputed): grid.x = (sponge, block), grid.y = group.
__device__ __forceinline__ void chain_check_body(const KArgs A, uint32_t bx, uint32_t by) {
    const uint32_t lane = threadIdx.x;
    const uint32_t pi = A.first + bx;
    const SpongeDesc sp = A.sponges[A.perm_sponge[pi]];
    const uint32_t b = A.perm_block[pi];
    const u64* G = A.bits + (uint64_t)by * A.group_stride;
    const uint32_t Ab = sp.abs_b + b * ABSORB_BITS, Kf = Ab + AB_KECCAKF;
    u64 bad = 0;
    // (one state word's ~17 loads per iteration: unrolled, all 400 loads are hoisted -- 256 VGPRs, or 556 B of scratch at 128)
#pragma unroll 1
    for (int i = 0; i < 25; i++) {
        const u64 st = G[sp.fs_b + b * 1600 + 64 * i + lane];                    // Final.s[b]
        if (b == 0) bad |= st;                                                     // s[0] <== 0  (:337-341)
        bad |= G[Ab + 1600 + 64 * i + lane] ^ st;                                  // Absorb.s
        u64 aux = st;
        if (i < 17) {
            const u64 blk = G[sp.src_b + b * 1088 + 64 * i + lane];                // KeccakBytes.inBlocks
            bad |= (G[sp.kin_b + b * 1088 + 64 * i + lane] ^ blk) | (G[sp.fin_b + b * 1088 + 64 * i + lane] ^ blk) | (G[Ab + 3200 + 64 * i + lane] ^ blk);
            const u64 o = st ^ blk;
            const uint32_t X = Ab + ABSORB_OWN + 384 * i;
            const u64* q = G + X + 192 + 3 * lane;
            bad |= (G[X + lane] ^ o) | (G[X + 64 + lane] ^ st) | (G[X + 128 + lane] ^ blk) | (q[0] ^ o) | (q[1] ^ st) | (q[2] ^ blk);
            aux = o;
        }
        bad |= (G[Ab + 4288 + 64 * i + lane] ^ aux) | (G[Kf + 1600 + 64 * i + lane] ^ aux) | (G[Kf + KF_MID + 64 * i + lane] ^ aux);
        const u64 last = G[Kf + KF_MID + 1600 * 24 + 64 * i + lane];               // midRound[24]
        bad |= (G[Kf + 64 * i + lane] ^ last) | (G[Ab + 64 * i + lane] ^ last) | (G[sp.fs_b + (b + 1) * 1600 + 64 * i + lane] ^ last);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { bad |= ((u64)__shfl_xor((uint32_t)(bad >> 32), o, 64) << 32) | __shfl_xor((uint32_t)bad, o, 64); }
    if ((bad >> lane) & 1) atomicMin(&A.bad_wire[by * 64 + lane], sp.abs_w + b * ABSORB_WIRES);
}
#ifdef POB_KECCAK_TU
__global__ void __launch_bounds__(64) POB_WAVES_PER_SIMD(4) k_chain_check(KArgs A) { chain_check_body(A, blockIdx.x, blockIdx.y); }
#endif

// Generation: one wavefront per (permutation, KR consecutive rounds, group).  Reads midRound[r0] (written by k_chain), writes the 76 gate-output arrays of
// each of its rounds (38.9 KB per 64 witnesses and round; rounds 1-3 stored all 102 656 wires of the block: 821 KB).  grid = ((24 / KR) x permutations, groups)
template <int KR> __device__ __forceinline__ void rounds_gen_body(const KArgs A, uint32_t x, uint32_t y) {
    static_assert(24 % KR == 0, "a chunk does not straddle two permutations");
    const uint32_t lane = threadIdx.x;
    const uint32_t pi = A.first + x / (24 / KR), r0 = x % (24 / KR) * KR;
    const SpongeDesc sp = A.sponges[A.perm_sponge[pi]];
    const uint32_t Ab = sp.abs_b + A.perm_block[pi] * ABSORB_BITS;
    u64* G = A.bits + (uint64_t)y * A.group_stride;
    GenIO io; io.lane = lane;
    const u64* mid = G + Ab + AB_KECCAKF + KF_MID + 1600 * r0;
#pragma unroll
    for (int i = 0; i < 25; i++) io.s[i] = mid[64 * i + lane];
    io.st = G + Ab + AB_DIRECT + r0 * KR_BITS;
#pragma unroll 1
    for (uint32_t r = r0; r < r0 + KR; r++) {
        io.refresh();
        round_walk(io, (int)r);
        io.st += KR_BITS;
    }
}
template <int KR> __global__ void __launch_bounds__(64) k_rounds_gen(KArgs A) { rounds_gen_body<KR>(A, blockIdx.x, blockIdx.y); }

// Constraint evaluation: one wavefront per (permutation, KR consecutive rounds, group).  It loads midRound[r0] once, then per round the 76 stored
// gate outputs and the stored midRound[r+1] (101 arrays = 51 712 B per 64 witnesses), checks every XOR / AND gate of the round on stored operands and
// keeps the verified midRound[r+1] in registers as the next round's input: (101 KR + 25) / KR arrays per round, each resident array of the chunk
// fetched once (round 4's one-round items fetched every state twice: 126 arrays per round).
#define KR_CHECK_ARRAYS(kr) (101u * (kr) + 25u)
#define POB_KCHK_ROUNDS 4
#define POB_KGEN_ROUNDS 8
// (the body is a MACRO, not a device function like the other kernels': inlined from a function -- by value, by reference, with or without lifetime markers -- the same
//  statements compile to 128 VGPRs + 112 spilled where the kernel's own scope gives 127 and no scratch; the ring of prefetched arrays is that sensitive to the allocator)
#define POB_ROUNDS_CHECK_BODY(A, X_, Y_, NT, KR, DP) do { \
    static_assert(24 % (KR) == 0, "a chunk does not straddle two permutations"); \
    __builtin_amdgcn_s_setprio(3);       /* beside the other calculators' kernels its loads issue first (experiment 15: 5-10 % less time in the step, the step itself unchanged) */ \
    const uint32_t lane = threadIdx.x; \
    const uint32_t pi = (A).first + (X_) / (24 / (KR)), r0 = (X_) % (24 / (KR)) * (KR); \
    const SpongeDesc sp = (A).sponges[(A).perm_sponge[pi]]; \
    const uint32_t Ab = sp.abs_b + (A).perm_block[pi] * ABSORB_BITS; \
    const u64* G = (A).bits + (uint64_t)(Y_) * (A).group_stride; \
    CheckIOT<NT, DP> io; io.lane = lane; io.bad = 0; \
    const u64* mid = G + Ab + AB_KECCAKF + KF_MID + 1600 * r0; \
    _Pragma("unroll") \
    for (int i = 0; i < 25; i++) io.s[i] = io.ldw(mid + 64 * i + lane); \
    io.st = G + Ab + AB_DIRECT + r0 * KR_BITS; \
    _Pragma("unroll 1") \
    for (uint32_t r = r0; r < r0 + (KR); r++) { \
        mid += 1600; io.out_ = mid; \
        io.begin_round(); \
        round_walk(io, (int)r); \
        io.st += KR_BITS; \
        if (__any(io.bad != 0)) {       /* (corrupted vectors only) which witnesses, and the round block the mismatch belongs to */ \
            u64 bad = io.bad; \
            _Pragma("unroll") \
            for (int o = 32; o > 0; o >>= 1) { bad |= ((u64)__shfl_xor((uint32_t)(bad >> 32), o, 64) << 32) | __shfl_xor((uint32_t)bad, o, 64); } \
            if ((bad >> lane) & 1) atomicMin(&(A).bad_wire[(Y_) * 64 + lane], sp.abs_w + (A).perm_block[pi] * ABSORB_WIRES + AB_KECCAKF + KF_ROUNDS + r * KECCAKF_ROUND_WIRES); \
            io.bad = 0; \
        } \
    } } while (0)
template <bool NT, int KR, int WAVES, int DP> __global__ void __launch_bounds__(64) POB_WAVES_PER_SIMD(WAVES) k_rounds_check(KArgs A) { POB_ROUNDS_CHECK_BODY(A, blockIdx.x, blockIdx.y, NT, KR, DP); }

// Generation AND constraint evaluation of the round blocks in one launch (in-order calculators, pob_set_inorder bit 2).  The wavefront of (permutation, KR rounds, group)
// stores each of the 76 gate-output arrays of a round and requests the SAME array back from memory right behind the store; DP gates later the loaded array is compared with
// the value the gate's definition gives.  midRound[r+1], which k_chain wrote, is loaded and compared with the round's computed output the same way.  The footprint between a
// store and its load is DP x 512 B per wavefront -- it stays in the XCD's L2 (write-back, write-allocate), so the evaluation's 1.77 GB per 1 024 production witnesses never
// come back from HBM: what the launch moves there is the expansion's 1.26 GB of stores and the 0.44 GB of stored states.
// What is evaluated: for every stored array X of the block, loaded(X) == def(X) on the COMPUTED operands, starting from the LOADED midRound[r0].  By induction over the walk
// every operand's loaded value equals its computed one, so every XOR / AND gate of the round holds between the stored wires -- the relations k_rounds_check evaluates on
// loaded operands; a stored word that differs from its definition is reported at the same wire (the round block's first).  The load's base pointer takes an opaque zero
// offset, so no load can be replaced by the value stored before it: the ISA holds 76 stores and 101 loads per round (tools/isa_counts.py).
// (First version, profiles/round6_experiments.txt 14: generate a whole round, then k_rounds_check's walk over it with the input state parked in LDS -- 38.9 KB between store
//  and load per wavefront, 10 MB per XCD: the loads missed L2 and the launch took as long as the two kernels it replaced, 0.50 ms.)
// FAULT (tests only): the store at BIT word A.fault_word of group A.fault_group is XORed with A.fault_mask on its way to memory while the walk goes on with the right
// value -- what the evaluation must then flag.
template <int DP, bool FAULT, bool NTM> struct GcIOT : DevIOBase {
    u64* st; const u64* ld; const u64* out_; u64 bad; V s[25];
    V ring[DP], pend[DP];                      // the loaded array of gate q and the value its definition gave, q % DP
    const u64* fault; u64 fault_mask;
    __device__ __forceinline__ V in(int i) const { return s[i]; }
    template <bool STORE> __device__ __forceinline__ V twin(int want, V v) {
        const int q = kchk_pos(want), k = q % DP;
        if (q >= DP) bad |= ring[k] ^ pend[k];
        if constexpr (STORE) {
            u64* p = st + want + lane;
            V w = v;
            if constexpr (FAULT) { if (p == fault) w ^= fault_mask; }
            *p = w;
            ring[k] = ld[want + lane];
        } else ring[k] = NTM ? __builtin_nontemporal_load(out_ + (-want - 1) + lane) : out_[(-want - 1) + lane];
        pend[k] = v;
        __builtin_amdgcn_sched_barrier(0);
        return v;
    }
    __device__ __forceinline__ V gx(uint32_t sl, V a, V b) { return twin<true>(64 * (int)sl, a ^ b); }
    __device__ __forceinline__ V ga(uint32_t sl, V a, V b) { return twin<true>(64 * (int)sl, a & b); }
    __device__ __forceinline__ V gxo(int i, V a, V b) { return twin<false>(-(64 * i) - 1, a ^ b); }
    __device__ __forceinline__ void out(int i, V v) { s[i] = v; }
    __device__ __forceinline__ void end_round() {       // the last DP gates of the walk
#pragma unroll
        for (int k = 0; k < DP; k++) bad |= ring[k] ^ pend[k];
    }
};
template <int KR, int DP, int WAVES, bool FAULT, bool NTM> __global__ void __launch_bounds__(64) POB_WAVES_PER_SIMD(WAVES) k_rounds_gc(KArgs A) {
    static_assert(24 % KR == 0, "a chunk does not straddle two permutations");
    static_assert(DP >= 1 && DP <= KCHK_LOADS, "loads in flight");
    const uint32_t lane = threadIdx.x;
    const uint32_t pi = A.first + blockIdx.x / (24 / KR), r0 = blockIdx.x % (24 / KR) * KR;
    const SpongeDesc sp = A.sponges[A.perm_sponge[pi]];
    const uint32_t Ab = sp.abs_b + A.perm_block[pi] * ABSORB_BITS;
    u64* G = A.bits + (uint64_t)blockIdx.y * A.group_stride;
    const u64* mid = G + Ab + AB_KECCAKF + KF_MID + 1600 * r0;
    GcIOT<DP, FAULT, NTM> io; io.lane = lane; io.bad = 0;
    if constexpr (FAULT) { io.fault = blockIdx.y == A.fault_group ? G + A.fault_word : nullptr; io.fault_mask = A.fault_mask; }
#pragma unroll
    for (int i = 0; i < 25; i++) io.s[i] = mid[64 * i + lane];
    io.st = G + Ab + AB_DIRECT + r0 * KR_BITS;
#pragma unroll 1
    for (uint32_t r = r0; r < r0 + KR; r++) {
        uint32_t zero = 0;
        POB_OPAQUE_S(zero);                    // (an opaque OFFSET: the pointer stays a global-memory one, but is no longer provably the one stored through)
        io.ld = io.st + zero; mid += 1600; io.out_ = mid;
        io.refresh();
        round_walk(io, (int)r);
        io.end_round();
        io.st += KR_BITS;
        if (__any(io.bad != 0)) {              // (corrupted stores only) which witnesses, and the round block the mismatch belongs to
            u64 bad = io.bad;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { bad |= ((u64)__shfl_xor((uint32_t)(bad >> 32), o, 64) << 32) | __shfl_xor((uint32_t)bad, o, 64); }
            if ((bad >> lane) & 1) atomicMin(&A.bad_wire[blockIdx.y * 64 + lane], sp.abs_w + A.perm_block[pi] * ABSORB_WIRES + AB_KECCAKF + KF_ROUNDS + r * KECCAKF_ROUND_WIRES);
            io.bad = 0;
        }
    }
}

// the 64-witness word of the wire at offset o of an Absorb block whose storage starts at BIT rank ab (A = this group's slab); *neg: the
// wire is the complement of that word
__device__ __forceinline__ u64 absorb_wire_word(const u64* A, uint32_t ab, uint32_t o, const uint16_t* tab, uint32_t* neg) {
    *neg = 0;
    if (o < AB_DIRECT) return A[ab + o];
    const uint32_t q = o - AB_DIRECT, r = q / KECCAKF_ROUND_WIRES, e = tab[q - r * KECCAKF_ROUND_WIRES];
    const uint32_t slot = e >> 7, k = (e >> 1) & 63u;
    *neg = e & 1u;
    if (slot < KS_ST) return A[ab + AB_KECCAKF + KF_MID + 1600 * r + 64 * slot + k];
    if (slot < KS_OUT) return A[ab + AB_DIRECT + r * KR_BITS + 64 * (slot - KS_ST) + k];
    if (slot < KS_ZERO) return A[ab + AB_KECCAKF + KF_MID + 1600 * (r + 1) + 64 * (slot - KS_OUT) + k];
    if (slot == KS_ZERO) return 0;
    return ((KECCAK_RC_DEV[r] >> k) & 1) ? ~0ULL : 0ULL;
}
#ifdef POB_KECCAK_TU
// .wtns expansion of the wires [o0, o0 + count) of ONE Absorb block for witness `sel` of a group: stored wires directly, alias wires through
// the table (keccak_round_alias_table)
__global__ void __launch_bounds__(256) k_emit_absorb(const u64* G, uint8_t* out, uint32_t ab, uint32_t o0, uint32_t count, uint32_t sel, const uint16_t* tab) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
        uint32_t neg;
        const u64 word = absorb_wire_word(G, ab, o0 + t, tab, &neg);
        const uint32_t v = (uint32_t)((word >> sel) & 1) ^ neg;
        uint4* q = (uint4*)(out + (uint64_t)t * 32);
        q[0] = make_uint4(v, 0, 0, 0); q[1] = make_uint4(0, 0, 0, 0);
    }
}
// the same for the reduced witness: wires [wire0, wire0 + count) = offsets o0.. of the block, kept wires land at their rank
__global__ void __launch_bounds__(256) k_emit_absorb_red(const u64* G, uint8_t* out, uint32_t wire0, uint32_t ab, uint32_t o0, uint32_t count, uint32_t sel, const uint16_t* tab,
                                                         const unsigned long long* rbits, const uint32_t* rpre, uint32_t k0, uint32_t kn) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
        const uint32_t w = wire0 + t;
        const unsigned long long word_k = rbits[w >> 6];
        if (!((word_k >> (w & 63)) & 1)) continue;
        const uint32_t p = rpre[w >> 6] + (uint32_t)__popcll(word_k & ((1ull << (w & 63)) - 1)) - k0;
        if (p >= kn) continue;
        uint32_t neg;
        const u64 word = absorb_wire_word(G, ab, o0 + t, tab, &neg);
        const uint32_t v = (uint32_t)((word >> sel) & 1) ^ neg;
        uint4* q = (uint4*)(out + (uint64_t)p * 32);
        q[0] = make_uint4(v, 0, 0, 0); q[1] = make_uint4(0, 0, 0, 0);
    }
}

// .wtns expansion of a contiguous run of BIT wires for witness `sel` of one group: 8 B in, 32 B out per wire.
__global__ void __launch_bounds__(256) k_emit_bits(const u64* G, uint8_t* out, uint32_t wire_base, uint32_t bit_base, uint32_t count, uint32_t sel) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)((G[bit_base + t] >> sel) & 1);
        uint4* q = (uint4*)(out + (uint64_t)(wire_base + t) * 32);
        q[0] = make_uint4(v, 0, 0, 0); q[1] = make_uint4(0, 0, 0, 0);
    }
}
// the same for the reduced witness (policy.hpp EmitP::w32): thread per wire of the run, dropped wires cost one bitmap test
__global__ void __launch_bounds__(256) k_emit_bits_red(const u64* G, uint8_t* out, uint32_t wire0, uint32_t bit_base, uint32_t count, uint32_t sel,
                                                       const unsigned long long* rbits, const uint32_t* rpre, uint32_t k0, uint32_t kn) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
        const uint32_t w = wire0 + t;
        const unsigned long long word = rbits[w >> 6];
        if (!((word >> (w & 63)) & 1)) continue;
        const uint32_t p = rpre[w >> 6] + (uint32_t)__popcll(word & ((1ull << (w & 63)) - 1)) - k0;
        if (p >= kn) continue;
        const uint32_t v = (uint32_t)((G[bit_base + t] >> sel) & 1);
        uint4* q = (uint4*)(out + (uint64_t)p * 32);
        q[0] = make_uint4(v, 0, 0, 0); q[1] = make_uint4(0, 0, 0, 0);
    }
}
#endif  // POB_KECCAK_TU
