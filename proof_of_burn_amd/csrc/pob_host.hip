// libpob_hip.so -- host side: layout planner, stage scheduler and the C ABI of include/pob_hip.h.
// Build: proof_of_burn_amd/build.py (hipcc --offload-arch=gfx950, one object per .hip file, linked -shared).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <array>
#include <functional>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/pob_hip.h"
#include "kernels_common.hpp"


__global__ void k_init_invlut(uint32_t* lut) {   // canonical inverses of -4096..4096
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > 8192) return;
    Fr c = fr_from_mont(fr_inv(fr_from_i64((int64_t)t - 4096)));
    for (int j = 0; j < 8; j++) lut[t * 8 + j] = c.l[j];
}
// status/outputs of the batch: commitment FR (Montgomery) -> canonical LE; 0xFFFFFFFF -> 0 (ok)
// and the same as ONE record per witness {u32 status, u32 check_status, u32 bad_wire, u8 commitment[32]} (44 B), written twice: into device
// memory (the payload of the multi-GPU result gather) and straight into pinned host memory (what the host reads per batch: no copy, no
// copy stream).  Runs at the end of the generation (chk == null: verdict = POB_NOT_EVALUATED) and again at the end of the evaluation.
// The per-witness words the units reduce into with atomicMin (status_raw; chk / bad) are reset here, after they have been read, so that
// neither pass starts with a memset in front of its first kernel.
__global__ void k_collect(const uint32_t* fr, uint64_t fr_stride, uint32_t out_idx, uint32_t* status_raw, uint32_t* status, uint8_t* outputs, uint32_t* records,
                          uint32_t* host_records, uint32_t* chk, uint32_t* bad, uint32_t n) {
    uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const uint32_t g = w / 64, lane = w % 64;
    Fr m = fr_zero();
    if (out_idx != 0xFFFFFFFFu) for (int k = 0; k < 8; k++) m.l[k] = fr[(uint64_t)g * fr_stride + (uint64_t)out_idx * 512 + k * 64 + lane];
    Fr c = fr_from_mont(m);
    uint32_t s, cs = POB_NOT_EVALUATED, bw = POB_NOT_EVALUATED;
    if (!chk) {                                           // end of the generation
        s = status_raw[w]; status_raw[w] = 0xFFFFFFFFu;
        s = s == 0xFFFFFFFFu ? 0 : s;
        status[w] = s;
        for (int k = 0; k < 8; k++) ((uint32_t*)(outputs + (uint64_t)w * 32))[k] = c.l[k];
    } else {                                              // end of the evaluation
        s = status[w];
        cs = chk[w]; bw = bad[w];
        chk[w] = 0xFFFFFFFFu; bad[w] = 0xFFFFFFFFu;
    }
    uint32_t* rec = records + (uint64_t)w * (POB_RECORD_BYTES / 4);
    uint32_t* hrec = host_records + (uint64_t)w * (POB_RECORD_BYTES / 4);
    rec[0] = hrec[0] = s; rec[1] = hrec[1] = cs; rec[2] = hrec[2] = bw;
    for (int k = 0; k < 8; k++) rec[3 + k] = hrec[3 + k] = c.l[k];
}
// The main component's small inputs (proof_of_burn.circom:43-72: numLeafAddressNibbles, layers[][], layerLens[], numLayers, blockHeader[], ...):
// packed rows [witness][nsm] -> the input wires' SM rows [wire][64 witnesses], 64 x 64 tiles through LDS so that both sides are coalesced
// 256-byte accesses.  As lane = witness units (U_POB_INPUT) every lane read its own 43 KB-strided row: 0.09 ms alone, 1.8-2 ms beside the
// other batch's round expansion -- at the head of every narrow chain of the generation.  CHECK: the stored rows against the inputs
// (the relation `wire === input`, the evaluator's U_POB_INPUT).  grid = (ceil(nsm / 64), groups).
extern __shared__ uint32_t g_lds[];
// MODE 0: generation; 1: evaluation (the CHECK above); 2: generation whose evaluation rides with it (in-order calculators, pob_set_inorder bit 2): the rows are stored, LOADED
// back through a pointer the compiler cannot identify with the one stored through, and compared with the inputs
#define POB_OPAQUE_PTR(p, T) ({ uint32_t zero_ = 0; POB_OPAQUE_S(zero_); (T)(p) + zero_; })
// (fault_k / fault_g / fault_lanes, MODE 2, tests: input row fault_k of group fault_g reaches memory with bit 0 flipped for the witnesses of fault_lanes; 0xFFFFFFFF = none)
template <int MODE> __global__ void __launch_bounds__(64) k_inputs(const int32_t* in_sm, uint32_t nsm, int32_t* sm, uint64_t sm_stride, uint32_t s0, uint32_t w0, uint32_t* bad_wire,
                                                                   uint32_t fault_k, uint32_t fault_g, uint64_t fault_lanes) {
    constexpr bool CHECK = MODE == 1;
    int32_t* t = (int32_t*)g_lds;                         // [64 witnesses][65]
    const uint32_t lane = threadIdx.x, k0 = blockIdx.x * 64, g = blockIdx.y;
    const int32_t* src = in_sm + (uint64_t)g * 64 * nsm;
    for (uint32_t w = 0; w < 64; w++) t[w * 65 + lane] = (k0 + lane < nsm) ? src[(uint64_t)w * nsm + k0 + lane] : 0;
    __syncthreads();
    int32_t* dst = sm + (uint64_t)g * sm_stride + (uint64_t)(s0 + k0) * 64;
    const uint32_t nk = nsm - k0 < 64 ? nsm - k0 : 64;
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t kk = 0; kk < nk; kk++) {
        const int32_t v = t[lane * 65 + kk];
        if (CHECK) { if (dst[kk * 64 + lane] != v && bad == 0xFFFFFFFFu) bad = w0 + k0 + kk; }
        else dst[kk * 64 + lane] = (MODE == 2 && k0 + kk == fault_k && g == fault_g && ((fault_lanes >> lane) & 1)) ? v ^ 1 : v;
    }
    if (MODE == 2) {
        const int32_t* back = POB_OPAQUE_PTR(dst, const int32_t*);
        for (uint32_t kk = 0; kk < nk; kk++) if (back[kk * 64 + lane] != t[lane * 65 + kk] && bad == 0xFFFFFFFFu) bad = w0 + k0 + kk;
    }
    if (MODE) { if (bad != 0xFFFFFFFFu) atomicMin(&bad_wire[g * 64 + lane], bad); }
}
static void launch_inputs(pob_ctx* h, int mode, uint32_t G, hipStream_t st);
// The same from the BYTE FORM of a batch (pob_upload_inputs8*: u8 rows + POB_EXC_CAP exception slots per witness), in ONE pass: a workgroup reads 64 witnesses x 128 inputs as
// bytes (32 per thread), widens them into an LDS tile, lays the group's exception slots that fall into its 128 inputs over the tile, and writes (CHECK: compares) the 128 SM
// rows, lane = witness.  Rounds 4-5 widened the whole batch into an int32 copy of the packed inputs first (k_widen_sm8 + k_apply_exc on the upload stream: 11 MB read, 45 MB
// written) and transposed that copy (45 MB read, 45 MB written): per batch two launches and 90 MB of traffic that this pass does not have.  grid = (ceil(nsm / 128), groups).
#define IN8_K 128
template <int MODE> __global__ void __launch_bounds__(256) k_inputs8(const uint8_t* sm8, const pob_sm_exc_t* exc, uint32_t nsm, uint32_t n, int32_t* sm, uint64_t sm_stride, uint32_t s0,
                                                                      uint32_t w0, uint32_t* bad_wire, uint32_t fault_k, uint32_t fault_g, uint64_t fault_lanes) {
    constexpr bool CHECK = MODE == 1;
    int32_t* t = (int32_t*)g_lds;                         // [64 witnesses][IN8_K + 1]
    const uint32_t tid = threadIdx.x, k0 = blockIdx.x * IN8_K, g = blockIdx.y;
    {   // bytes: thread = (witness row, 32-byte segment)
        const uint32_t row = tid >> 2, seg = tid & 3u, wg = g * 64 + row, kb = k0 + 32 * seg;
        const uint8_t* src = sm8 + (uint64_t)wg * nsm + kb;
        int32_t* d = t + row * (IN8_K + 1) + 32 * seg;
        if (wg < n && kb + 32 <= nsm && (nsm & 3u) == 0) {
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) { const uint32_t w = ((const uint32_t*)src)[j]; d[4 * j] = (int32_t)(w & 255u); d[4 * j + 1] = (int32_t)((w >> 8) & 255u); d[4 * j + 2] = (int32_t)((w >> 16) & 255u); d[4 * j + 3] = (int32_t)(w >> 24); }
        } else {
            for (uint32_t j = 0; j < 32; j++) d[j] = (wg < n && kb + j < nsm) ? (int32_t)src[j] : 0;
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 64 * POB_EXC_CAP; i += 256) {          // the group's exception slots (values outside 0..255: the lengths, deliberately out-of-range test inputs)
        const uint32_t row = i / POB_EXC_CAP, wg = g * 64 + row;
        if (wg >= n) continue;
        const pob_sm_exc_t e = exc[(uint64_t)wg * POB_EXC_CAP + i % POB_EXC_CAP];
        if (e.k < nsm && e.k - k0 < IN8_K) t[row * (IN8_K + 1) + (e.k - k0)] = e.v;
    }
    __syncthreads();
    const uint32_t lane = tid & 63u, q = tid >> 6;
    int32_t* dst = sm + (uint64_t)g * sm_stride + (uint64_t)(s0 + k0) * 64;
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t kk = 32 * q; kk < 32 * q + 32 && k0 + kk < nsm; kk++) {
        const int32_t v = t[lane * (IN8_K + 1) + kk];
        if (CHECK) { if (dst[kk * 64 + lane] != v && bad == 0xFFFFFFFFu) bad = w0 + k0 + kk; }
        else dst[kk * 64 + lane] = (MODE == 2 && k0 + kk == fault_k && g == fault_g && ((fault_lanes >> lane) & 1)) ? v ^ 1 : v;
    }
    if (MODE == 2) {
        const int32_t* back = POB_OPAQUE_PTR(dst, const int32_t*);
        for (uint32_t kk = 32 * q; kk < 32 * q + 32 && k0 + kk < nsm; kk++) if (back[kk * 64 + lane] != t[lane * (IN8_K + 1) + kk] && bad == 0xFFFFFFFFu) bad = w0 + k0 + kk;
    }
    if (MODE) { if (bad != 0xFFFFFFFFu) atomicMin(&bad_wire[g * 64 + lane], bad); }
}
// pob_upload_inputs8*: the byte rows widened into the int32 rows every kernel reads (four values per thread when the row length allows aligned words), then the
// exception slots on top (values outside 0..255: lengths, deliberately out-of-range test inputs)
__global__ void __launch_bounds__(256) k_widen_sm8(const uint8_t* sm8, int32_t* sm, uint64_t total) {
    // 16 bytes per thread and step (one 16-byte load, four 16-byte stores), grid-stride: a few thousand wavefronts -- with four values per thread the pass was 174 k
    // wavefronts per batch that competed for wave slots with whatever ran beside the upload (the round evaluation of the previous batch: the host enqueues the two together)
    const uint64_t nvec = total / 16;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 w = ((const uint4*)sm8)[t];
        uint4* o = (uint4*)sm + 4 * t;
        o[0] = make_uint4(w.x & 255u, (w.x >> 8) & 255u, (w.x >> 16) & 255u, w.x >> 24);
        o[1] = make_uint4(w.y & 255u, (w.y >> 8) & 255u, (w.y >> 16) & 255u, w.y >> 24);
        o[2] = make_uint4(w.z & 255u, (w.z >> 8) & 255u, (w.z >> 16) & 255u, w.z >> 24);
        o[3] = make_uint4(w.w & 255u, (w.w >> 8) & 255u, (w.w >> 16) & 255u, w.w >> 24);
    }
    if (blockIdx.x == 0) for (uint64_t t = nvec * 16 + threadIdx.x; t < total; t += blockDim.x) sm[t] = (int32_t)sm8[t];      // (a row length that is no multiple of 16)
}
__global__ void __launch_bounds__(256) k_apply_exc(const pob_sm_exc_t* exc, int32_t* sm, uint32_t n, uint32_t nsm) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * POB_EXC_CAP) return;
    const pob_sm_exc_t e = exc[t];
    if (e.k < nsm) sm[(uint64_t)(t / POB_EXC_CAP) * nsm + e.k] = e.v;
}
// test hook (pob_debug_fr_inv): both field inversions of the device code on n canonical inputs
__global__ void k_fr_inv_test(const uint32_t* in, uint32_t* out_kaliski, uint32_t* out_fermat, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Fr c; for (int j = 0; j < 8; j++) c.l[j] = in[t * 8 + j];
    const Fr x = fr_to_mont(c);
    const Fr a = fr_is_zero(x) ? fr_zero() : fr_from_mont(fr_inv(x)), b = fr_is_zero(x) ? fr_zero() : fr_from_mont(fr_inv_fermat(x));
    for (int j = 0; j < 8; j++) { out_kaliski[t * 8 + j] = a.l[j]; out_fermat[t * 8 + j] = b.l[j]; }
}
// ---- emit-time self-check (pob_emit_selfcheck): the relations of the DERIVED wires, evaluated on the canonical values as written into the emission window
// (reference: circomlib comparators.circom IsZero `out <== -in*inv + 1; in*out === 0`, IsEqual `in[1] - in[0] ==> isz.in; isz.out ==> out`;
//  substring_check.circom:45-49 `M[i+1] <== M[i] + mainInput[i] * 256^i`).  One thread per site; sites whose wires are not all inside the window are counted as
// skipped by the host (IsZero / IsEqual) or here (M).  res[0] = lowest violated wire, res[1] = M sites skipped.
// where wire w lies in the window: O0 (rbits null) position w - w0; reduced witness: its rank among the kept wires - w0, false if the wire is dropped
struct ScWin { const uint8_t* win; uint32_t w0, wn; const unsigned long long* rbits; const uint32_t* rpre; };
__device__ __forceinline__ bool sc_pos(const ScWin& W, uint32_t w, uint32_t* pos) {
    if (!W.rbits) { *pos = w - W.w0; return *pos < W.wn; }
    const unsigned long long word = W.rbits[w >> 6];
    if (!((word >> (w & 63)) & 1)) return false;
    *pos = W.rpre[w >> 6] + (uint32_t)__popcll(word & ((1ull << (w & 63)) - 1)) - W.w0;
    return *pos < W.wn;
}
__device__ __forceinline__ Fr sc_load(const ScWin& W, uint32_t pos) {
    const uint32_t* q = (const uint32_t*)(W.win + (size_t)pos * 32);
    Fr c; for (int j = 0; j < 8; j++) c.l[j] = q[j];
    return fr_to_mont(c);
}
// (a site with a wire outside the window -- or, in the reduced witness, a dropped wire: the relation then lives between class representatives the keep map alone does
//  not name -- is skipped and counted in res[1])
__global__ void __launch_bounds__(64) k_selfcheck_z(ScWin W, const uint32_t* sites, uint32_t n, uint32_t* res) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t s = sites[t], w = s & 0x7FFFFFFFu;                  // IsZero [out | in | inv] at w
    uint32_t po, pi, pv, pe = 0, pa = 0, pb = 0;
    bool have = sc_pos(W, w, &po) && sc_pos(W, w + 1, &pi) && sc_pos(W, w + 2, &pv);
    if (have && (s >> 31)) have = sc_pos(W, w - 3, &pe) && sc_pos(W, w - 2, &pa) && sc_pos(W, w - 1, &pb);
    if (!have) { atomicAdd(res + 1, 1u); return; }
    const Fr out = sc_load(W, po), in = sc_load(W, pi), inv = sc_load(W, pv);
    bool ok = fr_eq(fr_mul(in, inv), fr_sub(fr_one_mont(), out)) && fr_is_zero(fr_mul(in, out));
    if (s >> 31) {                                                       // IsEqual [out | in[2]] at w - 3
        const Fr eo = sc_load(W, pe), a = sc_load(W, pa), b = sc_load(W, pb);
        ok = ok && fr_eq(in, fr_sub(b, a)) && fr_eq(eo, out);
    }
    if (!ok) atomicMin(res, w);
}
// copy constraints a === b between a derived wire and the stored wire it must equal (pairs {higher wire, lower wire})
__global__ void __launch_bounds__(64) k_selfcheck_c(ScWin W, const uint32_t* sites, uint32_t n, uint32_t* res) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t a = sites[2 * t], b = sites[2 * t + 1];
    uint32_t pa, pb;
    if (!sc_pos(W, a, &pa) || !sc_pos(W, b, &pb)) { atomicAdd(res + 1, 1u); return; }      // (the lower wire lies in the window before, or one of the two is dropped)
    const uint4* p = (const uint4*)(W.win + (size_t)pa * 32); const uint4* q = (const uint4*)(W.win + (size_t)pb * 32);
    const uint4 x0 = p[0], x1 = p[1], y0 = q[0], y1 = q[1];
    if (x0.x != y0.x || x0.y != y0.y || x0.z != y0.z || x0.w != y0.w || x1.x != y1.x || x1.y != y1.y || x1.z != y1.z || x1.w != y1.w) atomicMin(res, a);
}
__global__ void __launch_bounds__(64) k_selfcheck_m(ScWin W, const uint32_t* sites, uint32_t n, const uint32_t* pow256, uint32_t* res) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t wn1 = sites[3 * t], wb = sites[3 * t + 1], k = sites[3 * t + 2];
    uint32_t pn, pp, pby;
    if (!sc_pos(W, wn1, &pn) || !sc_pos(W, wn1 - 1, &pp) || !sc_pos(W, wb, &pby)) { atomicAdd(res + 1, 1u); return; }
    Fr pw; for (int j = 0; j < 8; j++) pw.l[j] = pow256[(size_t)k * 8 + j];                      // 256^k, Montgomery
    const Fr next = sc_load(W, pn), prev = sc_load(W, pp), by = sc_load(W, pby);
    if (!fr_eq(next, fr_add(prev, fr_mul(by, pw)))) atomicMin(res, wn1);
}
// the same relations on explicit wire lists (reduced witness): every wire of a site is kept (the host left the others out); a site is evaluated in the window that holds
// its first wire -- if the others lie there too (else skipped: res[1]); res[2] counts the sites evaluated
__global__ void __launch_bounds__(64) k_selfcheck_zr(ScWin W, const uint32_t* zw, uint32_t n, uint32_t* res) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t* w6 = zw + 6 * (size_t)t;                            // IsZero out, in, inv | IsEqual out, in[0], in[1] (0xFFFFFFFF: a bare IsZero)
    uint32_t p[6];                                                      // (an entry with bit 31 set: a wire pinned to the small constant in its low bits -- not in the window at all)
    if (!sc_pos(W, w6[0], &p[0])) return;
    const bool iseq = w6[3] != 0xFFFFFFFFu;
    auto pos = [&](int j) { if (w6[j] >> 31) { p[j] = w6[j]; return true; } return sc_pos(W, w6[j], &p[j]); };
    bool have = pos(1) && pos(2);
    if (have && iseq) have = pos(3) && pos(4) && pos(5);
    if (!have) { atomicAdd(res + 1, 1u); return; }
    atomicAdd(res + 2, 1u);
    auto val = [&](int j) { if (p[j] >> 31) { Fr c = fr_zero(); c.l[0] = p[j] & 0x7FFFFFFFu; return fr_to_mont(c); } return sc_load(W, p[j]); };
    const Fr out = val(0), in = val(1), inv = val(2);
    bool ok = fr_eq(fr_mul(in, inv), fr_sub(fr_one_mont(), out)) && fr_is_zero(fr_mul(in, out));
    if (iseq) { const Fr eo = val(3), a = val(4), b = val(5); ok = ok && fr_eq(in, fr_sub(b, a)) && fr_eq(eo, out); }
    if (!ok) atomicMin(res, w6[0]);
}
__global__ void __launch_bounds__(64) k_selfcheck_mr(ScWin W, const uint32_t* mw, uint32_t n, const uint32_t* pow256, uint32_t* res) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t* w4 = mw + 4 * (size_t)t;                            // M[k+1], M[k], mainInput[k], k
    uint32_t pn, pp, pby;
    if (!sc_pos(W, w4[0], &pn)) return;
    if (!sc_pos(W, w4[1], &pp) || !sc_pos(W, w4[2], &pby)) { atomicAdd(res + 1, 1u); return; }
    atomicAdd(res + 2, 1u);
    Fr pw; for (int j = 0; j < 8; j++) pw.l[j] = pow256[(size_t)w4[3] * 8 + j];
    const Fr next = sc_load(W, pn), prev = sc_load(W, pp), by = sc_load(W, pby);
    if (!fr_eq(next, fr_add(prev, fr_mul(by, pw)))) atomicMin(res, w4[0]);
}
__global__ void k_xor_word(uint64_t* p, uint64_t mask) { *p ^= mask; }
__global__ void k_xor_u32(uint32_t* p, uint32_t mask) { *p ^= mask; }
// emission window pre-fill (0xEE..: not a field element, so a wire nobody owns is caught by the byte compare); rocclr's fill kernel
// took 4.5 ms per 256 MiB window, this one runs at the HBM write rate
__global__ void __launch_bounds__(256) k_fill_ee(uint4* p, uint64_t n16) {
    const uint4 v = make_uint4(0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu, 0xEEEEEEEEu);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---- proof-of-work search (input producer, reference tests/main.py:47-56): thread t hashes key = start + t
__device__ __forceinline__ uint64_t pow_rotl(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
__global__ void __launch_bounds__(256) k_pow_search(uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3,       // start key, big-endian words
                                                     const uint64_t* __restrict__ tail,                       // lanes 4..16 of the padded block
                                                     uint64_t base, uint64_t count, uint64_t mask, unsigned long long* found) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t off = base + t;
    uint64_t w3 = k3 + off, c = w3 < k3, w2 = k2 + c; c = w2 < c; uint64_t w1 = k1 + c; c = w1 < c; const uint64_t w0 = k0 + c;
    uint64_t a[25];
    a[0] = __builtin_bswap64(w0); a[1] = __builtin_bswap64(w1); a[2] = __builtin_bswap64(w2); a[3] = __builtin_bswap64(w3);
#pragma unroll
    for (int i = 4; i < 17; i++) a[i] = tail[i - 4];
#pragma unroll
    for (int i = 17; i < 25; i++) a[i] = 0;
    const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
                             0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
                             0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
                             0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
                             0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        uint64_t cc[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) cc[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            const uint64_t d = cc[(x + 4) % 5] ^ pow_rotl(cc[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 25; y += 5) a[y + x] ^= d;
        }
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = pow_rotl(a[x + 5 * y], ROT[x + 5 * y]);
#pragma unroll
        for (int y = 0; y < 25; y += 5)
#pragma unroll
            for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
        a[0] ^= RC[r];
    }
    if ((a[0] & mask) == 0) atomicMin(found, (unsigned long long)off);
}

// ------------------------------------------------------------------------------------------------ host side
struct pob_ctx {
    int device = 0, circuit = 0;
    Plan plan;
    uint32_t max_batch = 0, groups = 0, n = 0;
    std::string err;
    hipStream_t stream = nullptr;
    // device memory
    uint64_t* d_bits = nullptr; int32_t* d_sm = nullptr; uint32_t* d_fr = nullptr;
    UnitDesc* d_units = nullptr; uint32_t* d_order = nullptr; CircuitLayout* d_L = nullptr;
    SpongeDesc* d_sponges = nullptr; uint32_t *d_perm_sponge = nullptr, *d_perm_block = nullptr;
    uint32_t *d_pos = nullptr, *d_inv = nullptr, *d_pow256 = nullptr; uint32_t npow256 = 0;
    uint32_t* d_emit_ctr = nullptr;                    // EmitP's path counters (pob_debug_emit_counters)
    uint16_t* d_ktab = nullptr;                        // alias table of a KeccakfRound block (keccak_kernels.hpp): which stored wire / constant each of its 102 656 wires is
    // packed inputs, double-buffered: a batch is uploaded into the buffer the current batch does NOT use (generation AND evaluation read the
    // inputs), pob_generate switches; ev_in_done[b] = the last generation / evaluation that read buffer b
    uint8_t* d_in_fr[2] = {nullptr, nullptr}; int32_t* d_in_sm[2] = {nullptr, nullptr}; int in_cur = 0, in_next = 0; uint32_t n_next = 0;
    hipEvent_t ev_in_done[2] = {nullptr, nullptr}; bool in_done_rec[2] = {false, false};
    uint8_t* d_in_sm8[2] = {nullptr, nullptr}; pob_sm_exc_t* d_in_exc[2] = {nullptr, nullptr};      // staging of the byte form (pob_upload_inputs8*): 11 KB per witness and buffer
    bool in_bytes[2] = {false, false};                  // buffer b holds its batch in the byte form only (ProofOfBurn: the inputs' wires are written straight from it, k_inputs8; d_in_sm[b] is stale)
    uint32_t *d_status_raw = nullptr, *d_status = nullptr, *d_chk = nullptr, *d_bad = nullptr, *d_records = nullptr; uint8_t* d_outputs = nullptr;
    // streaming .wtns emission: two device windows + two pinned host windows, window k+1 is expanded and copied while the caller
    // consumes window k (pob_emit_begin / pob_emit_next)
    struct Emit {
        // THREE window slots: one with the caller, one being copied, one being expanded -- so that the first window of the NEXT witness
        // (pob_emit_queue) is expanded while the last windows of the current one are still on their way to the host
        enum { NSLOT = 3 };
        uint8_t* d_win[NSLOT] = {nullptr, nullptr, nullptr}; uint8_t* h_pin[NSLOT] = {nullptr, nullptr, nullptr};
        hipStream_t s_copy = nullptr; hipEvent_t ev_made[NSLOT] = {nullptr, nullptr, nullptr}, ev_copied[NSLOT] = {nullptr, nullptr, nullptr}, ev_free[NSLOT] = {nullptr, nullptr, nullptr};
        uint64_t win_wires = 0, alloc_wires = 0, next_make = 0, next_take = 0, nwin = 0; uint32_t idx = 0; bool active = false;
        uint32_t first_slot = 0;                        // slot of the current witness' window 0 (window k: (first_slot + k) % NSLOT)
        int64_t queued_idx = -1; bool pre_made = false; // pob_emit_queue: the witness the next pob_emit_begin* will ask for / its window 0 is made
        uint64_t pre_made_gen = 0;                       // ... from the resident vector of THIS generation (h->gen_count when the window was expanded)
        std::vector<uint32_t> status_host; uint64_t status_gen = 0;      // the batch's generation statuses, fetched once per generation
        const uint32_t* pin_ptr = nullptr; uint64_t pin_n = 0, pin_id = 0;     // reduced: a map the caller pinned (pob_reduced_map_pin): contents promised unchanged, not hashed again
        struct Run { uint32_t w, b, n, absorb; };
        std::vector<Run> runs;                          // the Keccak kernels' wires, sorted by wire index: contiguous stored runs (wire index, BIT rank, count) and Absorb blocks (absorb = 1: stored + alias wires, b = the block's first BIT rank)
        // which G units write into which window (found by one probe pass per window size): a window launches only those
        uint64_t probe_win = 0, probe_map = 0; uint32_t* d_order = nullptr; unsigned long long* d_probe = nullptr;
        struct WSeg { uint32_t cls, first, count; };
        std::vector<std::vector<WSeg>> wsegs;
        // reduced witness (pob_emit_begin_reduced): the kept O0 wire indices (host copy for the run intersections), their bitmap and the
        // per-word rank on the device; map_id = 0: O0 payload.  total = wires of the payload being emitted (W or the kept count)
        // self-check (pob_emit_selfcheck): site tables (sorted by wire), built once per handle by a recording pass of the emitter; d_sc_res: {lowest violated wire, M sites skipped}
        bool sc_on = false, sc_built = false; std::vector<uint32_t> sc_z, sc_m_next, sc_c_hi; uint32_t *d_sc_z = nullptr, *d_sc_m = nullptr, *d_sc_c = nullptr, *d_sc_res = nullptr;
        uint64_t sc_checked = 0, sc_skipped = 0;
        // reduced witness: the sites as explicit wire lists, every wire replaced by its class representative (pob_emit_selfcheck_alias) and the sites with a wire that is
        // not kept left out (counted in sc_red_host_skipped); built per map
        std::vector<std::array<uint32_t, 3>> sc_ms; std::vector<std::array<uint32_t, 2>> sc_cs;
        const int32_t* sc_alias = nullptr; uint64_t sc_alias_n = 0, sc_red_map = 0, sc_red_host_skipped = 0; uint32_t sc_red_nz = 0, sc_red_nm = 0;
        uint32_t *d_sc_zr = nullptr, *d_sc_mr = nullptr;
        bool red = false; uint64_t map_id = 0, total = 0; std::vector<uint32_t> keep; unsigned long long* d_rbits = nullptr; uint32_t* d_rpre = nullptr;
    } em;
    // schedule
    std::vector<uint32_t> order;                       // unit indices grouped by (stage, lds flag)
    struct Seg { uint32_t stage, lds, first, count; };
    std::vector<Seg> segs, emit_segs, chk_segs;        // per (stage, class) for generation; per class for emission; per FAMILY for evaluation
    hipStream_t stream2 = nullptr, stream3 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    struct KSeg { uint32_t stage, sp_first, sp_count, perm_first, perm_count; hipEvent_t ev_done; };
    std::vector<KSeg> ksegs;                           // ev_done: recorded behind the segment's sponge kernels in pob_generate
    hipStream_t stream_k = nullptr;                                     // the main track's round expansion in pipeline mode (pob_generate)
    hipEvent_t ev_rounds_fork = nullptr, ev_g_done = nullptr, ev_k_done = nullptr;   // pipeline: the G side of a generation is enqueued / the Keccak evaluation on the streaming stream is done
    // side tracks (Plan::track_fork/track_join): streams of the device's pool (StreamPool below), own fork/join events, start and end events
    // (ROCm multiplexes streams onto few hardware queues: a lone handle keeps to the caller's stream + 4 of the pool's --
    //  stream2, track 1's (also track 6), track 2's (also track 3, which runs before it anyway), the round expansion's; a track's BN254 and light
    //  launches of one stage share its stream)
    struct Track { hipStream_t s_main = nullptr, s_heavy = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_start = nullptr, ev_end = nullptr; };
    Track tracks[Plan::MAX_TRACKS];
    uint32_t nperms = 0;
    // IN-ORDER schedule (pob_set_inorder): the whole generation / evaluation of the calculator on the CALLER's stream, no side stream and no event; a job gets
    // its concurrency from keeping several such calculators in flight, each on a stream of its own.  The stage graph is flattened into LEVELS (longest
    // dependency path): every unit of a level, of whatever track, goes out in one launch per kernel class, then the level's sponges; the round expansion of
    // every sponge is ONE launch at the end (nothing of the generation reads it).
    bool inorder = false;
    // FUSED launch (pob_set_inorder(h, 3), what bench.py runs): the lane-spread Poseidon blocks share a launch with the sponge chain that does not depend on them (header and
    // layers: the longest chain) -- two launches of a few hundred wavefronts each, both a long serial chain per wavefront, neither near any throughput limit: side by side
    // they take the longer one's time.  gen_plan[0]: one launch per kernel (round 5), [1]: with the fused launch.
    // (Round 6 also fused the bandwidth-bound kernels with latency-bound ones -- slices of the round expansion with the levels behind the chain, the round evaluation
    //  interleaved with the wide evaluation families, the chain evaluation behind the narrow families: such a launch lasts about the SUM of its parts (a latency-bound
    //  wavefront makes next to no progress while the memory system is saturated), the loop did not move; removed.  profiles/round6_experiments.txt)
    bool fused = false;
    // EVALUATION THAT RIDES WITH THE GENERATION (pob_set_inorder bit 2): the generation's last launch is k_rounds_gc -- every round block is evaluated by the wavefront that has
    // just written it, its loads served by L2 -- and the input rows are compared with the inputs by the launch that writes them (k_inputs MODE 2); the evaluation that follows
    // skips those two kernels.  rode: the resident vector is the one those launches evaluated (cleared by every debug poke: the evaluation then runs k_rounds_check and the
    // input check over the vector as it is)
    bool gc = false, rode = false;
    // ... and so does the evaluation of everything else but the sponge chains (the two main circuits: every G unit's stores are loaded back and compared, policy.hpp GenPT<true>,
    // poseidon_wide.hpp PosWideT<true>): pob_constraint_check then runs k_chain_check and collects the records
    bool rode_g = false;
    bool fault_armed = false, fault_g = false; int fault_cls = 0; uint32_t fault_group = 0; uint64_t fault_word = 0, fault_mask = 0;     // pob_debug_store_fault (tests)
    struct GenLaunch { uint32_t kind, cls, first, count, k_first, k_count; };
    enum { GL_UNITS = 0, GL_CHAIN = 1, GL_POS_CHAIN = 2, GL_ROUNDS = 4 };
    std::vector<GenLaunch> gen_plan[2];
    Seg chk_narrow{0, 0, 0, 0};                          // in-order evaluation: the units of the four narrow families, one launch
    Seg chk_ride_kept{0, F_RL, 0, 0};                    // ... when the G units' evaluation rode with their generation: the units whose generation did not ride (circuits.hpp ride_keeps_evaluation)
    uint32_t nlevels = 0;
    bool generated = false; uint64_t gen_count = 0;
    // two-batch pipeline (pob_set_partner): this handle's generation starts with the partner's evaluation and its Keccak expansion
    // waits for the end of that evaluation; the evaluation then uses the pool's two evaluation streams
    pob_ctx* partner = nullptr; struct StreamPool* pool = nullptr;
    hipEvent_t ev_gen_done = nullptr, ev_check_done = nullptr; bool gen_done_rec = false, check_done_rec = false, evaluated = false;
    hipStream_t gen_stream = nullptr, chk_stream = nullptr; bool gen_ordered = false, chk_ordered = false;      // where the last generation / evaluation was enqueued (a call on ANOTHER stream is ordered behind it by its event)
    // service loop: asynchronous input upload (own stream, pob_upload_inputs_async) and per-batch result records into pinned memory
    hipStream_t s_upload = nullptr;                         // = the pool's upload stream
    hipEvent_t ev_upload = nullptr, ev_rec[2] = {nullptr, nullptr};   // ev_rec[s]: the records of buffer s are written
    // timing events around the Keccak round evaluation (pob_probe_check_kernel): one pair per record slot -- the kernel of the batch whose records sit in slot s -- because the
    // evaluation that rides with the expansion records them in pob_generate, and the caller reads the previous batch's pair after it has enqueued the next generation
    bool kchk_rec[2] = {false, false}; hipEvent_t ev_kchk[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint8_t* h_records[2] = {nullptr, nullptr}; int rec_slot = 0, fetch_slot = 0; uint32_t rec_n[2] = {0, 0}; bool upload_pending = false, have_next = false, fetch_pending = false;
};

static hipStream_t own_stream(pob_ctx* h) {
    if (!h->stream) {       // (high priority like the side tracks: when the caller passes no stream this one carries the generation's main track)
        int lo = 0, hi = 0;
        hipSetDevice(h->device);
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) hi = 0;
        if (hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, hi) != hipSuccess) h->stream = nullptr;
    }
    return h->stream;
}

// The side streams are shared by every handle of a device (a handle's launches on them are ordered by events anyway): ROCm multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues and two handles with six streams each fall off that cliff (two calculators in
// flight ran at 1/30 of the speed).  chk1/chk2 are created on first use (pipeline mode only).
struct StreamPool { int device = 0, refs = 0; hipStream_t stream2 = nullptr, stream_k = nullptr, track1 = nullptr, track2 = nullptr, chk1 = nullptr, chk2 = nullptr, s_in = nullptr; };
static std::mutex g_pool_mu;
static std::vector<StreamPool*> g_pools;

// hardware queues the HIP runtime was (most likely) initialised with: the variable is read once, at runtime initialisation
#define POB_PIPELINE_HW_QUEUES 12
static int hw_queues_env() { const char* e = getenv("GPU_MAX_HW_QUEUES"); return e ? atoi(e) : 4; }
// loaded before the HIP runtime initialises (the usual case for a ctypes / cgo caller): ask for enough queues unless the caller decided otherwise
__attribute__((constructor)) static void pob_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

struct pob_ctx;
static hipStream_t own_stream(pob_ctx* h);      // the handle's own stream (callers that pass no stream, emission, test hooks): created on first use

#define HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); return POB_E_HIP; } } while (0)

// generation scheduling class: 0 = light, 1 = BN254 (byte conversions, range checks, the composites around the Poseidon blocks),
// 3 = SubstringCheck BN254, 4 = Poseidon blocks with the state spread over lanes (poseidon_wide.hpp)
// 5 = gadget-level mains (gadget_mains.hpp)
#define N_GEN_CLASSES 6
#define CLS_ALL 6          // in-order calculators: classes 0, 1 and 3 of a level in one launch (g_gen_all.hip)
static uint32_t unit_class(uint32_t kind, bool gen = false) { return fam_of(kind) == F_GM ? 5 : kind == U_POS_WIDE ? 4 : fam_of(kind) == F_SC ? 3 : (gen ? unit_gen_is_heavy(kind) : unit_is_heavy(kind)) ? 1 : 0; }
static void launch_g_gen(const GArgs& A, uint32_t cls, uint32_t nunits, uint32_t ngroups, hipStream_t st, bool ride = false, bool fault = false) {      // ride: the units' evaluation rides with them (policy.hpp GenPT<true>)
    if (cls == 5) launch_g_gen_gm(A, nunits, ngroups, st);
    else if (cls == 4) launch_pos_wide(A, nunits, ngroups, st, ride, fault);
    else if (cls == CLS_ALL && ride && fault) launch_g_gen_all_ride_fault(A, nunits, ngroups, st);
    else if (cls == CLS_ALL && ride) launch_g_gen_all_ride(A, nunits, ngroups, st);
    else if (cls == 3) launch_g_gen_sc(A, nunits, ngroups, st);
    else if (cls == 1) launch_g_gen_n2b(A, nunits, ngroups, st);
    else if (cls == CLS_ALL) launch_g_gen_all(A, nunits, ngroups, st);
    else launch_g_gen_light(A, nunits, ngroups, st);
}
static void launch_g_check(const GArgs& A, uint32_t fam, uint32_t nunits, uint32_t ngroups, hipStream_t st) {
    switch (fam) {
    case F_MISC: launch_g_check_misc(A, nunits, ngroups, st); break;
    case F_RANGE: launch_g_check_range(A, nunits, ngroups, st); break;
    case F_SELROW: launch_g_check_selrow(A, nunits, ngroups, st); break;
    case F_LD: launch_g_check_ld(A, nunits, ngroups, st); break;
    case F_RL: launch_g_check_rl(A, nunits, ngroups, st); break;
    case F_SC: launch_g_check_sc(A, nunits, ngroups, st); break;
    case F_POS: launch_g_check_pos(A, nunits, ngroups, st); break;
    case F_GM: launch_g_check_gm(A, nunits, ngroups, st); break;
    default: launch_g_check_n2b(A, nunits, ngroups, st); break;
    }
}
static void launch_g_emit(const GArgs& A, uint32_t cls, uint32_t nunits, hipStream_t st) {
    if (cls == 5) launch_g_emit_gm(A, nunits, 1, st); else if (cls == 3) launch_g_emit_sc(A, nunits, 1, st); else if (cls) launch_g_emit_heavy(A, nunits, 1, st); else launch_g_emit_light(A, nunits, 1, st);
}

static void launch_inputs(pob_ctx* h, int mode, uint32_t G, hipStream_t st) {      // mode: 0 generation, 1 evaluation, 2 generation + its evaluation (k_inputs)
    if (h->circuit != POB_CIRCUIT_PROOF_OF_BURN || !h->plan.nsm_in) return;
    const SmRef r0 = h->plan.L.pm.numLeafAddressNibbles;  // the small inputs are contiguous SM ranks / wire indices from here (declaration order)
    const uint32_t fk = (mode == 2 && h->fault_armed && !h->fault_g && h->fault_cls == POB_CLASS_SM) ? (uint32_t)h->fault_word - r0.i : 0xFFFFFFFFu;     // pob_debug_store_fault: the input row
    if (h->in_bytes[h->in_cur]) {
        const dim3 grid8((h->plan.nsm_in + IN8_K - 1) / IN8_K, G);
        const size_t lds = 64 * (IN8_K + 1) * 4;
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid8, dim3(256), lds, st, h->d_in_sm8[h->in_cur], h->d_in_exc[h->in_cur], h->plan.nsm_in, h->n, h->d_sm, (uint64_t)h->plan.total.s * 64, r0.i, r0.w, h->d_bad, fk, h->fault_group, h->fault_mask); };
        if (mode == 1) go(k_inputs8<1>); else if (mode == 2) go(k_inputs8<2>); else go(k_inputs8<0>);
        return;
    }
    const dim3 grid((h->plan.nsm_in + 63) / 64, G);
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(64), 64 * 65 * 4, st, h->d_in_sm[h->in_cur], h->plan.nsm_in, h->d_sm, (uint64_t)h->plan.total.s * 64, r0.i, r0.w, h->d_bad, fk, h->fault_group, h->fault_mask); };
    if (mode == 1) go(k_inputs<1>); else if (mode == 2) go(k_inputs<2>); else go(k_inputs<0>);
}
static Fr limbs_to_mont(const uint64_t* l) {
    Fr c; for (int i = 0; i < 4; i++) { c.l[2 * i] = (uint32_t)l[i]; c.l[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    return fr_to_mont(c);
}

static GArgs gargs(pob_ctx* h) {
    GArgs A; memset(&A, 0, sizeof A);
    A.units = h->d_units; A.order = h->d_order; A.L = h->d_L;
    A.bits = h->d_bits; A.sm = h->d_sm; A.fr = h->d_fr;
    A.bits_stride = h->plan.total.b; A.sm_stride = (uint64_t)h->plan.total.s * 64; A.fr_stride = (uint64_t)h->plan.total.f * 512;
    A.pos_tab = h->d_pos; A.inv_lut = h->d_inv; A.pow256 = h->d_pow256; A.npow256 = h->npow256; A.in_fr = h->d_in_fr[h->in_cur]; A.in_sm = h->d_in_sm[h->in_cur];
    A.nfr_in = h->plan.nfr_in; A.nsm_in = h->plan.nsm_in;
    A.status = h->d_status_raw; A.chk_status = h->d_chk; A.bad_wire = h->d_bad; A.emit_counters = h->d_emit_ctr;
    A.fault_cls = 0xFFFFFFFFu;
    return A;
}
static KArgs kargs(pob_ctx* h) {
    KArgs K; memset(&K, 0, sizeof K);
    K.bits = (u64*)h->d_bits; K.group_stride = h->plan.total.b; K.sponges = h->d_sponges;
    K.perm_sponge = h->d_perm_sponge; K.perm_block = h->d_perm_block; K.bad_wire = h->d_bad;
    return K;
}

// a small template parameter: 4 x 64-bit limbs whose upper limbs must be zero
static bool small_param(const uint64_t* l, uint64_t lo, uint64_t hi, int* out) {
    if (l[1] | l[2] | l[3]) return false;
    if (l[0] < lo || l[0] > hi) return false;
    *out = (int)l[0];
    return true;
}
// a field-valued template parameter (the two balance bounds): reduced mod p, Montgomery
static Fr field_param(const uint64_t* l) {
    Fr c; for (int i = 0; i < 4; i++) { c.l[2 * i] = (uint32_t)l[i]; c.l[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    for (int k = 0; k < 6 && fr_geq_p(c); k++) c = fr_sub_p(c);        // 2^256 / p < 6
    return fr_to_mont(c);
}
static int make_plan(Plan& plan, std::string& err, int circuit, const uint64_t* params, int nparams) {
    if (circuit == POB_CIRCUIT_PROOF_OF_BURN) {
        if (nparams != 8) { err = "ProofOfBurn takes 8 template parameters"; return POB_E_ARG; }
        PobParams prm;
        if (!small_param(params + 0, 2, 64, &prm.L) || !small_param(params + 4, 1, 16, &prm.NB) || !small_param(params + 8, 1, 32, &prm.HB) ||
            !small_param(params + 12, 0, 64, &prm.minNib) || !small_param(params + 16, 1, 31, &prm.amountBytes) || !small_param(params + 20, 0, 32, &prm.powZero) ||
            4 + prm.L > MAX_KB || prm.L > MAX_SC) {
            err = "unsupported ProofOfBurn parameters (maxNumLayers 2..64, maxNodeBlocks 1..16, maxHeaderBlocks 1..32, minLeafAddressNibbles 0..64, "
                  "amountBytes 1..31, powMinimumZeroBytes 0..32)";
            return POB_E_ARG;
        }
        prm.maxIntended = field_param(params + 24); prm.maxActual = field_param(params + 28);
        // The layout uses 32-bit wire indices and class ranks with 32-bit byte offsets (BIT rank << 3, SM << 8, FR << 11, SB << 6):
        // reject instantiations that cannot fit BEFORE the cursors could wrap (2.6 M wires per Keccak-f permutation dominate).
        const uint64_t perms = (uint64_t)prm.L * prm.NB + prm.HB + 2 + 1 + 1;
        if (perms * 2600000ull >= (1ull << 29)) { err = "instantiation too large for the 32-bit layout offsets (more than 206 Keccak-f permutations)"; return POB_E_ARG; }
        plan.plan_pob(prm);
    } else if (circuit == POB_CIRCUIT_SPEND) {
        int mab = 0;
        if (nparams != 1 || !small_param(params, 1, 31, &mab)) { err = "Spend takes maxAmountBytes in 1..31"; return POB_E_ARG; }
        SpendParams sp; sp.maxAmountBytes = mab;
        plan.plan_spend(sp);
    } else if (circuit == POB_CIRCUIT_GADGET) {
        // params[0] = the template (pob_gadget_template), params[1..] = its template parameters
        int tid = 0, prm[4] = {0, 0, 0, 0};
        if (nparams < 1 || nparams > 5 || !small_param(params, 1, GM_TEMPLATE_COUNT - 1, &tid)) { err = "gadget main: params[0] must be a template id (pob_gadget_template)"; return POB_E_ARG; }
        for (int k = 1; k < nparams; k++) if (!small_param(params + 4 * k, 0, 1 << 20, &prm[k - 1])) { err = "gadget main: template parameter out of range"; return POB_E_ARG; }
        const char* why = plan.plan_gadget((uint32_t)tid, prm, nparams - 1);
        if (why) { err = std::string("gadget main: ") + why; return POB_E_ARG; }
    } else { err = "unknown circuit"; return POB_E_ARG; }
    const Cur t = plan.total;
    if (t.b >= (1u << 29) || t.s >= (1u << 24) || t.f >= (1u << 21) || t.q >= (1u << 30)) {
        err = "instantiation exceeds the layout's offset limits (BIT < 2^29, SM < 2^24, FR < 2^21 wires)"; return POB_E_ARG;
    }
    return POB_OK;
}
static void fill_info(const Plan& pl, uint32_t nperms, uint32_t max_batch, pob_info_t* info) {
    info->n_witness = pl.total.w; info->n_bit = pl.total.b; info->n_sm = pl.total.s; info->n_fr = pl.total.f;
    info->n_fr_inputs = pl.nfr_in; info->n_sm_inputs = pl.nsm_in; info->n_outputs = pl.L.circuit == 2 ? pl.L.gm.nout : 1;
    info->n_units = (uint32_t)pl.units.size(); info->n_sponges = (uint32_t)pl.sponges.size(); info->n_perms = nperms;
    { std::vector<char> used(pl.max_stage + 1, 0); for (const UnitDesc& u : pl.units) if (u.flags & UNIT_GEN) used[u.stage] = 1; for (const SpongeDesc& s : pl.sponges) used[s.stage] = 1;
      info->n_stages = 0; for (char c : used) info->n_stages += c; }
    info->max_batch = max_batch;
    info->group_bytes = (uint64_t)pl.total.b * 8 + (uint64_t)pl.total.s * 256 + (uint64_t)pl.total.f * 2048;      // (derived wires take no storage)
    info->n_derived = pl.total.q;
    info->n_alias = (uint64_t)nperms * ABSORB_ALIAS;
    info->kchk_rounds = (uint32_t)pob_kchk_rounds(); info->kgc_rounds = (uint32_t)pob_kgc_rounds();
    info->keccak_bit_wires = 0;
    for (const SpongeDesc& s : pl.sponges) info->keccak_bit_wires += (uint64_t)s.n * (ABSORB_WIRES + 2 * 1088) + (uint64_t)(s.n + 1) * 1600;
}

extern "C" {

int pob_plan_info(int circuit, const uint64_t* params, int nparams, pob_info_t* info) {
    if (!info) return POB_E_ARG;
    Plan* plan = new Plan(); std::string err;
    int rc;
    try { rc = make_plan(*plan, err, circuit, params, nparams); } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); rc = POB_E_STATE; }
    if (rc == POB_OK) { uint32_t np = 0; for (const SpongeDesc& sd : plan->sponges) np += sd.n; fill_info(*plan, np, 0, info); }
    delete plan;
    return rc;
}

int pob_gadget_template(const char* name, int* nparams) {
    if (!name) return -1;
    for (const GmName& g : GM_NAMES) if (!strcmp(g.name, name)) { if (nparams) *nparams = g.nparams; return (int)g.id; }
    return -1;
}

const char* pob_strerror(pob_handle h) { return h ? h->err.c_str() : "null handle"; }

int pob_open(int device, int circuit, const uint64_t* params, int nparams, uint32_t max_batch, pob_handle* out) {
    if (!out || max_batch == 0) return POB_E_ARG;
    pob_ctx* h = new pob_ctx();
    *out = h;
    h->device = device; h->circuit = circuit; h->max_batch = max_batch; h->groups = (max_batch + 63) / 64;
    try { int prc = make_plan(h->plan, h->err, circuit, params, nparams); if (prc) return prc; }
    catch (const std::exception& e) { h->err = e.what(); return POB_E_STATE; }
    Plan& pl = h->plan;
    {   // POSEIDON_PREFIX + 0/1/2  (constants.circom:3-14) = keccak("EIP-7503") mod p
        const uint64_t pre[4] = {0xf0363f983d892f7eULL, 0xd115b780980a6b46ULL, 0x007d2482cd46cec2ULL, 0x0ba44186ee7876b8ULL};
        Fr base = limbs_to_mont(pre);
        pl.L.prefix[0] = base; pl.L.prefix[1] = fr_add(base, fr_one_mont()); pl.L.prefix[2] = fr_add(pl.L.prefix[1], fr_one_mont());
    }
    // ---- schedule: sponges sorted by stage, perms flattened; units grouped by (stage, needs LDS table)
    // (a stage's LONG sponges -- the header's 16 blocks -- on a stream of their own, so that the layers' expansion does not wait for the
    //  long chain, was measured in round 2: the long chain beside the expansion takes 4.3-4.8 ms instead of 0.7 and the step does not
    //  get shorter; removed)
    std::stable_sort(pl.sponges.begin(), pl.sponges.end(), [&](const SpongeDesc& a, const SpongeDesc& b) { return a.stage < b.stage; });
    std::vector<uint32_t> perm_sponge, perm_block;
    for (uint32_t s = 0; s <= pl.max_stage; s++) {
        for (uint32_t lds = N_GEN_CLASSES; lds-- > 0;) {  // Poseidon / BN254 units first: they run on the second stream beside the light ones
            pob_ctx::Seg sg{s, lds, (uint32_t)h->order.size(), 0};
            for (uint32_t u = 0; u < pl.units.size(); u++) {
                if (!(pl.units[u].flags & UNIT_GEN)) continue;
                if (pl.units[u].stage == s && unit_class(pl.units[u].kind, true) == lds) h->order.push_back(u);
            }
            sg.count = (uint32_t)h->order.size() - sg.first;
            std::stable_sort(h->order.begin() + sg.first, h->order.end(), [&](uint32_t a, uint32_t b) { return pl.units[a].cost > pl.units[b].cost; });
            if (sg.count) h->segs.push_back(sg);
        }
        {
            pob_ctx::KSeg ks{s, 0, 0, (uint32_t)perm_sponge.size(), 0, nullptr};
            bool first = true;
            for (uint32_t i = 0; i < pl.sponges.size(); i++) if (pl.sponges[i].stage == s) {
                if (first) { ks.sp_first = i; first = false; }
                ks.sp_count++;
                for (uint32_t b = 0; b < pl.sponges[i].n; b++) { perm_sponge.push_back(i); perm_block.push_back(b); }
            }
            ks.perm_count = (uint32_t)perm_sponge.size() - ks.perm_first;
            if (ks.sp_count) h->ksegs.push_back(ks);
        }
    }
    h->nperms = (uint32_t)perm_sponge.size();
    for (uint32_t cls = 0; cls < N_GEN_CLASSES; cls++) {  // emission: every emitting unit once, grouped by class
        pob_ctx::Seg sg{0, cls, (uint32_t)h->order.size(), 0};
        for (uint32_t u = 0; u < pl.units.size(); u++) if ((pl.units[u].flags & UNIT_EMIT) && unit_class(pl.units[u].kind) == cls) h->order.push_back(u);
        sg.count = (uint32_t)h->order.size() - sg.first;
        if (sg.count) h->emit_segs.push_back(sg);
    }
    for (uint32_t fam = 0; fam < F_COUNT; fam++) {       // constraint evaluation: every evaluator unit once, one launch per family, long units first
        pob_ctx::Seg sg{0, fam, (uint32_t)h->order.size(), 0};
        for (uint32_t u = 0; u < pl.units.size(); u++) if ((pl.units[u].flags & UNIT_CHECK) && fam_of(pl.units[u].kind) == fam) h->order.push_back(u);
        sg.count = (uint32_t)h->order.size() - sg.first;
        std::stable_sort(h->order.begin() + sg.first, h->order.end(), [&](uint32_t a, uint32_t b) { return pl.units[a].cost > pl.units[b].cost; });
        if (sg.count) h->chk_segs.push_back(sg);
    }

    // ---- levels of the in-order schedule.  G units of global stage sid = node (sid, G); its sponges = node (sid, K) one level later.
    //      A stage depends on the stage before it in its track, on the stage its track forked after, and on the last stage of every track joined before it.
    //      Two plans: [0] one launch per kernel; [1] fused launches (pob_ctx::fused).  In the fused plan of the ProofOfBurn circuit the Poseidon blocks (side track 1, first
    //      stage) are held back to the level of the main track's first sponge chain -- header and layers, 17 blocks: the longest chain -- and share its launch; the units
    //      that continue from the blocks' outputs follow one level later than they could (they have the slack: the side tracks join the main track well behind it).
    for (int variant = 0; variant < 2; variant++) {
        const bool fused = variant == 1;
        const uint32_t NS = (pl.max_stage / Plan::TRACK_STRIDE + 1) * Plan::TRACK_STRIDE;
        std::vector<int> lvl_end(NS, 0), lvl_g(NS, 0);
        std::vector<char> has_g(NS, 0), has_k(NS, 0);
        for (const pob_ctx::Seg& sg : h->segs) has_g[sg.stage] = 1;
        for (const pob_ctx::KSeg& ks : h->ksegs) has_k[ks.stage] = 1;
        uint32_t pos_stage = 0xFFFFFFFFu;                 // the stage of the Poseidon blocks to hold back (fused plan, ProofOfBurn: track 1's first stage), chain_stage: the chain they join
        const uint32_t chain_stage = 2;
        if (fused && circuit == POB_CIRCUIT_PROOF_OF_BURN && chain_stage < NS && has_k[chain_stage])
            for (const pob_ctx::Seg& sg : h->segs) if (sg.lds == 4 && sg.stage / Plan::TRACK_STRIDE == 1) pos_stage = sg.stage;
        std::vector<int> track_end(Plan::MAX_TRACKS, 0);
        // tracks in an order in which every fork parent and every joined track is complete when needed: a track is joined only by a lower-numbered track and
        // forked from a lower-numbered one, so resolve iteratively until nothing changes (the graph is tiny)
        for (int pass = 0; pass < (int)Plan::MAX_TRACKS + 2; pass++) {
            for (uint32_t t = 0; t < std::max(pl.ntracks, 1u); t++) {
                int cur = t ? lvl_end[pl.track_fork[t]] : 0;
                for (uint32_t s2 = 0; s2 < Plan::TRACK_STRIDE; s2++) {
                    const uint32_t sid = t * Plan::TRACK_STRIDE + s2;
                    if (sid >= NS) break;
                    for (uint32_t u = 1; u < pl.ntracks; u++) if (pl.track_join[u] == sid) cur = std::max(cur, track_end[u]);
                    if (sid == pos_stage) cur = std::max(cur, lvl_end[chain_stage] - 1);      // (the whole stage moves: its other units are the five range checks of the inputs)
                    if (has_g[sid]) { cur += 1; lvl_g[sid] = cur; }
                    if (has_k[sid]) cur += 1;
                    lvl_end[sid] = cur;
                }
                track_end[t] = cur;
            }
        }
        int nlev = 0;
        for (uint32_t sid = 0; sid < NS; sid++) nlev = std::max(nlev, lvl_end[sid]);
        if (!fused) h->nlevels = (uint32_t)nlev;
        std::vector<pob_ctx::GenLaunch>& plan = h->gen_plan[variant];
        for (int lv = 1; lv <= nlev; lv++) {
            // (one launch per class -- BN254 | SubstringCheck | light -- instead of the merged one: 2.24 ms per step against 2.04 with 4 calculators in flight, 1.83 against 1.72
            //  with 8, two interleaved pairs on one box: profiles/round4_experiments.txt 10)
            pob_ctx::GenLaunch pos{pob_ctx::GL_UNITS, 4, 0, 0, 0, 0};
            for (uint32_t cls : {4u, 5u, (uint32_t)CLS_ALL}) {     // the lane-spread Poseidon blocks (a level's longest pole) first, gadget mains, then EVERYTHING else in one launch
                pob_ctx::GenLaunch gl{pob_ctx::GL_UNITS, cls, (uint32_t)h->order.size(), 0, 0, 0};
                for (const pob_ctx::Seg& sg : h->segs) if ((cls == CLS_ALL ? (sg.lds == 0 || sg.lds == 1 || sg.lds == 3) : sg.lds == cls) && has_g[sg.stage] && lvl_g[sg.stage] == lv)
                    for (uint32_t j2 = 0; j2 < sg.count; j2++) h->order.push_back(h->order[sg.first + j2]);
                gl.count = (uint32_t)h->order.size() - gl.first;
                std::stable_sort(h->order.begin() + gl.first, h->order.end(), [&](uint32_t a, uint32_t b) { return pl.units[a].cost > pl.units[b].cost; });
                if (!gl.count) continue;
                if (cls == 4 && fused) pos = gl;                   // held for the level's first chain launch (below); launched by itself if the level has none
                else plan.push_back(gl);
            }
            for (const pob_ctx::KSeg& ks : h->ksegs) if (lvl_end[ks.stage] == lv) {
                if (pos.count) { plan.push_back({pob_ctx::GL_POS_CHAIN, 4, pos.first, pos.count, ks.sp_first, ks.sp_count}); pos.count = 0; }
                else plan.push_back({pob_ctx::GL_CHAIN, 0, 0, 0, ks.sp_first, ks.sp_count});
            }
            if (pos.count) plan.push_back(pos);
        }
        // the round expansion: ONE launch at the end (nothing of the generation reads a round block)
        if (h->nperms) plan.push_back({pob_ctx::GL_ROUNDS, 0, 0, 0, 0, h->nperms});
        if (getenv("POB_DEBUG_LEVELS")) {       // (diagnostic: the in-order launch list with the unit kinds of every launch)
            fprintf(stderr, "---- in-order launch plan %d (%s), %d levels\n", variant, fused ? "fused" : "one launch per kernel", nlev);
            for (const pob_ctx::GenLaunch& gl : plan) {
                fprintf(stderr, "kind %u class %u: %5u units, sponges / permutations [%u, +%u)", gl.kind, gl.cls, gl.count, gl.k_first, gl.k_count);
                if (gl.count && gl.kind != pob_ctx::GL_CHAIN && gl.kind != pob_ctx::GL_ROUNDS) {
                    fprintf(stderr, " longest unit cost %u:", pl.units[h->order[gl.first]].cost);
                    std::vector<uint32_t> kinds(U_KIND_COUNT, 0);
                    for (uint32_t j2 = 0; j2 < gl.count; j2++) kinds[pl.units[h->order[gl.first + j2]].kind]++;
                    for (uint32_t k = 0; k < U_KIND_COUNT; k++) if (kinds[k]) fprintf(stderr, " %u x kind %u", kinds[k], k);
                }
                fprintf(stderr, "\n");
            }
        }
    }

    {   // in-order evaluation: the four narrow families as ONE launch (g_check_narrow.hip), the wide ones (RANGE, SELROW, LD, SC) and the gadget mains as they are
        h->chk_narrow = pob_ctx::Seg{0, 0, (uint32_t)h->order.size(), 0};
        for (uint32_t u = 0; u < pl.units.size(); u++) {
            const uint32_t f = fam_of(pl.units[u].kind);
            if ((pl.units[u].flags & UNIT_CHECK) && (f == F_MISC || f == F_RL || f == F_POS || f == F_N2B)) h->order.push_back(u);
        }
        h->chk_narrow.count = (uint32_t)h->order.size() - h->chk_narrow.first;
        std::stable_sort(h->order.begin() + h->chk_narrow.first, h->order.end(), [&](uint32_t a, uint32_t b) { return pl.units[a].cost > pl.units[b].cost; });
        h->chk_ride_kept.first = (uint32_t)h->order.size();
        for (uint32_t u = 0; u < pl.units.size(); u++) if ((pl.units[u].flags & UNIT_CHECK) && ride_keeps_evaluation(pl.units[u].kind)) h->order.push_back(u);
        h->chk_ride_kept.count = (uint32_t)h->order.size() - h->chk_ride_kept.first;
    }

    HIPC(hipSetDevice(device));
    // the side-track streams get dispatch priority: their few workgroups take the next free slots instead of queueing behind the
    // 30k workgroups of a Keccak expansion running on the caller's stream
    int prio_lo = 0, prio_hi = 0;
    HIPC(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    {   // the device's shared side streams (see StreamPool)
        std::lock_guard<std::mutex> lk(g_pool_mu);
        StreamPool* P = nullptr;
        for (StreamPool* q : g_pools) if (q->device == device) P = q;
        if (!P) {       // (registered only when every stream exists: a half-built pool must not be found by the next pob_open)
            P = new StreamPool(); P->device = device;
            // (s_in: the input uploads of every handle of the device -- copies, ordered by events like the rest)
            hipStream_t* want[7] = {&P->stream2, &P->stream_k, &P->track1, &P->track2, &P->chk1, &P->chk2, &P->s_in};
            hipError_t e = hipSuccess;
            // (the streaming stream at the lowest priority; normal priority measured the same step, the highest lets the round evaluation
            //  finish in 5.1 ms instead of 7.6 but the step grows from 13.2 to 14.8 ms: the G kernels it displaces are needed next)
            // (the evaluation families' two streams at the LOW priority like the streaming stream -- the device has two levels: their chip-filling launches
            //  have slack, the next batch's generation chain that shares the read phase with them does not: 13.31 -> 13.20 ms per step, five interleaved pairs)
            for (int k = 0; k < 7 && e == hipSuccess; k++) e = hipStreamCreateWithPriority(want[k], hipStreamNonBlocking, (k == 1 || k == 4 || k == 5) ? prio_lo : prio_hi);
            if (e != hipSuccess) {
                for (hipStream_t* q : want) if (*q) hipStreamDestroy(*q);
                delete P;
                h->err = std::string("hipStreamCreateWithPriority: ") + hipGetErrorString(e);
                return POB_E_HIP;
            }
            g_pools.push_back(P);
        }
        P->refs++; h->pool = P;
    }
    h->stream2 = h->pool->stream2; h->stream_k = h->pool->stream_k;
    hipStream_t p_track1 = h->pool->track1, p_track2 = h->pool->track2;
    HIPC(hipEventCreateWithFlags(&h->ev_join3, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ev_rounds_fork, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ev_g_done, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&h->ev_k_done, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ev_gen_done, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&h->ev_check_done, hipEventDisableTiming));
    for (pob_ctx::KSeg& ks : h->ksegs) HIPC(hipEventCreateWithFlags(&ks.ev_done, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    for (uint32_t t = 1; t < Plan::MAX_TRACKS; t++) {        // (streams for every track slot: the evaluation uses tracks 1 and 2's whatever the circuit)
        pob_ctx::Track& T = h->tracks[t];
        T.s_main = t == 1 ? p_track1 : t == 2 ? p_track2 : t == 3 ? p_track2 : t == 6 ? p_track1 : h->stream2;   // tracks 4 and 5 follow each other on the BN254 stream; 3 follows 2; 6 follows 1
        T.s_heavy = T.s_main;
        HIPC(hipEventCreateWithFlags(&T.ev_fork, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&T.ev_join, hipEventDisableTiming));
        HIPC(hipEventCreateWithFlags(&T.ev_start, hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&T.ev_end, hipEventDisableTiming));
    }
    h->stream3 = p_track1;
    const uint64_t G = h->groups, npad = G * 64;
    HIPC(hipMalloc(&h->d_bits, G * (uint64_t)std::max(pl.total.b, 1u) * 8));
    HIPC(hipMalloc(&h->d_sm, G * (uint64_t)std::max(pl.total.s, 1u) * 256));
    HIPC(hipMalloc(&h->d_fr, G * (uint64_t)std::max(pl.total.f, 1u) * 2048));
    HIPC(hipMalloc(&h->d_units, pl.units.size() * sizeof(UnitDesc)));
    HIPC(hipMalloc(&h->d_order, h->order.size() * sizeof(uint32_t)));
    HIPC(hipMalloc(&h->d_L, sizeof(CircuitLayout)));
    HIPC(hipMalloc(&h->d_sponges, std::max<size_t>(pl.sponges.size(), 1) * sizeof(SpongeDesc)));
    HIPC(hipMalloc(&h->d_perm_sponge, std::max<size_t>(h->nperms, 1) * 4));
    HIPC(hipMalloc(&h->d_perm_block, std::max<size_t>(h->nperms, 1) * 4));
    HIPC(hipMalloc(&h->d_pos, sizeof(POS_TABLE_MONT)));
    HIPC(hipMalloc(&h->d_inv, 8193 * 32));
    HIPC(hipMalloc(&h->d_emit_ctr, 4 * 4)); HIPC(hipMemset(h->d_emit_ctr, 0, 4 * 4));
    {   // 256^i (Montgomery) and 256^i * R^2 for i < 136*NB (SubstringCheck's M[] / exists[], substring_check.circom:45-49,91)
        h->npow256 = 136u * (uint32_t)std::max(pl.L.pob.NB, 1);
        std::vector<Fr> tab(2 * (size_t)h->npow256);
        const Fr c256 = fr_from_i64(256);
        Fr pw = fr_one_mont();
        for (uint32_t i = 0; i < h->npow256; i++) { tab[i] = pw; tab[h->npow256 + i] = fr_to_mont(pw); pw = fr_mul(pw, c256); }
        HIPC(hipMalloc(&h->d_pow256, tab.size() * sizeof(Fr)));
        HIPC(hipMemcpy(h->d_pow256, tab.data(), tab.size() * sizeof(Fr), hipMemcpyHostToDevice));
    }
    for (int k = 0; k < 2; k++) {
        HIPC(hipMalloc(&h->d_in_fr[k], std::max<uint64_t>(npad * (uint64_t)pl.nfr_in * 32, 32)));
        HIPC(hipMalloc(&h->d_in_sm[k], std::max<uint64_t>(npad * (uint64_t)pl.nsm_in * 4, 4)));
        HIPC(hipMemset(h->d_in_fr[k], 0, std::max<uint64_t>(npad * (uint64_t)pl.nfr_in * 32, 32)));
        HIPC(hipMemset(h->d_in_sm[k], 0, std::max<uint64_t>(npad * (uint64_t)pl.nsm_in * 4, 4)));
        if (pl.nsm_in) {      // (allocated here, not on first use: a service loop's first byte-form upload must not pay a hipMalloc)
            HIPC(hipMalloc(&h->d_in_sm8[k], npad * (uint64_t)pl.nsm_in + 16));
            HIPC(hipMalloc(&h->d_in_exc[k], npad * POB_EXC_CAP * sizeof(pob_sm_exc_t)));
        }
        HIPC(hipEventCreateWithFlags(&h->ev_in_done[k], hipEventDisableTiming));
    }
    HIPC(hipMalloc(&h->d_status_raw, npad * 4)); HIPC(hipMalloc(&h->d_status, npad * 4));
    HIPC(hipMalloc(&h->d_chk, npad * 4)); HIPC(hipMalloc(&h->d_bad, npad * 4));
    HIPC(hipMalloc(&h->d_outputs, npad * 32));
    HIPC(hipMalloc(&h->d_records, npad * POB_RECORD_BYTES));
    for (int k = 0; k < 2; k++) {                           // the host's copy of the records: pinned, written by the device directly
        HIPC(hipHostMalloc((void**)&h->h_records[k], npad * POB_RECORD_BYTES, hipHostMallocDefault));
        HIPC(hipEventCreateWithFlags(&h->ev_rec[k], hipEventDisableTiming));
    }

    HIPC(hipMemcpy(h->d_units, pl.units.data(), pl.units.size() * sizeof(UnitDesc), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(h->d_order, h->order.data(), h->order.size() * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(h->d_L, &pl.L, sizeof(CircuitLayout), hipMemcpyHostToDevice));
    if (!pl.sponges.empty()) HIPC(hipMemcpy(h->d_sponges, pl.sponges.data(), pl.sponges.size() * sizeof(SpongeDesc), hipMemcpyHostToDevice));
    if (h->nperms) {
        HIPC(hipMemcpy(h->d_perm_sponge, perm_sponge.data(), h->nperms * 4, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(h->d_perm_block, perm_block.data(), h->nperms * 4, hipMemcpyHostToDevice));
    }
    HIPC(hipMemcpy(h->d_pos, POS_TABLE_MONT, sizeof(POS_TABLE_MONT), hipMemcpyHostToDevice));
    if (h->nperms) {                                    // the round blocks' alias table (emission expands the unstored wires through it)
        std::vector<uint16_t> tab(KECCAKF_ROUND_WIRES);
        if (!keccak_alias_table_host(tab.data())) { h->err = "internal: the KeccakfRound walk does not name every wire of the block exactly once"; return POB_E_STATE; }
        HIPC(hipMalloc(&h->d_ktab, tab.size() * 2));
        HIPC(hipMemcpy(h->d_ktab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    }
    HIPC(hipMemset(h->d_status_raw, 0xFF, npad * 4)); HIPC(hipMemset(h->d_chk, 0xFF, npad * 4)); HIPC(hipMemset(h->d_bad, 0xFF, npad * 4));
    hipLaunchKernelGGL(k_init_invlut, dim3((8193 + 63) / 64), dim3(64), 0, h->stream2, h->d_inv);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(h->stream2));
    return POB_OK;
}

void pob_close(pob_handle h) {
    if (!h) return;
    hipSetDevice(h->device);
    void* ptrs[] = {h->d_bits, h->d_sm, h->d_fr, h->d_units, h->d_order, h->d_L, h->d_sponges, h->d_perm_sponge, h->d_perm_block, h->d_pos,
                    h->d_inv, h->d_pow256, h->d_ktab, h->d_emit_ctr, h->d_in_fr[0], h->d_in_fr[1], h->d_in_sm[0], h->d_in_sm[1], h->d_in_sm8[0], h->d_in_sm8[1], h->d_in_exc[0], h->d_in_exc[1], h->d_status_raw, h->d_status, h->d_chk, h->d_bad, h->d_outputs, h->d_records, h->em.d_win[0], h->em.d_win[1], h->em.d_win[2], h->em.d_order, h->em.d_probe, h->em.d_rbits, h->em.d_rpre, h->em.d_sc_z, h->em.d_sc_m, h->em.d_sc_c, h->em.d_sc_res, h->em.d_sc_zr, h->em.d_sc_mr};
    for (void* p : ptrs) if (p) hipFree(p);
    for (int k = 0; k < pob_ctx::Emit::NSLOT; k++) {
        if (h->em.h_pin[k]) hipHostFree(h->em.h_pin[k]);
        for (hipEvent_t e : {h->em.ev_made[k], h->em.ev_copied[k], h->em.ev_free[k]}) if (e) hipEventDestroy(e);
    }
    hipDeviceSynchronize();                             // (the pool's streams may still carry this handle's work)
    for (int k = 0; k < 2; k++) { if (h->h_records[k]) hipHostFree(h->h_records[k]); if (h->ev_rec[k]) hipEventDestroy(h->ev_rec[k]); }
    if (h->ev_upload) hipEventDestroy(h->ev_upload);
    for (hipEvent_t e : h->ev_in_done) if (e) hipEventDestroy(e);
    for (auto& pr : h->ev_kchk) for (hipEvent_t e : pr) if (e) hipEventDestroy(e);
    if (h->partner && h->partner->partner == h) h->partner->partner = nullptr;
    if (h->em.s_copy) hipStreamDestroy(h->em.s_copy);
    if (h->pool) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        StreamPool* P = h->pool;
        if (--P->refs == 0) {
            for (hipStream_t q : {P->stream2, P->stream_k, P->track1, P->track2, P->chk1, P->chk2, P->s_in}) if (q) hipStreamDestroy(q);
            g_pools.erase(std::find(g_pools.begin(), g_pools.end(), P));
            delete P;
        }
    }
    if (h->ev_gen_done) hipEventDestroy(h->ev_gen_done);
    if (h->ev_check_done) hipEventDestroy(h->ev_check_done);
    if (h->ev_rounds_fork) hipEventDestroy(h->ev_rounds_fork);
    if (h->ev_g_done) hipEventDestroy(h->ev_g_done);
    if (h->ev_k_done) hipEventDestroy(h->ev_k_done);
    for (pob_ctx::KSeg& ks : h->ksegs) if (ks.ev_done) hipEventDestroy(ks.ev_done);
    if (h->stream) hipStreamDestroy(h->stream);
    for (pob_ctx::Track& T : h->tracks) {
        for (hipEvent_t e : {T.ev_fork, T.ev_join, T.ev_start, T.ev_end}) if (e) hipEventDestroy(e);
    }
    if (h->ev_join3) hipEventDestroy(h->ev_join3);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    delete h;
}

int pob_get_info(pob_handle h, pob_info_t* info) {
    if (!h || !info) return POB_E_ARG;
    fill_info(h->plan, h->nperms, h->max_batch, info);
    return POB_OK;
}

// both upload forms write the buffer the current batch does not use; the next pob_generate switches to it
int pob_upload_inputs(pob_handle h, const uint8_t* fr_inputs, const int32_t* sm_inputs, uint32_t n) {
    if (!h || n == 0 || n > h->max_batch || (h->plan.nfr_in && !fr_inputs) || (h->plan.nsm_in && !sm_inputs)) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    const int t = h->in_cur ^ 1;
    if (h->upload_pending) HIPC(hipEventSynchronize(h->ev_upload));         // an asynchronous upload into the same buffer is still in flight
    if (h->in_done_rec[t]) HIPC(hipEventSynchronize(h->ev_in_done[t]));     // the batch before the current one read this buffer
    if (!h->ev_upload) { h->s_upload = h->pool->s_in; HIPC(hipEventCreateWithFlags(&h->ev_upload, hipEventDisableTiming)); }
    // (on the handle's own non-blocking stream: a hipMemcpy on the legacy stream would wait for every blocking stream of the process)
    if (h->plan.nfr_in) HIPC(hipMemcpyAsync(h->d_in_fr[t], fr_inputs, (uint64_t)n * h->plan.nfr_in * 32, hipMemcpyHostToDevice, h->s_upload));
    if (h->plan.nsm_in) HIPC(hipMemcpyAsync(h->d_in_sm[t], sm_inputs, (uint64_t)n * h->plan.nsm_in * 4, hipMemcpyHostToDevice, h->s_upload));
    HIPC(hipStreamSynchronize(h->s_upload));
    h->in_next = t; h->n_next = n; h->upload_pending = false; h->have_next = true; h->in_bytes[t] = false;
    return POB_OK;
}

int pob_upload_inputs_async(pob_handle h, const uint8_t* fr_inputs, const int32_t* sm_inputs, uint32_t n, void* stream_) {
    if (!h || n == 0 || n > h->max_batch || (h->plan.nfr_in && !fr_inputs) || (h->plan.nsm_in && !sm_inputs)) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    if (!h->ev_upload) { h->s_upload = h->pool->s_in; HIPC(hipEventCreateWithFlags(&h->ev_upload, hipEventDisableTiming)); }
    hipStream_t su = stream_ ? (hipStream_t)stream_ : h->s_upload;
    const int t = h->in_cur ^ 1;
    if (h->upload_pending) HIPC(hipStreamWaitEvent(su, h->ev_upload, 0));
    if (h->in_done_rec[t]) HIPC(hipStreamWaitEvent(su, h->ev_in_done[t], 0));   // the batch before the current one read this buffer (long done in a steady loop)
    if (h->plan.nfr_in) HIPC(hipMemcpyAsync(h->d_in_fr[t], fr_inputs, (uint64_t)n * h->plan.nfr_in * 32, hipMemcpyHostToDevice, su));
    if (h->plan.nsm_in) HIPC(hipMemcpyAsync(h->d_in_sm[t], sm_inputs, (uint64_t)n * h->plan.nsm_in * 4, hipMemcpyHostToDevice, su));
    HIPC(hipEventRecord(h->ev_upload, su));
    h->in_next = t; h->n_next = n; h->upload_pending = true; h->have_next = true; h->in_bytes[t] = false;
    return POB_OK;
}

static int upload8(pob_ctx* h, const uint8_t* fr_inputs, const uint8_t* sm8, const pob_sm_exc_t* exc, uint32_t n, hipStream_t su, bool async) {
    const int t = h->in_cur ^ 1;
    const uint64_t nsm = h->plan.nsm_in;
    const bool bytes_only = h->circuit == POB_CIRCUIT_PROOF_OF_BURN;
    if (async) { if (h->upload_pending) HIPC(hipStreamWaitEvent(su, h->ev_upload, 0)); if (h->in_done_rec[t]) HIPC(hipStreamWaitEvent(su, h->ev_in_done[t], 0)); }
    else { if (h->upload_pending) HIPC(hipEventSynchronize(h->ev_upload)); if (h->in_done_rec[t]) HIPC(hipEventSynchronize(h->ev_in_done[t])); }
    if (h->plan.nfr_in) HIPC(hipMemcpyAsync(h->d_in_fr[t], fr_inputs, (uint64_t)n * h->plan.nfr_in * 32, hipMemcpyHostToDevice, su));
    if (nsm) {
        HIPC(hipMemcpyAsync(h->d_in_sm8[t], sm8, (uint64_t)n * nsm, hipMemcpyHostToDevice, su));
        HIPC(hipMemcpyAsync(h->d_in_exc[t], exc, (uint64_t)n * POB_EXC_CAP * sizeof(pob_sm_exc_t), hipMemcpyHostToDevice, su));
        // ProofOfBurn: nothing else -- the generation's first kernel (and the evaluation's input check) read the byte form (k_inputs8).  The other circuits' kernels
        // read the packed int32 rows themselves (gadget mains: their inputs ARE the packed rows), so the bytes are widened here, on the upload stream
        if (!bytes_only) {
            const uint64_t total = (uint64_t)n * nsm;
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((total / 16 + 255) / 256 + 1, 1024);
            hipLaunchKernelGGL(k_widen_sm8, dim3(blocks), dim3(256), 0, su, h->d_in_sm8[t], h->d_in_sm[t], total);
            hipLaunchKernelGGL(k_apply_exc, dim3((n * POB_EXC_CAP + 255) / 256), dim3(256), 0, su, h->d_in_exc[t], h->d_in_sm[t], n, (uint32_t)nsm);
            HIPC(hipGetLastError());
        }
    }
    if (async) { HIPC(hipEventRecord(h->ev_upload, su)); } else HIPC(hipStreamSynchronize(su));
    h->in_next = t; h->n_next = n; h->upload_pending = async; h->have_next = true; h->in_bytes[t] = bytes_only && nsm;
    return POB_OK;
}
int pob_upload_inputs8(pob_handle h, const uint8_t* fr_inputs, const uint8_t* sm8, const pob_sm_exc_t* exc, uint32_t n) {
    if (!h || n == 0 || n > h->max_batch || (h->plan.nfr_in && !fr_inputs) || (h->plan.nsm_in && (!sm8 || !exc))) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    if (!h->ev_upload) { h->s_upload = h->pool->s_in; HIPC(hipEventCreateWithFlags(&h->ev_upload, hipEventDisableTiming)); }
    return upload8(h, fr_inputs, sm8, exc, n, h->s_upload, false);
}
int pob_upload_inputs8_async(pob_handle h, const uint8_t* fr_inputs, const uint8_t* sm8, const pob_sm_exc_t* exc, uint32_t n, void* stream_) {
    if (!h || n == 0 || n > h->max_batch || (h->plan.nfr_in && !fr_inputs) || (h->plan.nsm_in && (!sm8 || !exc))) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    if (!h->ev_upload) { h->s_upload = h->pool->s_in; HIPC(hipEventCreateWithFlags(&h->ev_upload, hipEventDisableTiming)); }
    return upload8(h, fr_inputs, sm8, exc, n, stream_ ? (hipStream_t)stream_ : h->s_upload, true);
}

int pob_host_alloc(void** p, uint64_t bytes) {
    if (!p || bytes == 0) return POB_E_ARG;
    return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess ? POB_OK : POB_E_NOMEM;
}
void pob_host_free(void* p) { if (p) hipHostFree(p); }

// status / outputs of the batch and its result records; `evaluated`: the evaluator's verdict is part of the record
static int enqueue_collect(pob_ctx* h, hipStream_t st, bool evaluated) {
    const uint32_t G = (h->n + 63) / 64;
    // (a gadget-level main has no commitment: its outputs are the witness' first wires, read through the emitter)
    const uint32_t out_idx = h->circuit == POB_CIRCUIT_PROOF_OF_BURN ? h->plan.L.pm.commitment.i : h->circuit == POB_CIRCUIT_SPEND ? h->plan.L.sm.commitment.i : 0xFFFFFFFFu;
    hipLaunchKernelGGL(k_collect, dim3((G * 64 + 255) / 256), dim3(256), 0, st, h->d_fr, (uint64_t)h->plan.total.f * 512, out_idx, h->d_status_raw, h->d_status, h->d_outputs,
                       h->d_records, (uint32_t*)h->h_records[h->rec_slot], evaluated ? h->d_chk : nullptr, evaluated ? h->d_bad : nullptr, G * 64);
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(h->ev_rec[h->rec_slot], st)); h->rec_n[h->rec_slot] = h->n;
    return POB_OK;
}

int pob_generate(pob_handle h, void* stream_) {
    if (!h || (h->n == 0 && !h->have_next)) return POB_E_STATE;
    HIPC(hipSetDevice(h->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : own_stream(h);
    if (h->have_next) { h->in_cur = h->in_next; h->n = h->n_next; h->have_next = false; }      // the uploaded batch becomes the current one (else: the same inputs again)
    const uint32_t G = (h->n + 63) / 64;
    if (h->upload_pending) { HIPC(hipStreamWaitEvent(st, h->ev_upload, 0)); h->upload_pending = false; }
    if (h->chk_ordered && h->chk_stream != st) HIPC(hipStreamWaitEvent(st, h->ev_check_done, 0));      // the previous batch's evaluation still reads the vector this generation overwrites
    if (h->gen_ordered && h->gen_stream != st) HIPC(hipStreamWaitEvent(st, h->ev_gen_done, 0));        // ... and a previous generation on another stream still writes it
    h->chk_ordered = false;
    h->rec_slot ^= 1;                                   // this batch's records go to the other pinned buffer: the previous batch's stay readable
    if (h->inorder) {
        GArgs A = gargs(h);
        KArgs K = kargs(h);
        if (h->gc && h->rode) { HIPC(hipMemsetAsync(h->d_bad, 0xFF, (size_t)G * 64 * 4, st)); HIPC(hipMemsetAsync(h->d_chk, 0xFF, (size_t)G * 64 * 4, st)); }     // a generation whose verdict nobody collected
        h->rode = false; h->rode_g = false;
        const bool ride_g = h->gc && (h->circuit == POB_CIRCUIT_PROOF_OF_BURN || h->circuit == POB_CIRCUIT_SPEND);
        launch_inputs(h, h->gc ? 2 : 0, G, st);
        const bool kf = h->gc && h->fault_armed && !h->fault_g && h->fault_cls == POB_CLASS_BIT;
        const bool gf = ride_g && h->fault_armed && h->fault_g;             // pob_debug_store_fault on a store of a G unit: the riding kernels' FAULT instantiations
        if (gf) { A.fault_cls = (uint32_t)h->fault_cls; A.fault_group = h->fault_group; A.fault_idx = (uint32_t)h->fault_word; A.fault_lanes = h->fault_mask; }
        K.fault_group = h->fault_group; K.fault_word = h->fault_word; K.fault_mask = h->fault_mask;
        for (const pob_ctx::GenLaunch& gl : h->gen_plan[h->fused ? 1 : 0]) {
            A.first = gl.first; K.first = gl.k_first;
            switch (gl.kind) {
            case pob_ctx::GL_UNITS: launch_g_gen(A, gl.cls, gl.count, G, st, ride_g, gf); break;
            case pob_ctx::GL_CHAIN: launch_k_chain(K, false, gl.k_count, G, st); break;
            case pob_ctx::GL_POS_CHAIN: launch_pos_chain(A, K, gl.count, gl.k_count, G, st, ride_g, gf); break;
            default:
                if (h->gc) {

                    if (h->ev_kchk[h->rec_slot][0]) { HIPC(hipEventRecord(h->ev_kchk[h->rec_slot][0], st)); h->kchk_rec[h->rec_slot] = true; }
                    launch_k_rounds_gc(K, gl.k_count, G, kf, st);
                    if (h->ev_kchk[h->rec_slot][1]) HIPC(hipEventRecord(h->ev_kchk[h->rec_slot][1], st));
                    h->rode = true; h->rode_g = ride_g;
                } else launch_k_rounds(K, false, gl.k_count, G, st);
                break;
            }
        }
        h->fault_armed = false;
        HIPC(hipEventRecord(h->ev_g_done, st));
        HIPC(hipGetLastError());
        { int rc = enqueue_collect(h, st, false); if (rc) return rc; }
        HIPC(hipEventRecord(h->ev_gen_done, st)); h->gen_done_rec = true; h->gen_stream = st; h->gen_ordered = true;
        HIPC(hipEventRecord(h->ev_in_done[h->in_cur], st)); h->in_done_rec[h->in_cur] = true;
        h->generated = true; h->evaluated = false; h->gen_count++;
        h->em.queued_idx = -1;
        return POB_OK;
    }
    // pipeline: this generation's latency-bound work starts with the partner's evaluation (= once the partner's generation is complete)
    // (starting ALL of them even earlier, beside the partner's expansion, was measured: no gain -- the chip-filling launches take from the
    //  expansion what they gain).  The first stage and the NARROW tracks forked after it (the burn-address / RLP / account chains: a few
    //  dozen wavefronts, next to no bandwidth) do start at once, i.e. beside the partner's expansion (+1.2 % on average of nine A/B pairs);
    //  the main track from its second stage on and the wide pre-work track wait for the partner's generation to be complete.  Track 4 is
    //  then enqueued AHEAD of the wide track 5, whose stream it normally shares: it moves to track 2's.
    // (a strictly PHASED schedule -- evaluation kernel alone | the G work of both batches | expansion alone -- was measured too: the evaluation
    //  kernel then runs at 0.77 of the HBM peak inside the step, but the G phase takes 5.3 ms by itself and the step 14.87 ms instead of 13.5)
    const bool gate = h->partner && h->partner->gen_done_rec;
    const bool gate_late = gate && h->plan.ntracks > 1;
    auto track_stream = [&](uint32_t t) { return (t == 4 && h->partner) ? h->pool->track2 : h->tracks[t].s_main; };
    if (gate && !gate_late) HIPC(hipStreamWaitEvent(st, h->partner->ev_gen_done, 0));
    GArgs A = gargs(h);
    KArgs K = kargs(h);
    const Plan& pl = h->plan;
    // one track: its stages in order; within a stage the BN254 / Poseidon units run on the track's second stream beside the light ones,
    // then the stage's Keccak sponges.  Tracks forked after a stage are enqueued completely (highest first) before the next stage, so every
    // event is recorded before anything waits on it.
    const bool rounds_async = h->partner != nullptr;          // the main track's round expansion leaves the track (pipeline mode)
    std::vector<hipEvent_t> pending;
    std::function<int(uint32_t)> run_track = [&](uint32_t t) -> int {
        hipStream_t sm = t ? track_stream(t) : st, sh = t ? track_stream(t) : h->stream2;
        hipEvent_t ef = t ? h->tracks[t].ev_fork : h->ev_fork, ej = t ? h->tracks[t].ev_join : h->ev_join;
        for (uint32_t sid = t * Plan::TRACK_STRIDE; sid < (t + 1) * Plan::TRACK_STRIDE && sid <= pl.max_stage; sid++) {
            for (uint32_t u = 1; u < pl.ntracks; u++) if (pl.track_join[u] == sid) HIPC(hipStreamWaitEvent(sm, h->tracks[u].ev_end, 0));
            bool forked = false;
            // a stage with ONE kind of launch needs no second stream: the fork / join pair costs ~0.15 ms of hand-over latency, and the
            // generation's tail is a chain of such stages (SubstringCheck heads | existence loop | sums | final)
            uint32_t n_light = 0, n_heavy = 0;
            for (const pob_ctx::Seg& sg : h->segs) if (sg.stage == sid) { if (sg.lds) n_heavy++; else n_light++; }
            const bool one_stream = n_light == 0 && n_heavy == 1;
            for (const pob_ctx::Seg& sg : h->segs) if (sg.stage == sid) {
                A.first = sg.first;
                if (sg.lds && one_stream) launch_g_gen(A, sg.lds, sg.count, G, sm);
                else if (sg.lds) {
                    if (!forked) { HIPC(hipEventRecord(ef, sm)); HIPC(hipStreamWaitEvent(sh, ef, 0)); }
                    launch_g_gen(A, sg.lds, sg.count, G, sh);
                    forked = true;
                } else launch_g_gen(A, 0, sg.count, G, sm);
            }
            if (forked) { HIPC(hipEventRecord(ej, sh)); HIPC(hipStreamWaitEvent(sm, ej, 0)); }
            for (const pob_ctx::KSeg& ks : h->ksegs) if (ks.stage == sid) {
                hipStream_t sk = sm;
                K.first = ks.sp_first;
                launch_k_chain(K, false, ks.sp_count, G, sk);
                K.first = ks.perm_first;
                if (rounds_async && t == 0) {
                    // nothing in the generation reads a KeccakfRound block's wires (k_chain wrote every state a later stage uses): the
                    // main track's HBM-streaming expansion leaves the track here and is only joined before the results are collected
                    HIPC(hipEventRecord(h->ev_rounds_fork, sk)); HIPC(hipStreamWaitEvent(h->stream_k, h->ev_rounds_fork, 0));
                    // pipeline: the write-saturating expansion does not run beside the partner's evaluation, it follows it -- IN ORDER on the
                    // device's one streaming stream, where the partner's Keccak evaluation was enqueued before (pob_constraint_check): the two
                    // HBM-saturating kernels of the two batches alternate on one hardware queue without an event hand-over between them
                    // (0.3 ms per phase change when the evaluation ran on the caller's stream and the expansion waited for its end through an event)
                    sk = h->stream_k; pending.push_back(ks.ev_done);
                }
                launch_k_rounds(K, false, ks.perm_count, G, sk);
                HIPC(hipEventRecord(ks.ev_done, sk));
            }
            for (int wide = 0; wide < 2; wide++) {          // narrow tracks first; (pipeline) the partner gate; then the wide ones
                if (wide && gate_late && t == 0 && sid == 0) HIPC(hipStreamWaitEvent(sm, h->partner->ev_gen_done, 0));
                for (uint32_t u = pl.ntracks; u-- > t + 1;) if (pl.track_fork[u] == sid && (int)((pl.wide_tracks >> u) & 1) == wide) {
                    HIPC(hipEventRecord(h->tracks[u].ev_start, sm)); HIPC(hipStreamWaitEvent(track_stream(u), h->tracks[u].ev_start, 0));
                    int rc = run_track(u); if (rc) return rc;
                    HIPC(hipEventRecord(h->tracks[u].ev_end, track_stream(u)));
                }
            }
        }
        return POB_OK;
    };
    launch_inputs(h, 0, G, st);                           // the small inputs' wires (tile transpose): ahead of stage 0, whose forks wait for the stream
    { int rc = run_track(0); if (rc) return rc; }
    HIPC(hipEventRecord(h->ev_g_done, st));              // every stage of the G side is enqueued behind this point of st (the joined tracks included)
    for (hipEvent_t e : pending) HIPC(hipStreamWaitEvent(st, e, 0));
    { int rc = enqueue_collect(h, st, false); if (rc) return rc; }
    HIPC(hipEventRecord(h->ev_gen_done, st)); h->gen_done_rec = true; h->gen_stream = st; h->gen_ordered = true;
    HIPC(hipEventRecord(h->ev_in_done[h->in_cur], st)); h->in_done_rec[h->in_cur] = true;
    h->generated = true; h->evaluated = false; h->gen_count++;
    h->em.queued_idx = -1;                              // an announcement (pob_emit_queue) names a witness of the batch it was made for; a window pre-made from that batch is not reused (pre_made_gen)
    return POB_OK;
}

int pob_constraint_check(pob_handle h, void* stream_) {
    if (!h || !h->generated) return POB_E_STATE;
    HIPC(hipSetDevice(h->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : own_stream(h);
    const uint32_t G = (h->n + 63) / 64;
    if (h->gen_ordered && h->gen_stream != st) HIPC(hipStreamWaitEvent(st, h->ev_gen_done, 0));      // evaluation on another stream than the generation's: behind its end
    GArgs A = gargs(h);
    // The evaluation has no dependencies between launches: one kernel per family (+ the two Keccak kernels), spread over the
    // caller's stream and two side streams: the HBM-streaming Keccak round + chain evaluation alone on the caller's stream from the start,
    // the eight G families on the side streams BESIDE it (N2B, SC, LD, RANGE | RL, POS, MISC, SELROW).  What was measured on the way to this
    // plan (30 configurations, profiles/round2_scheduling_experiments.txt): the Keccak evaluation after the families costs 8.1 ms per
    // pass, beside them 5.8; starting it before the END of the generation (per sponge segment, beside the generation's tail) gains
    // nothing -- the tail is latency-bound on the same memory system and stretches by what the evaluation saves.
    if (h->inorder) {
        // one stream: the Keccak round evaluation (the batch's one bandwidth-bound kernel) first, then the sponge chains, the inputs and the eight families
        // (the round evaluation on a high-priority stream of the device, forked and joined per batch, was measured: 4 / 6 / 8 / 12 calculators in flight
        //  2.39 / 2.04 / 1.91 / 2.14 ms per step against 2.19 / 2.09 / 1.87 / 2.07 here, the kernel 0.49-0.74 against 0.42-0.80 ms: nothing; removed)
        const bool rode = h->rode, rode_g = h->rode && h->rode_g;          // the round blocks and the input rows (rode_g: and every G unit) were evaluated by the launches that wrote them
        h->rode = false; h->rode_g = false;
        if (!h->plan.sponges.empty()) {
            KArgs K = kargs(h); K.first = 0;
            if (!rode) {             // (else: every round block was evaluated by the launch that wrote it, k_rounds_gc, and nothing has touched the vector since)
                if (h->ev_kchk[h->rec_slot][0]) { HIPC(hipEventRecord(h->ev_kchk[h->rec_slot][0], st)); h->kchk_rec[h->rec_slot] = true; }
                launch_k_rounds(K, true, h->nperms, G, st);
                if (h->ev_kchk[h->rec_slot][1]) HIPC(hipEventRecord(h->ev_kchk[h->rec_slot][1], st));
            }
            launch_k_chain(K, true, h->nperms, G, st);
        }
        if (!rode) launch_inputs(h, 1, G, st);
        // (the narrow kernel on a side stream of the calculator, forked behind the generation and joined here, was measured in round 5: 1.87-1.88 ms per step against 1.82-1.84
        //  with 4 in flight; the four wide families as ONE launch: nothing either -- profiles/round5_experiments.txt 4, 7)
        if (rode_g) {       // of the G units only the evaluation of the units that generated on the plain policy is left (circuits.hpp unit_run_ride: the RLP family)
            if (h->chk_ride_kept.count) { A.first = h->chk_ride_kept.first; launch_g_check(A, F_RL, h->chk_ride_kept.count, G, st); }
        } else {
            if (h->chk_narrow.count) { A.first = h->chk_narrow.first; launch_g_check_narrow(A, h->chk_narrow.count, G, st); }
            for (const pob_ctx::Seg& sg : h->chk_segs) if (sg.lds != F_MISC && sg.lds != F_RL && sg.lds != F_POS && sg.lds != F_N2B) { A.first = sg.first; launch_g_check(A, sg.lds, sg.count, G, st); }
        }
        { int rc = enqueue_collect(h, st, true); if (rc) return rc; }
        HIPC(hipEventRecord(h->ev_check_done, st)); h->check_done_rec = true; h->evaluated = true; h->chk_stream = st; h->chk_ordered = true;
        HIPC(hipEventRecord(h->ev_in_done[h->in_cur], st));
        HIPC(hipGetLastError());
        return POB_OK;
    }
    static const uint32_t side_plan[2][5] = {{F_N2B, F_SC, F_LD, F_RANGE, F_GM}, {F_RL, F_POS, F_MISC, F_SELROW, F_COUNT}};      // (F_GM: gadget-level mains only; F_COUNT: no family)
    // side streams: the generation's (idle during a lone handle's evaluation); in pipeline mode -- the partner generates meanwhile -- the
    // pool's two evaluation streams
    hipStream_t side[2] = {h->partner ? h->pool->chk1 : h->stream2, h->partner ? h->pool->chk2 : h->stream3};
    HIPC(hipEventRecord(h->ev_fork, st));
    for (int k = 0; k < 2; k++) {
        HIPC(hipStreamWaitEvent(side[k], h->ev_fork, 0));
        if (k == 1) launch_inputs(h, 1, G, side[k]);
        for (uint32_t fam : side_plan[k])
            for (const pob_ctx::Seg& sg : h->chk_segs) if (sg.lds == fam) { A.first = sg.first; launch_g_check(A, fam, sg.count, G, side[k]); }
    }
    if (!h->plan.sponges.empty()) {
        KArgs K = kargs(h);
        K.first = 0;
        // pipeline: on the device's streaming stream, directly behind this batch's round expansion (in order: no event between the two) and
        // ahead of the partner's next expansion; it needs the G side of the generation (its tail runs beside the expansion), not the
        // result collection on the caller's stream.  A lone handle: on the caller's stream.
        hipStream_t sk = h->partner ? h->stream_k : st;
        if (sk != st) HIPC(hipStreamWaitEvent(sk, h->ev_g_done, 0));
        if (h->ev_kchk[h->rec_slot][0]) { HIPC(hipEventRecord(h->ev_kchk[h->rec_slot][0], sk)); h->kchk_rec[h->rec_slot] = true; }   // measurement (pob_probe_check_kernel): the dominant kernel inside the step
        launch_k_rounds(K, true, h->nperms, G, sk);
        if (h->ev_kchk[h->rec_slot][1]) HIPC(hipEventRecord(h->ev_kchk[h->rec_slot][1], sk));
        // (pipeline: the narrow sponge-chain evaluation -- 84 wavefronts per group, 0.18 ms -- on an evaluation stream beside the families, not in
        //  order between the two chip-filling kernels of the streaming stream, where the machine idled for its duration: -0.02 ms, three interleaved pairs)
        launch_k_chain(K, true, h->nperms, G, h->partner ? side[1] : sk);
        if (sk != st) { HIPC(hipEventRecord(h->ev_k_done, sk)); HIPC(hipStreamWaitEvent(st, h->ev_k_done, 0)); }
    }
    HIPC(hipEventRecord(h->ev_join, side[0])); HIPC(hipEventRecord(h->ev_join3, side[1]));
    HIPC(hipStreamWaitEvent(st, h->ev_join, 0)); HIPC(hipStreamWaitEvent(st, h->ev_join3, 0));
    { int rc = enqueue_collect(h, st, true); if (rc) return rc; }          // the batch's records, now with the evaluator's verdict
    HIPC(hipEventRecord(h->ev_check_done, st)); h->check_done_rec = true; h->evaluated = true; h->chk_stream = st; h->chk_ordered = true;
    HIPC(hipEventRecord(h->ev_in_done[h->in_cur], st));                 // (the evaluation reads the packed inputs too: the input units' relations)
    HIPC(hipGetLastError());
    return POB_OK;
}

int pob_set_partner(pob_handle h, pob_handle partner) {
    if (!h || h == partner || (partner && partner->device != h->device)) return POB_E_ARG;
    if (partner && (h->inorder || partner->inorder)) { h->err = "the two-calculator pipeline links calculators with the track schedule, not in-order ones"; return POB_E_STATE; }
    if (partner && hw_queues_env() < POB_PIPELINE_HW_QUEUES) {
        // ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two linked handles keep ~10 streams busy and a job with
        // more live streams than queues runs 30-150x slower (unrelated kernels serialise behind each other): refuse instead of crawling
        h->err = "the two-calculator pipeline needs GPU_MAX_HW_QUEUES >= " + std::to_string(POB_PIPELINE_HW_QUEUES) + " in the environment BEFORE the HIP runtime "
                 "initialises (it is " + std::to_string(hw_queues_env()) + "); libpob_hip.so sets it when it is loaded first and the variable is unset";
        return POB_E_STATE;
    }
    // the link is symmetric (neither side may be left pointing at a handle that has another partner or is closed): undo both old links first
    for (pob_handle q : {h, partner}) if (q && q->partner) { q->partner->partner = nullptr; q->partner = nullptr; }
    h->partner = partner; h->gen_done_rec = h->check_done_rec = false;
    if (partner) { partner->partner = h; partner->gen_done_rec = partner->check_done_rec = false; }
    return POB_OK;
}

int pob_set_inorder(pob_handle h, int on) {
    if (!h) return POB_E_ARG;
    if (on && h->partner) { h->err = "an in-order calculator has no partner: unlink first (pob_set_partner(h, NULL))"; return POB_E_STATE; }
    h->inorder = on != 0; h->fused = (on & 2) != 0; h->gc = (on & 4) != 0;
    return POB_OK;
}

int pob_sync(pob_handle h) {
    if (!h) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    // this handle's work only: its last generation / evaluation (the side streams are joined into those events) and its copies
    if (h->gen_done_rec) HIPC(hipEventSynchronize(h->ev_gen_done));
    if (h->check_done_rec && h->evaluated) HIPC(hipEventSynchronize(h->ev_check_done));
    HIPC(hipStreamSynchronize(own_stream(h)));
    return POB_OK;
}

// The records are written straight into pinned host memory by the kernel that packs them (k_collect), into one of two buffers that
// alternate per batch: "fetching" is remembering which buffer / event belongs to the current batch, waiting is an event of THIS handle
// only -- a partner handle that is generating or evaluating meanwhile is neither waited for nor delayed.
int pob_results_fetch(pob_handle h) {
    if (!h || !h->generated) return POB_E_STATE;
    h->fetch_slot = h->rec_slot; h->fetch_pending = true;
    return POB_OK;
}

int pob_results_wait(pob_handle h, const uint8_t** records, uint32_t* n) {
    if (!h || !records) return POB_E_ARG;
    if (!h->fetch_pending) { h->err = "pob_results_wait without pob_results_fetch"; return POB_E_STATE; }
    HIPC(hipSetDevice(h->device));
    HIPC(hipEventSynchronize(h->ev_rec[h->fetch_slot]));
    *records = h->h_records[h->fetch_slot];
    if (n) *n = h->rec_n[h->fetch_slot];
    return POB_OK;
}

int pob_results(pob_handle h, uint32_t* status, uint8_t* outputs, uint32_t* check_status, uint32_t* bad_wire) {
    if (!h || !h->generated) return POB_E_STATE;
    int rc = pob_results_fetch(h); if (rc) return rc;
    const uint8_t* rec = nullptr; uint32_t n = 0;
    rc = pob_results_wait(h, &rec, &n); if (rc) return rc;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t* r = (const uint32_t*)(rec + (uint64_t)i * POB_RECORD_BYTES);
        if (status) status[i] = r[0];
        if (check_status) check_status[i] = r[1];       // POB_NOT_EVALUATED when no pob_constraint_check ran on this batch (0xFFFFFFFF = clean)
        if (bad_wire) bad_wire[i] = r[2];
        if (outputs) memcpy(outputs + (uint64_t)i * 32, r + 3, 32);
    }
    return POB_OK;
}

int pob_results_device(pob_handle h, void** d_status, void** d_outputs) {
    if (!h) return POB_E_ARG;
    if (d_status) *d_status = h->d_status;
    if (d_outputs) *d_outputs = h->d_outputs;
    return POB_OK;
}

// ---- the record gather over RCCL for C-ABI hosts.  librccl is dlopen'ed: no link-time dependency (the CPU shim of the tests and single-GPU users never need it)
#include <dlfcn.h>
typedef int (*pob_nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*pob_nccl_errstr_t)(int);
static pob_nccl_allgather_t g_nccl_allgather = nullptr; static pob_nccl_errstr_t g_nccl_errstr = nullptr; static bool g_nccl_tried = false;
int pob_gather_records(pob_handle h, void* comm, void* stream_, void* d_out, uint32_t n_per_rank) {
    if (!h || !comm || !d_out || n_per_rank == 0 || n_per_rank > h->groups * 64) return POB_E_ARG;
    if (!h->generated) return POB_E_STATE;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_nccl_tried) {
            g_nccl_tried = true;
            void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) { g_nccl_allgather = (pob_nccl_allgather_t)dlsym(lib, "ncclAllGather"); g_nccl_errstr = (pob_nccl_errstr_t)dlsym(lib, "ncclGetErrorString"); }
        }
    }
    if (!g_nccl_allgather) { h->err = "librccl.so could not be loaded (pob_gather_records)"; return POB_E_STATE; }
    HIPC(hipSetDevice(h->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : own_stream(h);
    // behind the batch's records: the evaluation's collect if one was enqueued since the generation, else the generation's
    if (h->evaluated && h->check_done_rec) { if (h->chk_stream != st) HIPC(hipStreamWaitEvent(st, h->ev_check_done, 0)); }
    else if (h->gen_done_rec && h->gen_stream != st) HIPC(hipStreamWaitEvent(st, h->ev_gen_done, 0));
    const int rc = g_nccl_allgather(h->d_records, d_out, (size_t)n_per_rank * POB_RECORD_BYTES, /* ncclUint8 */ 1, comm, st);
    if (rc != 0) { h->err = std::string("ncclAllGather: ") + (g_nccl_errstr ? g_nccl_errstr(rc) : "error"); return POB_E_HIP; }
    return POB_OK;
}

int pob_results_records_device(pob_handle h, void** d_records) {
    if (!h || !d_records) return POB_E_ARG;
    *d_records = h->d_records;
    return POB_OK;
}

// ---- streaming emission (reference: writeBinWitness, patch point tests/test.py:36).  The canonical payload (32 B per wire: 6.9 GB for
// the production instantiation) never exists on the device as a whole: it is expanded window by window from the compact resident
// vector, each window goes D2H into pinned memory on a copy stream while the next one is being expanded and the caller consumes
// the previous one.
static int emit_make_window(pob_ctx* h, uint32_t idx, uint64_t k, int slot) {
    pob_ctx::Emit& E = h->em;
    const uint64_t w0 = k * E.win_wires, wn = std::min(E.win_wires, E.total - w0);      // positions of the payload (kept wires in reduced mode)
    hipStream_t st = own_stream(h);
    HIPC(hipStreamWaitEvent(st, E.ev_free[slot], 0));                       // the copy of the window that used this slot before is done
    // any wire nobody owns stays 0xEE.. (not a field element: the byte compare with the oracle catches it); rocclr's fill kernel took
    // 4.5 ms per 256 MiB window, this one runs at the HBM write rate
    hipLaunchKernelGGL(k_fill_ee, dim3(2048), dim3(256), 0, st, (uint4*)E.d_win[slot], wn * 2);
    if (w0 == 0) { const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0}; HIPC(hipMemcpyAsync(E.d_win[slot], one, 32, hipMemcpyHostToDevice, st)); }   // wire 0 = 1 (always kept)
    GArgs A = gargs(h);
    A.emit_out = E.d_win[slot]; A.emit_sel = idx % 64; A.emit_group = idx / 64; A.emit_w0 = (uint32_t)w0; A.emit_wn = (uint32_t)wn;
    if (E.red) { A.emit_rbits = E.d_rbits; A.emit_rpre = E.d_rpre; }
    if (E.probe_win == E.win_wires && E.probe_map == E.map_id && k < E.wsegs.size()) {      // only the units that write into this window
        A.order = E.d_order;
        for (const pob_ctx::Emit::WSeg& sg : E.wsegs[k]) { A.first = sg.first; launch_g_emit(A, sg.cls, sg.count, st); }
    } else {
        for (const pob_ctx::Seg& sg : h->emit_segs) { A.first = sg.first; launch_g_emit(A, sg.lds, sg.count, st); }
    }
    const u64* Gp = (const u64*)h->d_bits + (uint64_t)(idx / 64) * h->plan.total.b;
    // the Keccak kernels' wires: contiguous BIT runs cut to the window (reduced: cut to the window's wire range, runs without a kept wire skipped)
    const uint64_t wire_lo = E.red ? E.keep[w0] : w0, wire_hi = E.red ? (uint64_t)E.keep[w0 + wn - 1] + 1 : w0 + wn;
    for (const pob_ctx::Emit::Run& r : E.runs) {
        const uint64_t lo = std::max<uint64_t>(r.w, wire_lo), hi = std::min<uint64_t>((uint64_t)r.w + r.n, wire_hi);
        if (lo >= hi) continue;
        if (!E.red) {
            if (r.absorb) launch_k_emit_absorb(Gp, E.d_win[slot] + (lo - w0) * 32, r.b, (uint32_t)(lo - r.w), (uint32_t)(hi - lo), idx % 64, h->d_ktab, st);
            else launch_k_emit_bits(Gp, E.d_win[slot] + (lo - w0) * 32, 0, (uint32_t)(r.b + (lo - r.w)), (uint32_t)(hi - lo), idx % 64, st);
        } else {
            const auto a = std::lower_bound(E.keep.begin() + w0, E.keep.begin() + w0 + wn, (uint32_t)lo), b = std::lower_bound(a, E.keep.begin() + w0 + wn, (uint32_t)hi);
            if (a == b) continue;                                           // (a round block whose wires are all dropped costs nothing)
            const uint64_t lo2 = *a, hi2 = (uint64_t)*(b - 1) + 1;
            if (r.absorb) launch_k_emit_absorb_red(Gp, E.d_win[slot], (uint32_t)lo2, r.b, (uint32_t)(lo2 - r.w), (uint32_t)(hi2 - lo2), idx % 64, h->d_ktab, E.d_rbits, E.d_rpre, (uint32_t)w0, (uint32_t)wn, st);
            else launch_k_emit_bits_red(Gp, E.d_win[slot], (uint32_t)lo2, (uint32_t)(r.b + (lo2 - r.w)), (uint32_t)(hi2 - lo2), idx % 64, E.d_rbits, E.d_rpre, (uint32_t)w0, (uint32_t)wn, st);
        }
    }
    if (E.sc_on && E.red) {   // reduced witness: every site whose wires are all kept (through their class representatives), at their ranks (lists built in emit_start)
        const ScWin W{E.d_win[slot], (uint32_t)w0, (uint32_t)wn, E.d_rbits, E.d_rpre};
        uint32_t* res = E.d_sc_res; const uint32_t* zr = E.d_sc_zr; const uint32_t* mr = E.d_sc_mr; const uint32_t* pw = h->d_pow256; const uint32_t nz = E.sc_red_nz, nm = E.sc_red_nm;
        if (nz) hipLaunchKernelGGL(k_selfcheck_zr, dim3((nz + 63) / 64), dim3(64), 0, st, W, zr, nz, res);
        if (nm) hipLaunchKernelGGL(k_selfcheck_mr, dim3((nm + 63) / 64), dim3(64), 0, st, W, mr, nm, pw, res);
    }
    if (E.sc_on && !E.red) {        // the derived wires' relations on the values just written into this window
        const ScWin W{E.d_win[slot], (uint32_t)w0, (uint32_t)wn, nullptr, nullptr};
        uint32_t* res = E.d_sc_res;
        // sites by the wire range of the window; the kernels skip (and count) a site one of whose wires is outside the window
        const auto za = std::lower_bound(E.sc_z.begin(), E.sc_z.end(), (uint32_t)wire_lo, [](uint32_t s, uint32_t v) { return (s & 0x7FFFFFFFu) < v; });
        const auto zb = std::lower_bound(za, E.sc_z.end(), (uint32_t)wire_hi, [](uint32_t s, uint32_t v) { return (s & 0x7FFFFFFFu) < v; });
        const uint32_t nz = (uint32_t)(zb - za);
        const uint32_t* zs = E.d_sc_z + (za - E.sc_z.begin());          // (plain locals: the CPU shim's launch macro evaluates its arguments inside a by-copy lambda)
        if (nz) hipLaunchKernelGGL(k_selfcheck_z, dim3((nz + 63) / 64), dim3(64), 0, st, W, zs, nz, res);
        E.sc_checked += nz;
        // M sites whose M[k+1] lies in this window
        const auto ma = std::lower_bound(E.sc_m_next.begin(), E.sc_m_next.end(), (uint32_t)wire_lo), mb = std::lower_bound(ma, E.sc_m_next.end(), (uint32_t)wire_hi);
        const uint32_t nm = (uint32_t)(mb - ma);
        const uint32_t* msites = E.d_sc_m + 3 * (ma - E.sc_m_next.begin()); const uint32_t* pw = h->d_pow256;
        if (nm) hipLaunchKernelGGL(k_selfcheck_m, dim3((nm + 63) / 64), dim3(64), 0, st, W, msites, nm, pw, res);
        E.sc_checked += nm;
        // copy sites whose HIGHER wire lies in this window
        const auto ca = std::lower_bound(E.sc_c_hi.begin(), E.sc_c_hi.end(), (uint32_t)wire_lo), cb = std::lower_bound(ca, E.sc_c_hi.end(), (uint32_t)wire_hi);
        const uint32_t ncs = (uint32_t)(cb - ca);
        const uint32_t* csites = E.d_sc_c + 2 * (ca - E.sc_c_hi.begin());
        if (ncs) hipLaunchKernelGGL(k_selfcheck_c, dim3((ncs + 63) / 64), dim3(64), 0, st, W, csites, ncs, res);
        E.sc_checked += ncs;
    }
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(E.ev_made[slot], st));
    HIPC(hipStreamWaitEvent(E.s_copy, E.ev_made[slot], 0));
    HIPC(hipMemcpyAsync(E.h_pin[slot], E.d_win[slot], wn * 32, hipMemcpyDeviceToHost, E.s_copy));
    HIPC(hipEventRecord(E.ev_copied[slot], E.s_copy));
    HIPC(hipEventRecord(E.ev_free[slot], E.s_copy));
    return POB_OK;
}

static uint64_t fnv64(const void* p, size_t n) {
    const uint8_t* q = (const uint8_t*)p; uint64_t hsh = 1469598103934665603ull;
    // (8 bytes at a time: the map of a production witness has ~20 M entries)
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, q + i, 8); hsh = (hsh ^ v) * 1099511628211ull; }
    for (; i < n; i++) hsh = (hsh ^ q[i]) * 1099511628211ull;
    return hsh ? hsh : 1;
}

// the generation statuses of the current batch on the host (one D2H per generation): no witness is emitted for a failed input
static int emit_statuses(pob_ctx* h) {
    pob_ctx::Emit& E = h->em;
    if (E.status_gen == h->gen_count && E.status_host.size() == h->n) return POB_OK;
    HIPC(hipEventSynchronize(h->ev_gen_done));
    E.status_host.resize(h->n);
    HIPC(hipMemcpyAsync(E.status_host.data(), h->d_status, (size_t)h->n * 4, hipMemcpyDeviceToHost, own_stream(h)));
    HIPC(hipStreamSynchronize(own_stream(h)));
    E.status_gen = h->gen_count;
    return POB_OK;
}

// common part of pob_emit_begin / pob_emit_begin_reduced: E.red / E.map_id / E.total are set
static int emit_start(pob_ctx* h, uint32_t idx, uint64_t window_wires) {
    pob_ctx::Emit& E = h->em;
    const int NS = pob_ctx::Emit::NSLOT;
    // this handle's work only (a partner handle may be busy): the generation of the batch
    HIPC(hipEventSynchronize(h->ev_gen_done));
    if (h->evaluated) HIPC(hipEventSynchronize(h->ev_check_done));
    {   // like the reference binary, no witness is written for an input that failed an assert (tests/test.py:65-68)
        int rc = emit_statuses(h); if (rc) return rc;
        const uint32_t st_w = E.status_host[idx];
        if (st_w != 0) { h->err = "witness " + std::to_string(idx) + " failed an assert (status " + std::to_string(st_w) + "): nothing to emit"; E.queued_idx = -1; E.pre_made = false; return POB_E_STATE; }
    }
    if (window_wires == 0) window_wires = 8ull << 20;                       // 8 Mi wires = 256 MiB windows
    window_wires = std::min<uint64_t>(window_wires, E.total);
    const uint64_t nwin_ = (E.total + window_wires - 1) / window_wires;
    // the witness announced with pob_emit_queue, same payload and window size: its first window has been expanded (and is being copied)
    // behind the previous witness' last windows already -- continue from there, nothing to wait for
    if (E.pre_made && E.pre_made_gen == h->gen_count && E.queued_idx == (int64_t)idx && E.win_wires == window_wires && E.nwin == nwin_ && E.probe_map == E.map_id) {
        E.first_slot = (E.first_slot + (uint32_t)E.nwin) % NS;
        E.idx = idx; E.next_make = 1; E.next_take = 0; E.active = true; E.queued_idx = -1; E.pre_made = false;
        return POB_OK;
    }
    if (E.pre_made || E.queued_idx == (int64_t)idx) E.queued_idx = -1;      // (an announcement made BEFORE this begin, for the witness after this one, stays)
    E.pre_made = false;
    if (E.s_copy) HIPC(hipStreamSynchronize(E.s_copy));                    // the copies of a previous witness (an abandoned emission, a discarded first window)
    if (!E.s_copy) {
        HIPC(hipStreamCreateWithPriority(&E.s_copy, hipStreamNonBlocking, 0));
        for (int k = 0; k < NS; k++) {
            HIPC(hipEventCreateWithFlags(&E.ev_made[k], hipEventDisableTiming)); HIPC(hipEventCreateWithFlags(&E.ev_copied[k], hipEventDisableTiming));
            HIPC(hipEventCreateWithFlags(&E.ev_free[k], hipEventDisableTiming));
        }
        for (const SpongeDesc& sp : h->plan.sponges) {
            E.runs.push_back({sp.kin_w, sp.kin_b, sp.n * 1088, 0}); E.runs.push_back({sp.fin_w, sp.fin_b, sp.n * 1088, 0});
            E.runs.push_back({sp.fs_w, sp.fs_b, (sp.n + 1) * 1600, 0});
            for (uint32_t b = 0; b < sp.n; b++) E.runs.push_back({sp.abs_w + b * ABSORB_WIRES, sp.abs_b + b * ABSORB_BITS, ABSORB_WIRES, 1});
        }
        std::sort(E.runs.begin(), E.runs.end(), [](const pob_ctx::Emit::Run& a, const pob_ctx::Emit::Run& b) { return a.w < b.w; });
    }
    if (E.alloc_wires < window_wires) {                                     // (re)allocated only when a larger window is asked for: K witnesses reuse the buffers
        for (int k = 0; k < NS; k++) {
            if (E.d_win[k]) { HIPC(hipFree(E.d_win[k])); E.d_win[k] = nullptr; }
            if (E.h_pin[k]) { HIPC(hipHostFree(E.h_pin[k])); E.h_pin[k] = nullptr; }
            HIPC(hipMalloc(&E.d_win[k], window_wires * 32));
            HIPC(hipHostMalloc((void**)&E.h_pin[k], window_wires * 32, hipHostMallocDefault));
        }
        E.alloc_wires = window_wires;
    }
    if ((E.probe_win != window_wires || E.probe_map != E.map_id) && nwin_ <= 64) {
        // probe pass: every G unit runs once with the emitter's stores replaced by "mark window position / window_wires"; a window then
        // launches only the units that can write into it (most windows hold nothing but Keccak round wires)
        const size_t nu = h->plan.units.size();
        if (!E.d_probe) HIPC(hipMalloc(&E.d_probe, nu * 8));
        HIPC(hipMemsetAsync(E.d_probe, 0, nu * 8, own_stream(h)));
        GArgs A = gargs(h);
        A.emit_sel = 0; A.emit_group = idx / 64; A.emit_w0 = 0; A.emit_wn = (uint32_t)window_wires; A.emit_probe = E.d_probe; A.emit_out = nullptr;
        if (E.red) { A.emit_rbits = E.d_rbits; A.emit_rpre = E.d_rpre; }
        for (const pob_ctx::Seg& sg : h->emit_segs) { A.first = sg.first; launch_g_emit(A, sg.lds, sg.count, own_stream(h)); }
        HIPC(hipGetLastError());
        std::vector<unsigned long long> mask(nu);
        HIPC(hipMemcpyAsync(mask.data(), E.d_probe, nu * 8, hipMemcpyDeviceToHost, own_stream(h)));
        HIPC(hipStreamSynchronize(own_stream(h)));
        std::vector<uint32_t> order;
        E.wsegs.assign(nwin_, {});
        for (uint64_t wi = 0; wi < nwin_; wi++)
            for (const pob_ctx::Seg& sg : h->emit_segs) {
                pob_ctx::Emit::WSeg ws{sg.lds, (uint32_t)order.size(), 0};
                for (uint32_t j = 0; j < sg.count; j++) { const uint32_t u = h->order[sg.first + j]; if ((mask[u] >> wi) & 1) order.push_back(u); }
                ws.count = (uint32_t)order.size() - ws.first;
                if (ws.count) E.wsegs[wi].push_back(ws);
            }
        if (E.d_order) { HIPC(hipFree(E.d_order)); E.d_order = nullptr; }
        HIPC(hipMalloc(&E.d_order, std::max<size_t>(order.size(), 1) * 4));
        if (!order.empty()) HIPC(hipMemcpy(E.d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        E.probe_win = window_wires; E.probe_map = E.map_id;
    } else if (nwin_ > 64) { E.probe_win = 0; E.probe_map = E.map_id; }       // (too many windows for the probe's 64-bit masks: every unit runs for every window)
    if (E.sc_on && !E.sc_built) {
        // site-recording pass: every emitting unit once with EmitP::sites set (nothing is written); the sites are layout constants of the handle
        const uint32_t cap = std::max(h->plan.total.q, 1u);
        uint32_t* d_rec = nullptr;
        const size_t words = 4 + (size_t)cap + 3 * (size_t)cap + 2 * (size_t)cap;
        HIPC(hipMalloc(&d_rec, words * 4));
        HIPC(hipMemsetAsync(d_rec, 0, 16, own_stream(h)));
        GArgs A = gargs(h);
        A.emit_sel = idx % 64; A.emit_group = idx / 64; A.emit_w0 = 0; A.emit_wn = (uint32_t)E.total; A.emit_out = nullptr; A.emit_sites = d_rec; A.emit_sites_cap = cap;
        for (const pob_ctx::Seg& sg : h->emit_segs) { A.first = sg.first; launch_g_emit(A, sg.lds, sg.count, own_stream(h)); }
        HIPC(hipGetLastError());
        std::vector<uint32_t> rec(words);
        HIPC(hipMemcpyAsync(rec.data(), d_rec, words * 4, hipMemcpyDeviceToHost, own_stream(h)));
        HIPC(hipStreamSynchronize(own_stream(h)));
        HIPC(hipFree(d_rec));
        if (rec[0] > cap || rec[1] > cap || rec[2] > cap) { h->err = "internal: more self-check sites than derived wires"; return POB_E_STATE; }
        E.sc_z.assign(rec.begin() + 4, rec.begin() + 4 + rec[0]);
        std::sort(E.sc_z.begin(), E.sc_z.end(), [](uint32_t a, uint32_t b) { return (a & 0x7FFFFFFFu) < (b & 0x7FFFFFFFu); });
        std::vector<std::array<uint32_t, 3>> ms(rec[1]);
        for (uint32_t i = 0; i < rec[1]; i++) for (int j = 0; j < 3; j++) ms[i][j] = rec[4 + (size_t)cap + 3 * (size_t)i + j];
        std::sort(ms.begin(), ms.end());
        E.sc_ms = ms;
        E.sc_m_next.resize(ms.size());
        for (size_t i = 0; i < ms.size(); i++) E.sc_m_next[i] = ms[i][0];
        std::vector<std::array<uint32_t, 2>> cs(rec[2]);
        for (uint32_t i = 0; i < rec[2]; i++) for (int j = 0; j < 2; j++) cs[i][j] = rec[4 + 4 * (size_t)cap + 2 * (size_t)i + j];
        std::sort(cs.begin(), cs.end());
        E.sc_cs = cs;
        E.sc_c_hi.resize(cs.size());
        for (size_t i = 0; i < cs.size(); i++) E.sc_c_hi[i] = cs[i][0];
        HIPC(hipMalloc(&E.d_sc_c, std::max<size_t>(cs.size(), 1) * 8));
        if (!cs.empty()) HIPC(hipMemcpy(E.d_sc_c, cs.data(), cs.size() * 8, hipMemcpyHostToDevice));
        HIPC(hipMalloc(&E.d_sc_z, std::max<size_t>(E.sc_z.size(), 1) * 4)); HIPC(hipMalloc(&E.d_sc_m, std::max<size_t>(ms.size(), 1) * 12)); HIPC(hipMalloc(&E.d_sc_res, 16));
        if (!E.sc_z.empty()) HIPC(hipMemcpy(E.d_sc_z, E.sc_z.data(), E.sc_z.size() * 4, hipMemcpyHostToDevice));
        if (!ms.empty()) HIPC(hipMemcpy(E.d_sc_m, ms.data(), ms.size() * 12, hipMemcpyHostToDevice));
        E.sc_built = true;
    }
    if (E.sc_on) {
        const uint32_t init[3] = {0xFFFFFFFFu, 0, 0};
        HIPC(hipMemcpyAsync(E.d_sc_res, init, 12, hipMemcpyHostToDevice, own_stream(h)));
        HIPC(hipStreamSynchronize(own_stream(h)));
        E.sc_checked = 0; E.sc_skipped = 0;
    }
    if (E.sc_on && E.red && E.sc_red_map != E.map_id) {
        // the sites of this map: every wire through its class representative (pob_emit_selfcheck_alias; without one a wire stands for itself), a site with a wire
        // that is pinned to a constant or not kept is left out and counted; a copy site is a tautology between class members: counted as skipped
        const bool have_alias = E.sc_alias && E.sc_alias_n == h->plan.total.w;
        auto rep = [&](uint32_t w, uint32_t* out, bool may_be_const = true) -> bool {
            int64_t r = w;
            if (have_alias) {
                r = E.sc_alias[w];
                if (r < 0) {                                 // pinned to a constant: -1 - c for c < 2^30, INT32_MIN for a constant that does not fit
                    if (!may_be_const || r == INT32_MIN) return false;
                    *out = 0x80000000u | (uint32_t)(-1 - r); return true;
                }
            }
            if (!std::binary_search(E.keep.begin(), E.keep.end(), (uint32_t)r)) return false;
            *out = (uint32_t)r; return true;
        };
        std::vector<uint32_t> zr, mr; uint64_t left_out = E.sc_cs.size();
        for (uint32_t s0 : E.sc_z) {
            const uint32_t w = s0 & 0x7FFFFFFFu; uint32_t q[6] = {0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            bool ok = rep(w, &q[0], false) && rep(w + 1, &q[1]) && rep(w + 2, &q[2]);
            if (ok && (s0 >> 31)) ok = rep(w - 3, &q[3]) && rep(w - 2, &q[4]) && rep(w - 1, &q[5]);
            if (!ok) { left_out++; continue; }
            zr.insert(zr.end(), q, q + 6);
        }
        for (const std::array<uint32_t, 3>& m3 : E.sc_ms) {
            uint32_t q[4] = {0, 0, 0, m3[2]};
            if (!(rep(m3[0], &q[0], false) && rep(m3[0] - 1, &q[1], false) && rep(m3[1], &q[2], false))) { left_out++; continue; }
            mr.insert(mr.end(), q, q + 4);
        }
        if (E.d_sc_zr) { HIPC(hipFree(E.d_sc_zr)); E.d_sc_zr = nullptr; }
        if (E.d_sc_mr) { HIPC(hipFree(E.d_sc_mr)); E.d_sc_mr = nullptr; }
        HIPC(hipMalloc(&E.d_sc_zr, std::max<size_t>(zr.size(), 1) * 4)); HIPC(hipMalloc(&E.d_sc_mr, std::max<size_t>(mr.size(), 1) * 4));
        if (!zr.empty()) HIPC(hipMemcpy(E.d_sc_zr, zr.data(), zr.size() * 4, hipMemcpyHostToDevice));
        if (!mr.empty()) HIPC(hipMemcpy(E.d_sc_mr, mr.data(), mr.size() * 4, hipMemcpyHostToDevice));
        E.sc_red_nz = (uint32_t)(zr.size() / 6); E.sc_red_nm = (uint32_t)(mr.size() / 4); E.sc_red_host_skipped = left_out; E.sc_red_map = E.map_id;
    }
    for (int k = 0; k < NS; k++) HIPC(hipEventRecord(E.ev_free[k], E.s_copy));
    E.win_wires = window_wires; E.nwin = nwin_; E.idx = idx; E.next_make = 0; E.next_take = 0; E.first_slot = 0; E.active = true;
    for (; E.next_make < std::min<uint64_t>(1, E.nwin); E.next_make++) { int rc = emit_make_window(h, idx, E.next_make, (int)((E.first_slot + E.next_make) % NS)); if (rc) return rc; }
    return POB_OK;
}

int pob_emit_begin(pob_handle h, uint32_t idx, uint64_t window_wires) {
    if (!h) return POB_E_ARG;
    if (!h->generated || idx >= h->n) { h->err = "nothing generated / witness index out of range"; return POB_E_STATE; }
    HIPC(hipSetDevice(h->device));
    h->em.red = false; h->em.map_id = 0; h->em.total = h->plan.total.w;
    return emit_start(h, idx, window_wires);
}

int pob_emit_begin_reduced(pob_handle h, uint32_t idx, const uint32_t* keep, uint64_t n_keep, uint64_t window_wires) {
    if (!h || !keep || n_keep == 0) return POB_E_ARG;
    if (!h->generated || idx >= h->n) { h->err = "nothing generated / witness index out of range"; return POB_E_STATE; }
    HIPC(hipSetDevice(h->device));
    pob_ctx::Emit& E = h->em;
    const uint64_t W = h->plan.total.w;
    // a map the caller PINNED (pob_reduced_map_pin: same address and length, contents promised unchanged) is not hashed again; any other map is
    // hashed in full every time (86 MB / 9 ms for the production map): an address can be recycled for a different map of the same length
    const bool pinned = E.pin_id && keep == E.pin_ptr && n_keep == E.pin_n;
    const uint64_t id = pinned ? E.pin_id : (fnv64(keep, n_keep * 4) ^ (n_keep << 1));
    if (id != E.map_id || E.keep.size() != n_keep) {
        // a new map: validate (wire 0 first, strictly increasing, inside the circuit), build the bitmap and the per-word ranks, upload
        if (keep[0] != 0 || keep[n_keep - 1] >= W) { h->err = "reduced map: keep[0] must be wire 0 and every index must be < nWitness"; return POB_E_ARG; }
        const uint64_t nwords = (W + 63) / 64;
        std::vector<unsigned long long> bits(nwords, 0); std::vector<uint32_t> pre(nwords, 0);
        for (uint64_t i = 0; i < n_keep; i++) {
            if (i && keep[i] <= keep[i - 1]) { h->err = "reduced map: wire indices must be strictly increasing"; return POB_E_ARG; }
            if (keep[i] >= W) { h->err = "reduced map: every index must be < nWitness"; return POB_E_ARG; }      // (before bits[] is touched)
            bits[keep[i] >> 6] |= 1ull << (keep[i] & 63);
        }
        uint32_t acc = 0;
        for (uint64_t i = 0; i < nwords; i++) { pre[i] = acc; acc += (uint32_t)__builtin_popcountll(bits[i]); }
        if (E.s_copy) HIPC(hipStreamSynchronize(E.s_copy));
        HIPC(hipStreamSynchronize(own_stream(h)));
        if (E.d_rbits) { HIPC(hipFree(E.d_rbits)); E.d_rbits = nullptr; }
        if (E.d_rpre) { HIPC(hipFree(E.d_rpre)); E.d_rpre = nullptr; }
        HIPC(hipMalloc(&E.d_rbits, nwords * 8)); HIPC(hipMalloc(&E.d_rpre, nwords * 4));
        HIPC(hipMemcpy(E.d_rbits, bits.data(), nwords * 8, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(E.d_rpre, pre.data(), nwords * 4, hipMemcpyHostToDevice));
        E.keep.assign(keep, keep + n_keep);
        E.queued_idx = -1; E.pre_made = false;
    }
    E.red = true; E.map_id = id; E.total = n_keep;
    return emit_start(h, idx, window_wires);
}

int pob_reduced_map_pin(pob_handle h, const uint32_t* keep, uint64_t n_keep) {
    if (!h) return POB_E_ARG;
    pob_ctx::Emit& E = h->em;
    if (!keep || n_keep == 0) { E.pin_ptr = nullptr; E.pin_n = 0; E.pin_id = 0; return POB_OK; }     // unpin
    E.pin_ptr = keep; E.pin_n = n_keep; E.pin_id = fnv64(keep, n_keep * 4) ^ (n_keep << 1);
    return POB_OK;
}

int pob_emit_next(pob_handle h, const uint8_t** data, uint64_t* first_wire, uint64_t* n_wires) {
    if (!h || !data || !first_wire || !n_wires) return POB_E_ARG;
    pob_ctx::Emit& E = h->em;
    if (!E.active) { h->err = "pob_emit_next without pob_emit_begin"; return POB_E_STATE; }
    if (E.next_take == E.nwin) { E.active = false; *data = nullptr; *first_wire = E.total; *n_wires = 0; return POB_OK; }
    HIPC(hipSetDevice(h->device));
    const int NS = pob_ctx::Emit::NSLOT;
    // the window handed out by the previous call is released now.  Two windows beyond the one returned here are kept in the making: one is
    // being copied while the caller works on this one, the next one is being expanded
    while (E.next_make < E.nwin && E.next_make <= E.next_take + 2) {
        int rc = emit_make_window(h, E.idx, E.next_make, (int)((E.first_slot + E.next_make) % NS)); if (rc) return rc;
        E.next_make++;
    }
    // ... and when this witness has no more windows to make, the first window of the witness announced with pob_emit_queue
    if (E.next_make == E.nwin && E.nwin - E.next_take <= 2 && E.queued_idx >= 0 && !E.pre_made) {
        int rc = emit_make_window(h, (uint32_t)E.queued_idx, 0, (int)((E.first_slot + E.nwin) % NS)); if (rc) return rc;
        E.pre_made = true; E.pre_made_gen = h->gen_count;
    }
    const uint64_t k = E.next_take++;
    const int slot = (int)((E.first_slot + k) % NS);
    HIPC(hipEventSynchronize(E.ev_copied[slot]));
    *data = E.h_pin[slot]; *first_wire = k * E.win_wires; *n_wires = std::min(E.win_wires, E.total - k * E.win_wires);
    return POB_OK;
}

int pob_emit_selfcheck(pob_handle h, int enable) {
    if (!h) return POB_E_ARG;
    h->em.sc_on = enable != 0;
    h->em.queued_idx = -1; h->em.pre_made = false;       // (a window pre-made without the check is not part of a checked emission)
    return POB_OK;
}

int pob_emit_selfcheck_alias(pob_handle h, const int32_t* alias, uint64_t n_wires) {
    if (!h || (alias && n_wires != h->plan.total.w)) return POB_E_ARG;
    h->em.sc_alias = alias; h->em.sc_alias_n = alias ? n_wires : 0; h->em.sc_red_map = 0;       // (the lists of the next reduced emission are rebuilt)
    return POB_OK;
}

int pob_emit_selfcheck_result(pob_handle h, uint64_t* checked, uint64_t* skipped, uint32_t* first_bad_wire) {
    if (!h) return POB_E_ARG;
    pob_ctx::Emit& E = h->em;
    if (!E.sc_on || !E.sc_built) { h->err = "no self-checked emission yet (pob_emit_selfcheck, then an emission)"; return POB_E_STATE; }
    HIPC(hipSetDevice(h->device));
    HIPC(hipStreamSynchronize(own_stream(h)));
    uint32_t res[3];
    HIPC(hipMemcpy(res, E.d_sc_res, 12, hipMemcpyDeviceToHost));
    const uint64_t all = E.sc_z.size() + E.sc_m_next.size() + E.sc_c_hi.size();
    const uint64_t done = E.red ? (uint64_t)res[2] : E.sc_checked - res[1];          // (reduced: the kernels count what they evaluate)
    if (checked) *checked = done;
    if (skipped) *skipped = all - done;
    if (first_bad_wire) *first_bad_wire = res[0];
    return POB_OK;
}

int pob_emit_queue(pob_handle h, uint32_t next_idx) {
    if (!h) return POB_E_ARG;
    pob_ctx::Emit& E = h->em;
    if (!h->generated) { h->err = "nothing generated"; return POB_E_STATE; }
    if (next_idx >= h->n) { h->err = "witness index out of range"; return POB_E_STATE; }
    if (E.pre_made) return POB_OK;                      // (a first window is on its way already: one announcement at a time)
    if (E.sc_on) return POB_OK;                         // self-checked emissions are not overlapped: the check's result names ONE witness (pob_emit_selfcheck_result)
    HIPC(hipSetDevice(h->device));
    { int rc = emit_statuses(h); if (rc) return rc; }
    if (E.status_host[next_idx] != 0) { h->err = "witness " + std::to_string(next_idx) + " failed an assert: nothing to emit"; return POB_E_STATE; }
    E.queued_idx = next_idx;
    return POB_OK;
}

int pob_emit_witness(pob_handle h, uint32_t idx, uint8_t* dst, uint64_t cap) {
    if (!h || !dst) return POB_E_ARG;
    const uint64_t bytes = (uint64_t)h->plan.total.w * 32;
    if (cap < bytes) { h->err = "destination too small"; return POB_E_ARG; }
    int rc = pob_emit_begin(h, idx, 0);
    if (rc) return rc;
    for (;;) {
        const uint8_t* p; uint64_t w0, wn;
        rc = pob_emit_next(h, &p, &w0, &wn);
        if (rc) return rc;
        if (!wn) break;
        memcpy(dst + w0 * 32, p, wn * 32);
    }
    return POB_OK;
}

// iden3 .wtns container around the window stream ("wtns" v2, 2 sections: (1) n8 = 32, prime, nWitness  (2) values); the emission is begun
static int write_wtns_stream(pob_ctx* h, const char* path) {
    const uint64_t W = h->em.total, bytes = W * 32;
    FILE* f = fopen(path, "wb");
    if (!f) { h->err = std::string("cannot open ") + path; h->em.active = false; return POB_E_IO; }
    uint8_t hdr[76];
    const uint64_t P64[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    uint32_t u32; uint64_t u64v;
    memcpy(hdr, "wtns", 4); u32 = 2; memcpy(hdr + 4, &u32, 4); memcpy(hdr + 8, &u32, 4);
    u32 = 1; memcpy(hdr + 12, &u32, 4); u64v = 40; memcpy(hdr + 16, &u64v, 8);
    u32 = 32; memcpy(hdr + 24, &u32, 4); memcpy(hdr + 28, P64, 32); u32 = (uint32_t)W; memcpy(hdr + 60, &u32, 4);
    u32 = 2; memcpy(hdr + 64, &u32, 4); u64v = bytes; memcpy(hdr + 68, &u64v, 8);
    bool ok = fwrite(hdr, 1, 76, f) == 76;
    for (;;) {
        const uint8_t* p; uint64_t w0, wn;
        int rc = pob_emit_next(h, &p, &w0, &wn);
        if (rc) { fclose(f); remove(path); return rc; }
        if (!wn) break;
        if (ok) ok = fwrite(p, 1, wn * 32, f) == wn * 32;
    }
    fclose(f);
    if (!ok) { remove(path); h->err = "write failed"; return POB_E_IO; }
    return POB_OK;
}
int pob_write_wtns(pob_handle h, uint32_t idx, const char* path) {
    if (!h || !path) return POB_E_ARG;
    int rc = pob_emit_begin(h, idx, 0);
    return rc ? rc : write_wtns_stream(h, path);
}
int pob_write_wtns_reduced(pob_handle h, uint32_t idx, const uint32_t* keep, uint64_t n_keep, const char* path) {
    if (!h || !path) return POB_E_ARG;
    int rc = pob_emit_begin_reduced(h, idx, keep, n_keep, 0);
    return rc ? rc : write_wtns_stream(h, path);
}

// emission throughput: K witnesses back to back through the window pipeline, the windows only touched (first + last cache line).
// keep != NULL: the reduced form.  The buffers (two device windows, two pinned host windows, the probe tables) are set up by the
// first witness a handle emits at a window size: time a first call to see that cost, a second one for the steady state.
int pob_emit_measure_ex(pob_handle h, uint32_t first_idx, uint32_t count, uint64_t window_wires, const uint32_t* keep, uint64_t n_keep, double* seconds, uint64_t* bytes) {
    if (!h || !seconds || !bytes || count == 0) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    uint64_t total = 0; volatile uint8_t sink = 0;
    timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t i = 0; i < count; i++) {
        int rc = keep ? pob_emit_begin_reduced(h, first_idx + i, keep, n_keep, window_wires) : pob_emit_begin(h, first_idx + i, window_wires);
        if (rc) return rc;
        if (i + 1 < count) { rc = pob_emit_queue(h, first_idx + i + 1); if (rc) return rc; }
        for (;;) {
            const uint8_t* p; uint64_t w0, wn;
            rc = pob_emit_next(h, &p, &w0, &wn);
            if (rc) return rc;
            if (!wn) break;
            sink = sink ^ p[0] ^ p[wn * 32 - 1]; total += wn * 32;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec); *bytes = total;
    return POB_OK;
}
int pob_emit_measure(pob_handle h, uint32_t first_idx, uint32_t count, uint64_t window_wires, double* seconds, uint64_t* bytes) {
    return pob_emit_measure_ex(h, first_idx, count, window_wires, nullptr, 0, seconds, bytes);
}

// Measurement inside a running job: enable = 1 makes every following pob_constraint_check record HIP events (on the stream the kernel is
// launched on) around its Keccak round evaluation kernel; ms (may be NULL) = the duration of the LAST such kernel, which must have
// completed (e.g. its batch's results are in).  enable = 0 stops recording.
int pob_probe_check_kernel(pob_handle h, int enable, float* ms) {
    if (!h) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    if (ms) {
        // the newest pair whose kernel has run: this slot's, or -- the next generation is already enqueued -- the other one's
        uint32_t sl = h->rec_slot;
        if (!h->kchk_rec[sl] || hipEventQuery(h->ev_kchk[sl][1]) != hipSuccess) sl ^= 1;
        if (!h->ev_kchk[sl][0] || !h->kchk_rec[sl]) { h->err = "no timed evaluation yet"; return POB_E_STATE; }
        HIPC(hipEventSynchronize(h->ev_kchk[sl][1]));
        HIPC(hipEventElapsedTime(ms, h->ev_kchk[sl][0], h->ev_kchk[sl][1]));
    }
    if (enable && !h->ev_kchk[0][0]) for (auto& pr : h->ev_kchk) for (hipEvent_t& e : pr) HIPC(hipEventCreate(&e));
    if (!enable) { for (auto& pr : h->ev_kchk) for (hipEvent_t& e : pr) if (e) { hipEventDestroy(e); e = nullptr; } h->kchk_rec[0] = h->kchk_rec[1] = false; }
    return POB_OK;
}

int pob_time_kernel(pob_handle h, int which, int iters, void* stream_, float* avg_ms) {
    if (!h || !h->generated || iters < 1 || !avg_ms) return POB_E_STATE;
    HIPC(hipSetDevice(h->device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : own_stream(h);
    const uint32_t G = (h->n + 63) / 64;
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    KArgs K = kargs(h); K.first = 0;
    GArgs A = gargs(h);
    uint32_t* d_sel = nullptr; uint32_t nsel = 0, sel_cls = 0, sel_fam = 0;
    if (which >= 100 && which < 300) {   // 100 + kind: evaluation, 200 + kind: generation of all units of one kind, alone on the device
        const uint32_t kind = (uint32_t)which % 100, need = which >= 200 ? UNIT_GEN : UNIT_CHECK;
        std::vector<uint32_t> sel;
        for (uint32_t u = 0; u < h->plan.units.size(); u++) if (h->plan.units[u].kind == kind && (h->plan.units[u].flags & need)) sel.push_back(u);
        if (sel.empty()) { *avg_ms = 0; hipEventDestroy(e0); hipEventDestroy(e1); return POB_OK; }
        nsel = (uint32_t)sel.size(); sel_cls = unit_class(kind, which >= 200); sel_fam = fam_of(kind);
        HIPC(hipMalloc(&d_sel, nsel * 4)); HIPC(hipMemcpy(d_sel, sel.data(), nsel * 4, hipMemcpyHostToDevice));
        A.order = d_sel; A.first = 0;
    }
    HIPC(hipEventRecord(e0, st));
    for (int it = 0; it < iters; it++) {
        if (which >= 300) {          // 300 + family: that family's evaluation kernel alone
            for (const pob_ctx::Seg& sg : h->chk_segs) if (sg.lds == (uint32_t)which - 300) { A.first = sg.first; launch_g_check(A, sg.lds, sg.count, G, st); }
        } else if (which >= 200) launch_g_gen(A, sel_cls, nsel, G, st);
        else if (which >= 100) launch_g_check(A, sel_fam, nsel, G, st);
        else if (which == 0) launch_k_rounds(K, false, h->nperms, G, st);
        else if (which == 1) launch_k_rounds(K, true, h->nperms, G, st);
        else if (which == 6) launch_k_rounds_gc(K, h->nperms, G, false, st);
        else if (which == 2) {
            for (const pob_ctx::Seg& sg : h->chk_segs) {
                A.first = sg.first;
                launch_g_check(A, sg.lds, sg.count, G, st);
            }
        } else launch_k_chain(K, false, (uint32_t)h->plan.sponges.size(), G, st);
    }
    HIPC(hipEventRecord(e1, st));
    HIPC(hipEventSynchronize(e1));
    float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (d_sel) hipFree(d_sel);
    return POB_OK;
}

int pob_debug_stream_create(int device, const uint32_t* cu_mask, uint32_t words, void** stream) {
    if (!cu_mask || !words || !stream) return POB_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return POB_E_HIP;
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, words, cu_mask) != hipSuccess) return POB_E_HIP;
    *stream = (void*)st;
    return POB_OK;
}
void pob_debug_stream_destroy(int device, void* stream) { if (stream && hipSetDevice(device) == hipSuccess) hipStreamDestroy((hipStream_t)stream); }

int pob_debug_store_fault(pob_handle h, int cls, uint32_t group, uint64_t index, uint64_t mask, uint32_t* wire) {
    if (!h || group >= h->groups) return POB_E_ARG;
    if (!h->inorder || !h->gc) { h->err = "pob_debug_store_fault: the calculator's generation carries no evaluation (pob_set_inorder bit 2)"; return POB_E_STATE; }
    bool found = false;
    if (cls == POB_CLASS_BIT) {
        if (index >= h->plan.total.b) return POB_E_ARG;
        for (const SpongeDesc& sp : h->plan.sponges) {
            if (index < sp.abs_b || index >= sp.abs_b + (uint64_t)sp.n * ABSORB_BITS) continue;
            const uint32_t b = (uint32_t)((index - sp.abs_b) / ABSORB_BITS), o = (uint32_t)((index - sp.abs_b) % ABSORB_BITS);
            if (o < AB_DIRECT) break;          // (the sponge chain's own words: their evaluation is a launch of pob_constraint_check, k_chain_check)
            if (wire) *wire = sp.abs_w + b * ABSORB_WIRES + AB_DIRECT + (o - AB_DIRECT) / KR_BITS * KECCAKF_ROUND_WIRES;     // (AB_DIRECT: every wire of the block ahead of the round blocks is stored, so it is their wire offset too)
            found = true; break;
        }
    } else if (cls == POB_CLASS_SM && h->circuit == POB_CIRCUIT_PROOF_OF_BURN) {
        const SmRef r0 = h->plan.L.pm.numLeafAddressNibbles;
        if (index >= r0.i && index < r0.i + h->plan.nsm_in) { found = true; if (wire) *wire = r0.w + (uint32_t)(index - r0.i); }
    }
    // anything else: a store of a G unit (policy.hpp GenPT<true, true> / poseidon_wide.hpp): armed without knowing the wire -- or whether a riding unit stores the word at all
    // (the sponge chains' words, the RLP units' wires and the words no unit of this instantiation writes are not): *wire = 0xFFFFFFFF
    h->fault_g = !found;
    if (!found) {
        const Cur t = h->plan.total;
        if (!(h->circuit == POB_CIRCUIT_PROOF_OF_BURN || h->circuit == POB_CIRCUIT_SPEND) || (cls == POB_CLASS_BIT ? index >= t.b : cls == POB_CLASS_SM ? index >= t.s : cls == POB_CLASS_FR ? index >= t.f : true)) {
            h->err = "pob_debug_store_fault: not a stored word whose evaluation rides with its generation"; return POB_E_ARG;
        }
        if (wire) *wire = 0xFFFFFFFFu;
    }
    h->fault_armed = true; h->fault_cls = cls; h->fault_group = group; h->fault_word = index; h->fault_mask = mask;
    return POB_OK;
}

int pob_debug_xor_bits(pob_handle h, uint32_t group, uint64_t bit_index, uint64_t mask) {
    if (!h || group >= h->groups || bit_index >= h->plan.total.b) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    h->rode = false;
    hipLaunchKernelGGL(k_xor_word, dim3(1), dim3(1), 0, own_stream(h), h->d_bits + (uint64_t)group * h->plan.total.b + bit_index, mask);
    HIPC(hipStreamSynchronize(own_stream(h)));
    return POB_OK;
}

int pob_debug_poke(pob_handle h, int cls, uint32_t group, uint64_t index, uint32_t sub, uint32_t lane, uint32_t xor_mask) {
    if (!h || group >= h->groups || lane >= 64) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    h->rode = false;
    const Cur t = h->plan.total;
    if (cls == POB_CLASS_BIT) {
        if (index >= t.b) return POB_E_ARG;
        hipLaunchKernelGGL(k_xor_word, dim3(1), dim3(1), 0, own_stream(h), h->d_bits + (uint64_t)group * t.b + index, (uint64_t)(xor_mask & 1u) << lane);
    } else if (cls == POB_CLASS_SM) {
        if (index >= t.s) return POB_E_ARG;
        hipLaunchKernelGGL(k_xor_u32, dim3(1), dim3(1), 0, own_stream(h), (uint32_t*)h->d_sm + ((uint64_t)group * t.s + index) * 64 + lane, xor_mask);
    } else if (cls == POB_CLASS_FR) {
        if (index >= t.f || sub >= 8) return POB_E_ARG;
        hipLaunchKernelGGL(k_xor_u32, dim3(1), dim3(1), 0, own_stream(h), h->d_fr + ((uint64_t)group * t.f + index) * 512 + sub * 64 + lane, xor_mask);
    } else return POB_E_ARG;
    HIPC(hipStreamSynchronize(own_stream(h)));
    return POB_OK;
}

// the two field inversions of the device code (Kaliski almost-inverse: fr_inv; Fermat ladder: fr_inv_fermat, the emitter's fall-back beyond its table of
// small inverses) on n canonical 32-byte LE inputs < p; 0 -> 0
int pob_debug_fr_inv(int device, const uint8_t* in, uint32_t n, uint8_t* out_kaliski, uint8_t* out_fermat) {
    if (!in || !out_kaliski || !out_fermat || n == 0) return POB_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return POB_E_HIP;
    uint32_t *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
    const size_t bytes = (size_t)n * 32;
    int rc = POB_E_HIP;
    if (hipMalloc(&d_in, bytes) == hipSuccess && hipMalloc(&d_a, bytes) == hipSuccess && hipMalloc(&d_b, bytes) == hipSuccess &&
        hipMemcpy(d_in, in, bytes, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(k_fr_inv_test, dim3((n + 63) / 64), dim3(64), 0, 0, d_in, d_a, d_b, n);
        if (hipMemcpy(out_kaliski, d_a, bytes, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(out_fermat, d_b, bytes, hipMemcpyDeviceToHost) == hipSuccess) rc = POB_OK;
    }
    if (d_in) hipFree(d_in); if (d_a) hipFree(d_a); if (d_b) hipFree(d_b);
    return rc;
}

// which inverse paths of the emitter have run since the last reset (policy.hpp EmitP::ctr); the emission must be complete (pob_emit_next returned its last window)
int pob_debug_emit_counters(pob_handle h, uint64_t out[4], int reset) {
    if (!h || !out) return POB_E_ARG;
    HIPC(hipSetDevice(h->device));
    uint32_t c[4];
    HIPC(hipStreamSynchronize(own_stream(h)));
    HIPC(hipMemcpy(c, h->d_emit_ctr, sizeof c, hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) out[k] = c[k];
    if (reset) HIPC(hipMemset(h->d_emit_ctr, 0, sizeof c));
    return POB_OK;
}

// where a few named wires live (test hook for the evaluator's corruption tests): storage class, rank within the class, wire index
int pob_debug_ref(pob_handle h, const char* name, uint32_t k, int* cls, uint64_t* index, uint64_t* wire) {
    if (!h || !name || !cls || !index || !wire) return POB_E_ARG;
    const Plan& pl = h->plan; const CircuitLayout& L = pl.L;
    const std::string n = name;
    auto set = [&](int c, uint64_t i, uint64_t w) { *cls = c; *index = i; *wire = w; return POB_OK; };
    if (n == "poseidon") {            // k-th wire of the first Poseidon block (every wire of it is an FR wire)
        for (const UnitDesc& u : pl.units) if (u.kind == CK_POS_SEG) {
            if (k >= pos_wires((int)u.a[0], pos_off((int)u.a[0]).rp)) return POB_E_ARG;
            return set(POB_CLASS_FR, u.cur.f + k, u.cur.w + k);
        }
        return POB_E_ARG;
    }
    if (n == "commitment") { const FrRef r = h->circuit == POB_CIRCUIT_PROOF_OF_BURN ? L.pm.commitment : L.sm.commitment; return set(POB_CLASS_FR, r.i, r.w); }
    if (n == "kb.inLen") { if (k >= L.nkb) return POB_E_ARG; return set(POB_CLASS_SM, L.kbs[k].inLen.i, L.kbs[k].inLen.w); }
    if (n == "pad.div.out" || n == "pad.div.rem") {       // of KeccakBytes instance k
        if (k >= L.nkb) return POB_E_ARG;
        const KBRefs& r = L.kbs[k];
        const uint32_t o = n == "pad.div.rem" ? 1 : 0;
        return set(POB_CLASS_SM, r.c_div.s + o, r.c_div.w + o);
    }
    if (h->circuit == POB_CIRCUIT_PROOF_OF_BURN && L.nsc > 1) {                  // SubstringCheck of layer 1 (substring_check.circom:45-49, :91)
        const ScRefs& sc = L.scs[1];
        const uint32_t mm = 136u * (uint32_t)L.pob.NB, kk = mm - 31 + 1;
        if (n == "sc.exists" && k < kk) return set(POB_CLASS_BIT, sc.ex.i + k, sc.ex.w + k);
    }
    return POB_E_ARG;
}

// ---- host Keccak-256 (FIPS-202 permutation, original 0x01 padding) for input producers
static const int KROT_H[25] = {1, 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
static uint64_t rotl64h(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
static void keccak_f_host(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) { uint64_t d = c[(x + 4) % 5] ^ rotl64h(c[(x + 1) % 5], 1); for (int y = 0; y < 25; y += 5) a[x + y] ^= d; }
        b[0] = a[0];
        for (int i = 0; i < 24; i++) b[KROT_H[i + 1]] = rotl64h(a[KROT_H[i]], ((i + 1) * (i + 2) / 2) % 64);
        for (int i = 0; i < 25; i++) { int y = i / 5 * 5; a[i] = b[i] ^ (~b[y + (i + 1) % 5] & b[y + (i + 2) % 5]); }
        a[0] ^= RC[r];
    }
}
void pob_keccak256(const uint8_t* msg, uint64_t len, uint8_t out[32]) {
    uint64_t st[25]; memset(st, 0, sizeof st);
    uint8_t blk[136];
    for (uint64_t off = 0;; off += 136) {
        uint64_t n = len - off < 136 ? len - off : 136;
        memset(blk, 0, 136); if (n) memcpy(blk, msg + off, n);
        const bool last = n < 136;
        if (last) { blk[n] ^= 0x01; blk[135] ^= 0x80; }
        for (int i = 0; i < 17; i++) { uint64_t v; memcpy(&v, blk + 8 * i, 8); st[i] ^= v; }
        keccak_f_host(st);
        if (last) break;
    }
    memcpy(out, st, 32);
}

// Proof-of-work search of the input producer (reference tests/main.py:47-56): smallest key >= start (big-endian 256-bit
// counter) such that keccak256(key | postfix) starts with `zero_bytes` zero bytes.  Host helper; returns tries or -1.
int64_t pob_pow_search(const uint8_t start_key[32], const uint8_t* postfix, uint32_t postfix_len, uint32_t zero_bytes, uint64_t max_tries, uint8_t out_key[32]) {
    if (postfix_len > 100 || zero_bytes > 8) return -1;
    uint8_t msg[136];
    memcpy(msg, start_key, 32); memcpy(msg + 32, postfix, postfix_len);
    const uint32_t len = 32 + postfix_len;
    for (uint64_t t = 0; t < max_tries; t++) {
        uint8_t blk[136]; memset(blk, 0, 136); memcpy(blk, msg, len); blk[len] ^= 0x01; blk[135] ^= 0x80;
        uint64_t st[25]; memset(st, 0, sizeof st);
        for (int i = 0; i < 17; i++) { uint64_t v; memcpy(&v, blk + 8 * i, 8); st[i] = v; }
        keccak_f_host(st);
        const uint64_t mask = zero_bytes == 8 ? ~0ULL : ((1ULL << (8 * zero_bytes)) - 1);
        if ((st[0] & mask) == 0) { memcpy(out_key, msg, 32); return (int64_t)t; }
        for (int i = 31; i >= 0; i--) if (++msg[i]) break;      // big-endian increment
    }
    return -1;
}

// The same search on the GPU: windows of 2^24 candidate keys, first window with a hit wins and its smallest offset is returned,
// so the result equals the sequential search's.  Returns the number of increments or -1 (exhausted / bad arguments), -2 on a HIP error.
int64_t pob_pow_search_gpu(int device, const uint8_t start_key[32], const uint8_t* postfix, uint32_t postfix_len, uint32_t zero_bytes,
                           uint64_t max_tries, uint8_t out_key[32]) {
    if (postfix_len > 100 || zero_bytes > 8 || zero_bytes == 0) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    uint8_t blk[136]; memset(blk, 0, 136);
    memcpy(blk + 32, postfix, postfix_len);
    const uint32_t len = 32 + postfix_len;
    blk[len] ^= 0x01; blk[135] ^= 0x80;
    uint64_t tail[13];
    for (int i = 0; i < 13; i++) memcpy(&tail[i], blk + 32 + 8 * i, 8);
    uint64_t kw[4];
    for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = (v << 8) | start_key[8 * i + j]; kw[i] = v; }
    uint64_t* d_tail = nullptr; unsigned long long* d_found = nullptr;
    if (hipMalloc(&d_tail, sizeof tail) != hipSuccess || hipMalloc(&d_found, 8) != hipSuccess) return -2;
    hipMemcpy(d_tail, tail, sizeof tail, hipMemcpyHostToDevice);
    const uint64_t mask = zero_bytes == 8 ? ~0ULL : ((1ULL << (8 * zero_bytes)) - 1);
    const uint64_t WIN = 1ULL << 24;
    int64_t result = -1;
    for (uint64_t base = 0; base < max_tries; base += WIN) {
        const uint64_t cnt = max_tries - base < WIN ? max_tries - base : WIN;
        unsigned long long init = ~0ULL;
        hipMemcpy(d_found, &init, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_pow_search, dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, 0, kw[0], kw[1], kw[2], kw[3], d_tail, base, cnt, mask, d_found);
        unsigned long long got = ~0ULL;
        if (hipMemcpy(&got, d_found, 8, hipMemcpyDeviceToHost) != hipSuccess) { result = -2; break; }
        if (got != ~0ULL) { result = (int64_t)got; break; }
    }
    hipFree(d_tail); hipFree(d_found);
    if (result >= 0) {           // key = start + result (256-bit big-endian)
        uint64_t w[4] = {kw[0], kw[1], kw[2], kw[3]};
        uint64_t add = (uint64_t)result;
        for (int i = 3; i >= 0 && add; i--) { const uint64_t o = w[i]; w[i] += add; add = w[i] < o; }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) out_key[8 * i + j] = (uint8_t)(w[i] >> (56 - 8 * j));
    }
    return result;
}

}  // extern "C"
