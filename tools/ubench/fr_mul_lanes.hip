// Micro-kernel: the BN254 Montgomery product with the LIMBS SPREAD OVER LANES -- 8 lanes per field element, lane k holds limb k, carries and broadcasts by DPP moves --
// (BASELINE.json north_star: "4x64-bit-limb Montgomery field mul using wavefront shuffle for carries"; here 8 x 32-bit limbs: CDNA4's integer multiplier is 32-bit) against
// the shipped product (proof_of_burn_amd/csrc/fr_dev.hpp fr_mul: one element per lane, 8 limbs in 8 VGPRs, product scanning).  Built to be MEASURED:
//   * correctness: 8 elements per wavefront through a chain of products, compared with fr_mul on the same values;
//   * latency of a dependent chain for a lone wavefront per SIMD (what a Poseidon block pays: 304 products deep), and with 2 / 4 wavefronts per SIMD;
//   * throughput: field products per microsecond of the whole device, both forms.
//   hipcc --offload-arch=gfx950 -O3 -I proof_of_burn_amd/csrc tools/ubench/fr_mul_lanes.hip -o /tmp/fr_mul_lanes && /tmp/fr_mul_lanes
// Word-serial CIOS over b, lane-parallel over a and p: per step i  t += a_k * b_i (b_i broadcast inside the 8-lane group: quad_perm + row_shr/shl:4 under a bank mask),
// m = t_0 * (-p^-1) broadcast the same way, t += m * p_k, then t / 2^32: lane k takes the high part of its own accumulator and the low word of lane k + 1 (row_shl:1).
// The result is < 2p in redundant limbs; seven ripple steps make the limbs canonical 32-bit words again (operands of the next product).  R = 2^256 > 4p, so values < 2p
// chain without a conditional subtraction (done once, where a wire is stored).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include "fr_dev.hpp"

#define MADC(acc, hi, x, y) asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc")
template <int I> __device__ __forceinline__ uint32_t bcast8(uint32_t v) {        // every lane of an 8-lane group := the group's lane I
    constexpr int q = I & 3, qp = q | (q << 2) | (q << 4) | (q << 6);
    uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, qp, 0xF, 0xF, false);                   // quad_perm: each quad := its lane q
    if (I < 4) x = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xF, 0xA, false);              // row_shr:4 into the odd quads (lanes 4-7, 12-15)
    else x = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);                    // row_shl:4 into the even quads
    return x;
}
// a, b: limb k of the two operands in lane k of the group (values < 2p); pk: limb k of p; k = lane & 7
__device__ __forceinline__ uint32_t fr_mul_lanes(uint32_t a, uint32_t b, uint32_t pk, uint32_t k) {
    uint64_t acc = 0; uint32_t hi = 0;
#define STEP(I) { \
        const uint32_t bi = bcast8<I>(b); \
        MADC(acc, hi, a, bi); \
        const uint32_t m = bcast8<0>((uint32_t)acc) * FR_NINV32; \
        MADC(acc, hi, m, pk); \
        uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)acc, 0x101, 0xF, 0xF, false);   /* row_shl:1: lane k := lane k + 1's low word */ \
        if (k == 7) nxt = 0; \
        acc = ((acc >> 32) | ((uint64_t)hi << 32)) + nxt; hi = 0; }
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
#undef STEP
#pragma unroll
    for (int s = 0; s < 7; s++) {                                                                             // ripple: limbs back to 32 bits
        uint32_t c = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(acc >> 32), 0x111, 0xF, 0xF, false);    // row_shr:1: lane k := lane k - 1's carry
        if (k == 0) c = 0;
        acc = (acc & 0xFFFFFFFFull) + c;
    }
    return (uint32_t)acc;
}

__global__ void k_lanes(uint32_t* io, int n) {          // io[wave][64]: limb k of element e of the wave at lane 8e + k; x <- x * y, n times (y = the initial x)
    const uint32_t P[8] = FR_P_LIMBS;
    const uint32_t lane = threadIdx.x, k = lane & 7;
    uint32_t pk = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) if (k == (uint32_t)j) pk = P[j];
    uint32_t x = io[blockIdx.x * 64 + lane]; const uint32_t y = x;
    for (int i = 0; i < n; i++) x = fr_mul_lanes(x, y, pk, k);
    io[blockIdx.x * 64 + lane] = x;
}
__global__ void k_ref(Fr* io, int n) {
    Fr x = io[blockIdx.x * 64 + threadIdx.x]; const Fr y = x;
    for (int i = 0; i < n; i++) x = fr_mul(x, y);
    io[blockIdx.x * 64 + threadIdx.x] = x;
}

static bool geq_p(const uint32_t* a) { const uint32_t P[8] = FR_P_LIMBS; for (int i = 7; i >= 0; i--) { if (a[i] != P[i]) return a[i] > P[i]; } return true; }
int main() {
    int clk_khz; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    // ---- correctness: 8 waves x 8 elements, 37 chained products, against fr_mul
    const int NW = 8, NCH = 37;
    std::vector<uint32_t> h(NW * 64); std::vector<Fr> r(NW * 64);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int w = 0; w < NW; w++) for (int e = 0; e < 8; e++) {
        Fr v; for (int j = 0; j < 8; j++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.l[j] = (uint32_t)s; }
        v.l[7] &= 0x1FFFFFFFu;                                        // < 2^253 < p
        for (int j = 0; j < 8; j++) h[w * 64 + 8 * e + j] = v.l[j];
        r[w * 64 + e] = v;
    }
    uint32_t* d; Fr* dr; hipMalloc(&d, h.size() * 4); hipMalloc(&dr, r.size() * sizeof(Fr));
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, r.data(), r.size() * sizeof(Fr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_lanes, dim3(NW), dim3(64), 0, 0, d, NCH); hipLaunchKernelGGL(k_ref, dim3(NW), dim3(64), 0, 0, dr, NCH);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, r.size() * sizeof(Fr), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < NW; w++) for (int e = 0; e < 8; e++) {
        uint32_t v[8]; for (int j = 0; j < 8; j++) v[j] = h[w * 64 + 8 * e + j];
        if (geq_p(v)) { const uint32_t P[8] = FR_P_LIMBS; uint64_t br = 0; for (int j = 0; j < 8; j++) { uint64_t dd = (uint64_t)v[j] - P[j] - br; v[j] = (uint32_t)dd; br = (dd >> 63) & 1; } }     // (the lane form returns < 2p)
        for (int j = 0; j < 8; j++) if (v[j] != r[w * 64 + e].l[j]) { bad++; break; }
    }
    printf("correctness: %d chained products, %d elements: %s\n", NCH, NW * 8, bad ? "MISMATCH" : "equal to fr_mul");
    // ---- latency / throughput
    const int n = 2000;
    uint32_t* big; Fr* bigr; const size_t WV = 256 * 4 * 4;
    hipMalloc(&big, WV * 64 * 4); hipMemset(big, 1, WV * 64 * 4); hipMalloc(&bigr, WV * 64 * sizeof(Fr)); hipMemset(bigr, 1, WV * 64 * sizeof(Fr));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int form = 0; form < 2; form++) for (int w : {1, 2, 4}) {
        const int blocks = 256 * 4 * w;
        if (form) hipLaunchKernelGGL(k_lanes, dim3(blocks), dim3(64), 0, 0, big, 10); else hipLaunchKernelGGL(k_ref, dim3(blocks), dim3(64), 0, 0, bigr, 10);
        hipEventRecord(e0);
        if (form) hipLaunchKernelGGL(k_lanes, dim3(blocks), dim3(64), 0, 0, big, n); else hipLaunchKernelGGL(k_ref, dim3(blocks), dim3(64), 0, 0, bigr, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us_per = ms * 1e3 / n, elems = (double)blocks * (form ? 8 : 64);
        printf("%-34s %d wavefront(s) per SIMD: %.3f us per dependent product (%.0f cycles); device throughput %.0f field products per us\n",
               form ? "limbs over lanes (8 lanes/element)" : "fr_mul (element per lane)", w, us_per, us_per * 1e-6 * clk_khz * 1e3, elems / us_per);
    }
    return bad != 0;
}
