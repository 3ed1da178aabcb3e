// Micro-benchmark: issue cost (cycles per wave64 instruction) of the instructions a BN254 product could be made of -- the 32-bit integer multiplier against the
// double-precision FMA pipe -- and the latency of the shipped fr_mul in a dependent chain (one wavefront per SIMD, and 2 / 4).
//   hipcc --offload-arch=gfx950 -O3 -I proof_of_burn_amd/csrc tools/ubench/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fr_dev.hpp"

#define REP 1024
template <int WHICH, int CHAINS> __global__ void k(uint64_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t c[4] = {threadIdx.x, 1, 2, 3};
    double d[4] = {1.0 + threadIdx.x, 2.0, 3.0, 4.0}; const double da = 1.0000001, db = 0.5;
    uint32_t x[4] = {a, b, a ^ b, a + b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            const int q = r % CHAINS;
            if (WHICH == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[q]) : "v"(a), "v"(b) : "vcc");
            else if (WHICH == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[q]) : "v"(c[(q + 1) % 4]));
            else if (WHICH == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[q]) : "v"(da), "v"(db));
            else if (WHICH == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[q]) : "v"(da));
            else if (WHICH == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(a), "v"(b));
            else if (WHICH == 5) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(c[q]), "+v"(x[q]) : "v"(a), "v"(b) : "vcc");
            else if (WHICH == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 20" : "+v"(x[q]) : "v"(a));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = c[0] + c[1] + c[2] + c[3] + x[0] + x[1] + x[2] + x[3] + (uint64_t)(d[0] + d[1] + d[2] + d[3]);
}
__global__ void kmul(Fr* out, int n) {
    Fr x = out[blockIdx.x * 64 + threadIdx.x], y = x;
    for (int i = 0; i < n; i++) x = fr_mul(x, y);
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
static int clk_khz;
template <int W, int CH> void run(const char* name, int waves_per_simd) {
    uint64_t* d; hipMalloc(&d, 1 << 24);
    const int iters = 64, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<W, CH>), dim3(blocks), dim3(64), 0, 0, d, 7u, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<W, CH>), dim3(blocks), dim3(64), 0, 0, d, 7u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)REP * iters;
    printf("%-28s chains %d waves/SIMD %d: %8.3f ms  %6.1f cycles per instruction per wave, %5.1f per SIMD\n", name, CH, waves_per_simd, ms, ms * 1e-3 / n * clk_khz * 1e3,
           ms * 1e-3 / n * clk_khz * 1e3 / waves_per_simd);
    hipFree(d);
}
int main() {
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("clock %d MHz\n", clk_khz / 1000);
    for (int w : {1, 2}) {
        if (w == 1) {
            run<0, 4>("v_mad_u64_u32", 1); run<0, 1>("v_mad_u64_u32", 1); run<5, 1>("v_mad_u64_u32 + v_addc (pair)", 1); run<1, 4>("v_lshl_add_u64", 1); run<1, 1>("v_lshl_add_u64", 1);
            run<2, 4>("v_fma_f64", 1); run<2, 1>("v_fma_f64", 1); run<3, 4>("v_add_f64", 1); run<3, 1>("v_add_f64", 1); run<4, 4>("v_and_or_b32", 1); run<6, 4>("v_alignbit_b32", 1);
        } else {
            run<0, 4>("v_mad_u64_u32", 2); run<5, 1>("v_mad_u64_u32 + v_addc (pair)", 2); run<1, 4>("v_lshl_add_u64", 2); run<2, 4>("v_fma_f64", 2); run<3, 4>("v_add_f64", 2);
        }
    }
    Fr* f; hipMalloc(&f, 256 * 4 * 4 * 64 * sizeof(Fr)); hipMemset(f, 1, 256 * 4 * 4 * 64 * sizeof(Fr));
    for (int w : {1, 2, 4}) {
        const int n = 2000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kmul, dim3(256 * 4 * w), dim3(64), 0, 0, f, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmul, dim3(256 * 4 * w), dim3(64), 0, 0, f, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("fr_mul chain, %d waves/SIMD: %.3f us per product per wave (%.0f cycles), %.0f cycles per SIMD\n", w, ms * 1e3 / n, ms * 1e-3 / n * clk_khz * 1e3, ms * 1e-3 / n * clk_khz * 1e3 / w);
    }
    return 0;
}
