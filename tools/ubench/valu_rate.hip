// Micro-benchmark: issue cost (cycles per wave64 instruction) of the integer instructions the BN254 code is made of.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 4096
template <int WHICH> __global__ void k(uint64_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3;
    uint32_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (WHICH == 0) {        // v_mad_u64_u32, 4 independent chains
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b) : "vcc");
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b) : "vcc");
            } else if (WHICH == 1) { // v_lshl_add_u64
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c0) : "v"(c1));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c1) : "v"(c2));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c2) : "v"(c3));
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c3) : "v"(c0));
            } else if (WHICH == 2) { // v_add_u32
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(x2));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x2) : "v"(x3));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x3) : "v"(x0));
            } else if (WHICH == 3) { // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x1) : "v"(x2));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x2) : "v"(x3));
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x3) : "v"(x0));
            } else if (WHICH == 4) { // v_add_co_u32 / v_addc_co_u32 pair
                asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(x0), "+v"(x1) : "v"(x2), "v"(x3) : "vcc");
                asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(x2), "+v"(x3) : "v"(x0), "v"(x1) : "vcc");
            } else if (WHICH == 5) { // v_mul_hi_u32
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x1) : "v"(x2));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x2) : "v"(x3));
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x3) : "v"(x0));
            } else if (WHICH == 6) { // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x1) : "v"(x2), "v"(x3));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x2) : "v"(x3), "v"(x0));
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x3) : "v"(x0), "v"(x1));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + x0 + x1 + x2 + x3;
}

template <int W> void run(const char* name, int waves_per_simd) {
    uint64_t* d; hipMalloc(&d, 1 << 24);
    const int iters = 64, blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64), 0, 0, d, 7u, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64), 0, 0, d, 7u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);      // kHz
    const double n = (double)REP * iters * (W == 4 ? 1.0 : 1.0);
    printf("%-16s waves/SIMD %d: %.3f ms, %.2f ns per instr per wave-slot, ~%.1f cycles @%d MHz (per SIMD: %.1f cycles/instr)\n", name, waves_per_simd, ms,
           ms * 1e6 / n, ms * 1e-3 / n * clk * 1e3, clk / 1000, ms * 1e-3 / n * clk * 1e3 / waves_per_simd);
    hipFree(d);
}
int main() {
    for (int w : {1, 2}) {
        if (w == 1) { run<0>("v_mad_u64_u32", 1); run<1>("v_lshl_add_u64", 1); run<2>("v_add_u32", 1); run<3>("v_mul_lo_u32", 1); run<4>("add_co+addc_co", 1); run<5>("v_mul_hi_u32", 1); run<6>("v_mad_u32_u24", 1); }
        else { run<0>("v_mad_u64_u32", 2); run<1>("v_lshl_add_u64", 2); run<2>("v_add_u32", 2); run<3>("v_mul_lo_u32", 2); }
    }
    return 0;
}
