// Micro-benchmark: what does a LONE latency-bound wavefront pay for running beside an HBM-streaming kernel?
// (the generation's BN254 / RLP chains beside the Keccak round expansion).
//   hipcc --offload-arch=gfx950 -O3 -I proof_of_burn_amd/csrc tools/ubench/contend.hip -o /tmp/contend && /tmp/contend
// A "streamer" (30 720 wavefronts x 1 604 coalesced 512-byte stores, or loads with the READ variant) runs on one stream; a probe of
// 16..80 wavefronts on another.  Probes: (alu) dependent v_mad_u64_u32 chain, no memory; (alu+st) the same with 8 x 256-byte stores
// every 364 instructions (a BN254 wire per Montgomery product); (chase) dependent loads over 1 GB; (chase-hot) over 32 KB;
// (ld+st) one load then a dependent store per step (a store->load round trip).  Each probe is timed alone and beside the streamer,
// on the whole chip and with the probe's stream confined to CUs the streamer's stream is masked away from.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <bool READ> __global__ void __launch_bounds__(64) streamer(uint64_t* buf, uint64_t words_per_wave, uint64_t* sink) {
    uint64_t* p = buf + (uint64_t)blockIdx.x * words_per_wave + threadIdx.x;
    uint64_t acc = 0;
    for (uint64_t i = 0; i < words_per_wave; i += 64 * 4) {        // 1 604 rows of 64 words = 401 x 4
#pragma unroll
        for (int k = 0; k < 4; k++) { if (READ) acc ^= __builtin_nontemporal_load(p + i + 64 * k); else p[i + 64 * k] = i + k; }
    }
    if (READ && acc == 0x1234567) sink[0] = acc;
}

// probe kinds
enum { P_ALU = 0, P_ALU_ST, P_CHASE, P_CHASE_HOT, P_LD_ST, P_ALU_PRIO, P_ALU_BIG, P_NKINDS };
static const char* PNAME[] = {"alu", "alu+st", "chase-1GB", "chase-32KB", "ld->st->ld", "alu prio3", "alu 96KB code"};

template <int KIND> __global__ void __launch_bounds__(64) probe(uint32_t* out, const uint32_t* chase, uint32_t steps, uint32_t* scratch) {
    if (KIND == P_ALU_PRIO) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x, w = blockIdx.x;
    if (KIND == P_ALU || KIND == P_ALU_ST || KIND == P_ALU_PRIO) {
        uint64_t c = lane; uint32_t a = lane * 2654435761u + 1, b = w + 12345;
        uint32_t* dst = scratch + (uint64_t)w * steps * 8 * 64 + lane;
        for (uint32_t s = 0; s < steps; s++) {
#pragma unroll
            for (int r = 0; r < 364; r++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
            if (KIND == P_ALU_ST) {
#pragma unroll
                for (int k = 0; k < 8; k++) dst[(uint64_t)(s * 8 + k) * 64] = (uint32_t)c + k;
            }
        }
        out[w * 64 + lane] = (uint32_t)c;
    } else if (KIND == P_ALU_BIG) {      // the same instruction count as `alu`, but as 96 KB of straight-line code walked 32 times (64 KB I-cache)
        uint64_t c = lane; uint32_t a = lane * 2654435761u + 1, b = w + 12345;
        for (uint32_t s = 0; s < steps / 32; s++) {
            asm volatile(".rept 12000\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n .endr" : "+v"(c) : "v"(a), "v"(b) : "vcc");
        }
        out[w * 64 + lane] = (uint32_t)c;
    } else if (KIND == P_CHASE || KIND == P_CHASE_HOT) {
        uint32_t idx = KIND == P_CHASE ? w * 1024 * 16 : (w * 16) % 8192;      // wave-uniform chain: one line per step
        for (uint32_t s = 0; s < steps; s++) idx = chase[idx];
        out[w * 64 + lane] = idx;
    } else {
        volatile uint32_t* q = scratch + (uint64_t)w * 64 * 1024 + lane;
        uint32_t v = lane;
        for (uint32_t s = 0; s < steps; s++) { q[(s & 1023) * 64] = v + s; __builtin_amdgcn_s_waitcnt(0); v = q[(s & 1023) * 64] + 1; }
        out[w * 64 + lane] = v;
    }
}

static void launch_probe(int kind, uint32_t waves, hipStream_t st, uint32_t* out, const uint32_t* big, const uint32_t* hot, uint32_t* scratch) {
    switch (kind) {
    case P_ALU: hipLaunchKernelGGL(probe<P_ALU>, dim3(waves), dim3(64), 0, st, out, big, 1000u, scratch); break;
    case P_ALU_ST: hipLaunchKernelGGL(probe<P_ALU_ST>, dim3(waves), dim3(64), 0, st, out, big, 1000u, scratch); break;
    case P_CHASE: hipLaunchKernelGGL(probe<P_CHASE>, dim3(waves), dim3(64), 0, st, out, big, 600u, scratch); break;
    case P_CHASE_HOT: hipLaunchKernelGGL(probe<P_CHASE_HOT>, dim3(waves), dim3(64), 0, st, out, hot, 3000u, scratch); break;
    case P_LD_ST: hipLaunchKernelGGL(probe<P_LD_ST>, dim3(waves), dim3(64), 0, st, out, big, 600u, scratch); break;
    case P_ALU_BIG: hipLaunchKernelGGL(probe<P_ALU_BIG>, dim3(waves), dim3(64), 0, st, out, big, 1024u, scratch); break;
    case P_ALU_PRIO: hipLaunchKernelGGL(probe<P_ALU_PRIO>, dim3(waves), dim3(64), 0, st, out, big, 1000u, scratch); break;
    }
}

int main(int argc, char** argv) {
    const uint32_t waves = argc > 1 ? atoi(argv[1]) : 16;
    const uint32_t reserve = argc > 2 ? atoi(argv[2]) : 2;     // CUs per XCD for the masked variant
    const uint64_t WPW = 102656 + 1600 * 0;                       // 8-byte words per streamer wavefront (one KeccakfRound block)
    const uint32_t NW = 30720;
    uint64_t* buf; CK(hipMalloc(&buf, NW * WPW * 8));
    CK(hipMemset(buf, 1, NW * WPW * 8));
    uint64_t* sink; CK(hipMalloc(&sink, 64));
    const uint32_t NBIG = 1u << 28;                               // 1 GB of chase indices
    std::vector<uint32_t> hb(NBIG);
    {   // random single cycle-ish permutation at 64-byte granularity (16 words), enough to defeat caches
        uint64_t x = 88172645463325252ull;
        for (uint32_t i = 0; i < NBIG; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hb[i] = (uint32_t)(x % (NBIG / 16)) * 16; }
    }
    uint32_t *big, *hot, *out, *scratch;
    CK(hipMalloc(&big, (uint64_t)NBIG * 4)); CK(hipMemcpy(big, hb.data(), (uint64_t)NBIG * 4, hipMemcpyHostToDevice));
    const uint32_t NHOT = 8192;
    std::vector<uint32_t> hh(NHOT); for (uint32_t i = 0; i < NHOT; i++) hh[i] = (uint32_t)((i * 2654435761u) % (NHOT / 16)) * 16;
    CK(hipMalloc(&hot, NHOT * 4)); CK(hipMemcpy(hot, hh.data(), NHOT * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMalloc(&scratch, (uint64_t)128 * 1000 * 8 * 64 * 4 + (1 << 20)));

    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const uint32_t ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> mb(words, 0), mc(words, 0);
    for (uint32_t i = 0; i < ncu; i++) (i >= ncu - 8 * reserve ? mc : mb)[i >> 5] |= 1u << (i & 31);
    int plo, phi; CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    hipStream_t s_bulk, s_probe, s_bulk_m, s_probe_m;
    CK(hipStreamCreateWithPriority(&s_bulk, hipStreamNonBlocking, plo)); CK(hipStreamCreateWithPriority(&s_probe, hipStreamNonBlocking, phi));
    CK(hipExtStreamCreateWithCUMask(&s_bulk_m, words, mb.data())); CK(hipExtStreamCreateWithCUMask(&s_probe_m, words, mc.data()));
    hipEvent_t e0, e1, b0, b1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));

    auto time_bulk = [&](bool read, hipStream_t sb) {
        CK(hipEventRecord(b0, sb));
        if (read) hipLaunchKernelGGL(streamer<true>, dim3(NW), dim3(64), 0, sb, buf, WPW, sink); else hipLaunchKernelGGL(streamer<false>, dim3(NW), dim3(64), 0, sb, buf, WPW, sink);
        CK(hipEventRecord(b1, sb)); CK(hipEventSynchronize(b1));
        float ms; CK(hipEventElapsedTime(&ms, b0, b1)); return ms;
    };
    printf("CUs %u, probe waves %u, reserved CUs/XCD (masked variant) %u\n", ncu, waves, reserve);
    for (int read = 0; read < 2; read++) {
        time_bulk(read, s_bulk);
        printf("streamer %s alone: %.3f ms (%.0f GB/s); on %u CUs: %.3f ms\n", read ? "READ" : "WRITE", time_bulk(read, s_bulk), NW * WPW * 8 / (time_bulk(read, s_bulk) * 1e6),
               ncu - 8 * reserve, time_bulk(read, s_bulk_m));
    }
    printf("%-12s %10s %14s %14s %14s %14s\n", "probe", "alone ms", "beside WRITE", "beside READ", "masked WRITE", "masked READ");
    for (int kind = 0; kind < P_NKINDS; kind++) {
        float r[5] = {0, 0, 0, 0, 0};
        for (int mode = 0; mode < 5; mode++) {       // 0 alone, 1 write, 2 read, 3 masked write, 4 masked read
            const bool masked = mode >= 3, read = mode == 2 || mode == 4;
            hipStream_t sb = masked ? s_bulk_m : s_bulk, sp = masked ? s_probe_m : s_probe;
            float best = 1e9;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipDeviceSynchronize());
                if (mode) {      // three streamers back to back so the probe (<= ~10 ms) never outlives them
                    for (int k = 0; k < 4; k++) { if (read) hipLaunchKernelGGL(streamer<true>, dim3(NW), dim3(64), 0, sb, buf, WPW, sink); else hipLaunchKernelGGL(streamer<false>, dim3(NW), dim3(64), 0, sb, buf, WPW, sink); }
                    CK(hipEventRecord(b0, sb));
                }
                CK(hipEventRecord(e0, sp));
                launch_probe(kind, waves, sp, out, big, hot, scratch);
                CK(hipEventRecord(e1, sp));
                CK(hipEventSynchronize(e1));
                const bool still = mode && hipEventQuery(b0) == hipErrorNotReady;      // streamers still running when the probe ended
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (mode && !still) ms = -ms;        // (flag: the probe outlived the streamers)
                best = std::min(best, ms);
            }
            r[mode] = best;
        }
        printf("%-12s %10.3f %14.3f %14.3f %14.3f %14.3f\n", PNAME[kind], r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
