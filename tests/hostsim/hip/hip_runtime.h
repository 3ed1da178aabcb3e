/* TEST INFRASTRUCTURE -- "HIP on fibers": a stand-in for <hip/hip_runtime.h> that lets the UNCHANGED sources of libpob_hip.so
 * (kernels, device policies, host scheduler, C ABI) compile with a host compiler and run on the CPU, one fiber per GPU thread.
 * It exists so that the logic of the product's own kernels -- layout arithmetic, the constraint evaluator's relations, the emitter --
 * can be exercised by `pytest -m "not gpu"` in a container without a GPU.  It is not a CPU fallback: the product
 * (proof_of_burn_amd/) never loads it, nothing here is optimised, and only tests/ build and load tests/hostsim/libpob_hostsim.so.
 *
 * Model: a kernel launch runs its blocks one after the other; the threads of a block are fibers switched round-robin at the
 * wave-level operations (ballot / readlane / readfirstlane / bpermute / shfl / any), which exchange values through a small mailbox.
 * Streams and events are no-ops (everything is synchronous), device memory is host memory.  Raw buffer loads/stores emulate the
 * bounds check of a buffer resource (out-of-range loads return 0, stores are dropped) because the kernels rely on it.
 */
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <functional>

#define __host__
#define __device__
#define __global__
#define __constant__
#define __shared__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3 { uint32_t x, y, z; dim3(uint32_t a = 1, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {} };
struct hs_idx { uint32_t x, y, z; };
extern hs_idx blockIdx, blockDim, gridDim;
struct hs_tid { uint32_t y = 0, z = 0; struct X { operator uint32_t() const; } x; };
extern hs_tid threadIdx;

struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint4 r = {a, b, c, d}; return r; }

// ---- wave-level operations (64-lane wavefront = the block's fibers)
uint64_t hs_ballot(bool pred);
uint32_t hs_readlane(uint32_t v, uint32_t lane);
uint32_t hs_readfirstlane(uint32_t v);        // checks that the value is in fact wave-uniform (the kernels claim it is "by construction")
uint32_t hs_shfl(uint32_t v, uint32_t src);
bool hs_any(bool pred);
#define __ballot(p) hs_ballot(p)
#define __any(p) hs_any(p)
#define __shfl(v, src, w) hs_shfl((uint32_t)(v), (uint32_t)(src))
#define __shfl_xor(v, m, w) hs_shfl((uint32_t)(v), ((uint32_t)threadIdx.x & 63u) ^ (uint32_t)(m))
void hs_sync();                                // the whole block meets (ballots / shuffles / readlane: the calling fiber's wavefront, 64 consecutive threads)
#define __syncthreads() hs_sync()
#define __builtin_amdgcn_readlane(v, l) ((int)hs_readlane((uint32_t)(v), (uint32_t)(l)))
#define __builtin_amdgcn_readfirstlane(v) ((int)hs_readfirstlane((uint32_t)(v)))
#define __builtin_amdgcn_ds_bpermute(addr, v) ((int)hs_shfl((uint32_t)(v), ((uint32_t)(addr) >> 2) & 63u))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

// ---- buffer resources
struct hs_rsrc { char* base; uint32_t size; };
#define __amdgpu_buffer_rsrc_t hs_rsrc
static inline hs_rsrc hs_make_rsrc(const void* p, int num) { hs_rsrc r = {(char*)p, (uint32_t)num}; return r; }
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) hs_make_rsrc((const void*)(p), (int)(num))
typedef int hs_v2i __attribute__((ext_vector_type(2)));
template <class T> static inline T hs_buf_ld(hs_rsrc r, int voff, int soff) {
    const uint64_t o = (uint64_t)(uint32_t)voff + (uint32_t)soff;
    T v; memset(&v, 0, sizeof v);
    if (o + sizeof(T) <= r.size) memcpy(&v, r.base + o, sizeof(T));
    return v;
}
template <class T> static inline void hs_buf_st(T v, hs_rsrc r, int voff, int soff) {
    const uint64_t o = (uint64_t)(uint32_t)voff + (uint32_t)soff;
    if (o + sizeof(T) <= r.size) memcpy(r.base + o, &v, sizeof(T));
}
#define __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, aux) hs_buf_ld<int>(r, vo, so)
#define __builtin_amdgcn_raw_buffer_load_b8(r, vo, so, aux) hs_buf_ld<char>(r, vo, so)
#define __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, aux) hs_buf_ld<hs_v2i>(r, vo, so)
#define __builtin_amdgcn_raw_buffer_store_b32(v, r, vo, so, aux) hs_buf_st<int>(v, r, vo, so)
#define __builtin_amdgcn_raw_buffer_store_b8(v, r, vo, so, aux) hs_buf_st<char>(v, r, vo, so)
#define __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, aux) hs_buf_st<hs_v2i>(v, r, vo, so)

#define __popcll(x) __builtin_popcountll(x)
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }

// ---- runtime API (synchronous)
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventBlockingSync = 1, hipHostMallocDefault = 0 };
static inline const char* hipGetErrorString(hipError_t) { return "hostsim error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned) { *p = (T*)malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

void hs_launch(dim3 grid, dim3 block, const std::function<void()>& body);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hs_launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
