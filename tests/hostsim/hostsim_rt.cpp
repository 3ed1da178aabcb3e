// TEST INFRASTRUCTURE -- fiber runtime of the "HIP on fibers" shim (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>
#include <execinfo.h>
#include <dlfcn.h>

// AddressSanitizer build (tests/hostsim/build.py san=True): the fiber switches are announced to the runtime, which otherwise takes a fiber's
// stack for a wild pointer into the launcher's
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HS_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif
#ifndef HS_ASAN
#define HS_ASAN 0
#endif

hs_idx blockIdx, blockDim, gridDim;
hs_tid threadIdx;
uint32_t g_lds[1 << 16];                  // `extern __shared__ uint32_t g_lds[]` of g_units.hpp

namespace {
const int MAXT = 256;
const size_t STACK = 4u << 20;
struct Fiber { void* sp; char* stack; bool alive; void* fake; };
const void* main_bottom; size_t main_size; void* main_fake;      // (ASan: the launcher's stack, learnt at the first switch away from it)
Fiber fib[MAXT];
void* main_sp;
int nthreads, cur, alive;
const std::function<void()>* body;
// Meetings have a SCOPE: the wavefront of the calling fiber (ballots, shuffles, readlane: scopes 0 .. MAXT / 64 - 1) or the whole block (__syncthreads: scope NSCOPE - 1)
const int NSCOPE = MAXT / 64 + 1, BLOCK = NSCOPE - 1;
struct Scope {
    uint64_t mbox[2][MAXT]; bool present[2][MAXT];   // present: deposited in this meeting (a fiber may END before the others read)
    uint32_t cnt[2]; uint64_t tag[2]; int alive;
    uint64_t gen_of[MAXT];
} sc[NSCOPE];

// x86-64 System V context switch: callee-saved registers + stack pointer
extern "C" void hs_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl hs_switch
.type hs_switch,@function
hs_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

void yield_next() {            // round-robin to the next living fiber, or back to the launcher when none is left
    const int me = cur;
    for (int k = 1; k <= nthreads; k++) {
        const int t = (me + k) % nthreads;
        if (fib[t].alive) {
            if (t == me) return;
            cur = t;
#if HS_ASAN
            __sanitizer_start_switch_fiber(fib[me].alive ? &fib[me].fake : nullptr, fib[t].stack, STACK);
#endif
            hs_switch(&fib[me].sp, fib[t].sp);
#if HS_ASAN
            __sanitizer_finish_switch_fiber(fib[me].fake, nullptr, nullptr);
#endif
            return;
        }
    }
#if HS_ASAN
    __sanitizer_start_switch_fiber(nullptr, main_bottom, main_size);      // the last fiber of the block ends: back to the launcher
#endif
    hs_switch(&fib[me].sp, main_sp);
}
extern "C" void hs_entry() {
#if HS_ASAN
    { const void* b; size_t n; __sanitizer_finish_switch_fiber(nullptr, &b, &n); if (!main_bottom) { main_bottom = b; main_size = n; } }
#endif
    (*body)();
    fib[cur].alive = false; alive--; sc[BLOCK].alive--; sc[cur / 64].alive--;
    yield_next();
    abort();
}
void make_fiber(int t) {
    Fiber& f = fib[t];
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == MAP_FAILED) { perror("hostsim: mmap"); abort(); }
    }
    void** sp = (void**)(f.stack + STACK - 64);
    *--sp = nullptr;                       // alignment slot: hs_entry starts with rsp % 16 == 8, like after a call
    *--sp = (void*)hs_entry;               // return address of hs_switch
    for (int i = 0; i < 6; i++) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    f.sp = sp; f.alive = true;
    for (Scope& c : sc) c.gen_of[t] = 0;
}
// all living fibers of the scope deposit v and meet; returns the mailbox of this meeting
struct Met { const uint64_t* m; const bool* pr; };
Met exchange(uint64_t v, int scope) {
    Scope& c = sc[scope];
    const int me = cur;
    const uint64_t g = c.gen_of[me]++;
    const int b = (int)(g & 1);
    if (c.tag[b] != g) { c.tag[b] = g; c.cnt[b] = 0; memset(c.present[b], 0, sizeof c.present[b]); }
    c.mbox[b][me] = v; c.present[b][me] = true; c.cnt[b]++;
    while ((int)c.cnt[b] < c.alive) yield_next();
    return {c.mbox[b], c.present[b]};
}
inline int wave_of() { return cur / 64; }
inline int wave_n() { const int left = nthreads - 64 * wave_of(); return left < 64 ? left : 64; }
}  // namespace

hs_tid::X::operator uint32_t() const { return (uint32_t)cur; }

uint64_t hs_ballot(bool pred) {
    const int w0 = 64 * wave_of(), n = wave_n();
    const Met x = exchange(pred ? 1 : 0, wave_of());
    uint64_t r = 0;
    for (int t = 0; t < n; t++) if (x.pr[w0 + t] && x.m[w0 + t]) r |= 1ull << t;
    return r;
}
bool hs_any(bool pred) { return hs_ballot(pred) != 0; }
void hs_sync() { exchange(0, BLOCK); }
uint32_t hs_readlane(uint32_t v, uint32_t lane) { const int w0 = 64 * wave_of(), n = wave_n(); return (uint32_t)exchange(v, wave_of()).m[w0 + lane % (uint32_t)n]; }
uint32_t hs_shfl(uint32_t v, uint32_t src) { const int w0 = 64 * wave_of(), n = wave_n(); return (uint32_t)exchange(v, wave_of()).m[w0 + src % (uint32_t)n]; }
// The kernels use readfirstlane only to tell the compiler that a wire index is wave-uniform ("by construction"), also inside
// divergent code such as `cond ? p.get(ref) : 0`, where on the GPU only the active lanes execute it.  Independent fibers cannot
// meet there, so by default this is the identity; HOSTSIM_STRICT_UNIFORM=1 makes it a meeting that verifies uniformity (and
// aborts at the divergent call sites).
static const bool strict_uniform = getenv("HOSTSIM_STRICT_UNIFORM") != nullptr;
uint32_t hs_readfirstlane(uint32_t v) {
    if (!strict_uniform) return v;
    const Met x = exchange(v, wave_of());
    const uint64_t* m = x.m; const bool* pr = x.pr;
    int first = -1;
    for (int t = 64 * wave_of(); t < 64 * wave_of() + wave_n(); t++) if (pr[t]) { if (first < 0) first = t; else if ((uint32_t)m[t] != (uint32_t)m[first]) {
        fprintf(stderr, "hostsim: readfirstlane of a NON-UNIFORM value (lane %d: %u, lane %d: %u) in block (%u,%u)\n", first, (uint32_t)m[first], t, (uint32_t)m[t], blockIdx.x, blockIdx.y);
        void* bt[16]; const int n = backtrace(bt, 16);
        for (int i = 0; i < n; i++) { Dl_info di; if (dladdr(bt[i], &di) && di.dli_fbase) fprintf(stderr, "  frame %d: %s +0x%lx\n", i, di.dli_fname, (unsigned long)((char*)bt[i] - (char*)di.dli_fbase)); }
        abort(); } }
    return (uint32_t)m[first];
}

void hs_launch(dim3 grid, dim3 block, const std::function<void()>& fn) {
    if (block.x > (uint32_t)MAXT || block.y != 1 || block.z != 1) { fprintf(stderr, "hostsim: unsupported block shape\n"); abort(); }
    gridDim = {grid.x, grid.y, grid.z}; blockDim = {block.x, 1, 1};
    body = &fn;
    for (uint32_t bz = 0; bz < grid.z; bz++) for (uint32_t by = 0; by < grid.y; by++) for (uint32_t bx = 0; bx < grid.x; bx++) {
        blockIdx = {bx, by, bz};
        nthreads = (int)block.x; alive = nthreads;
        for (int k = 0; k < NSCOPE; k++) { sc[k].tag[0] = sc[k].tag[1] = ~0ull; sc[k].alive = k == BLOCK ? nthreads : (nthreads - 64 * k < 0 ? 0 : nthreads - 64 * k > 64 ? 64 : nthreads - 64 * k); }
        for (int t = 0; t < nthreads; t++) make_fiber(t);
        cur = 0;
#if HS_ASAN
        __sanitizer_start_switch_fiber(&main_fake, fib[0].stack, STACK);
#endif
        hs_switch(&main_sp, fib[0].sp);
#if HS_ASAN
        __sanitizer_finish_switch_fiber(main_fake, nullptr, nullptr);
#endif
    }
}
