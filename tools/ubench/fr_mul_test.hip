#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../proof_of_burn_amd/csrc/fr_dev.hpp"
__global__ void k(const Fr* a, const Fr* b, Fr* out, int n) { int i = blockIdx.x * 64 + threadIdx.x; if (i < n) out[i] = fr_mul(a[i], b[i]); }
__global__ void kinv(const Fr* a, Fr* out, int n) { int i = blockIdx.x * 64 + threadIdx.x; if (i < n) out[i] = fr_inv(a[i]); }
// host reference: same CIOS
int main() {
    const int n = 4096;
    Fr *ha = new Fr[n], *hb = new Fr[n], *ho = new Fr[n];
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    const uint32_t P[8] = FR_P_LIMBS;
    for (int i = 0; i < n; i++) for (int j = 0; j < 8; j++) { ha[i].l[j] = rnd(); hb[i].l[j] = rnd(); }
    for (int i = 0; i < n; i++) { ha[i].l[7] &= 0x0fffffff; hb[i].l[7] &= 0x0fffffff; }    // < p
    for (int j = 0; j < 8; j++) { ha[0].l[j] = P[j]; hb[0].l[j] = P[j]; ha[1].l[j] = 0; hb[2].l[j] = 0xffffffffu; ha[2].l[j] = 0xffffffffu; }
    ha[0].l[0] -= 1; hb[0].l[0] -= 1; ha[2].l[7] = hb[2].l[7] = 0x30644e72u; ha[2].l[6] = hb[2].l[6] = 0xe131a028u;
    Fr *da, *db, *dout; hipMalloc(&da, n * sizeof(Fr)); hipMalloc(&db, n * sizeof(Fr)); hipMalloc(&dout, n * sizeof(Fr));
    hipMemcpy(da, ha, n * sizeof(Fr), hipMemcpyHostToDevice); hipMemcpy(db, hb, n * sizeof(Fr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, da, db, dout, n);
    hipMemcpy(ho, dout, n * sizeof(Fr), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) { Fr r = fr_mul(ha[i], hb[i]); for (int j = 0; j < 8; j++) if (r.l[j] != ho[i].l[j]) { bad++; break; } }
    printf("fr_mul device vs host: %d mismatches of %d\n", bad, n);
    // fr_inv: device (Kaliski, batched shifts) vs host (binary Euclid) on Montgomery-form operands, incl. 0, 1, p-1, powers of two
    for (int j = 0; j < 8; j++) { ha[3].l[j] = 0; ha[4].l[j] = j == 0; ha[5].l[j] = 0; }
    ha[5].l[3] = 0x100;
    for (int i = 6; i < 40; i++) { for (int j = 0; j < 8; j++) ha[i].l[j] = 0; ha[i].l[(i * 7) >> 5 & 7] = 1u << ((i * 7) & 31); ha[i].l[7] &= 0x0fffffff; }
    hipMemcpy(da, ha, n * sizeof(Fr), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kinv, dim3(n / 64), dim3(64), 0, 0, da, dout, n);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kinv, dim3(n / 64), dim3(64), 0, 0, da, dout, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(ho, dout, n * sizeof(Fr), hipMemcpyDeviceToHost);
    int badi = 0;
    for (int i = 0; i < n; i++) { Fr r = fr_inv(ha[i]); for (int j = 0; j < 8; j++) if (r.l[j] != ho[i].l[j]) { if (badi < 3) printf("  inv mismatch at %d\n", i); badi++; break; } }
    printf("fr_inv device vs host: %d mismatches of %d (one wave per SIMD: %.1f us per inversion)\n", badi, n, ms * 1e3);
    return bad != 0 || badi != 0;
}
