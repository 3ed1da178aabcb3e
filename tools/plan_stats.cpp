// Host tool: stored wires per unit kind and storage class of the production instantiation (what the G side writes per 64 witnesses).
//   built like tests/hostsim (the shim stands in for <hip/hip_runtime.h>): see tools/plan_stats.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>
#include "circuits.hpp"
int main() {
    static Plan plan;
    PobParams prm; prm.L = 16; prm.NB = 4; prm.HB = 16; prm.minNib = 50; prm.amountBytes = 31; prm.powZero = 2; prm.maxIntended = fr_zero(); prm.maxActual = fr_zero();
    plan.plan_pob(prm);
    struct Acc { uint64_t units = 0, nb = 0, ns = 0, nf = 0; };
    std::map<uint32_t, Acc> acc;
    for (const UnitDesc& d : plan.units) {
        if (!(d.flags & UNIT_GEN)) continue;
        Acc& a = acc[d.kind]; a.units++;
        if (d.kind == U_POS_WIDE) { a.nf += pos_wires((int)d.a[0], pos_off((int)d.a[0]).rp); continue; }
        CountP q; unit_run_all(q, d, plan.L); a.nb += q.nb; a.ns += q.ns; a.nf += q.nf;
    }
    uint64_t tb = 0, ts = 0, tf = 0;
    printf("%-6s %8s %10s %10s %8s %12s\n", "kind", "units", "BIT", "SM", "FR", "bytes/group");
    for (auto& kv : acc) { const Acc& a = kv.second; printf("%-6u %8llu %10llu %10llu %8llu %12llu\n", kv.first, (unsigned long long)a.units, (unsigned long long)a.nb, (unsigned long long)a.ns, (unsigned long long)a.nf, (unsigned long long)(a.nb * 8 + a.ns * 256 + a.nf * 2048)); tb += a.nb; ts += a.ns; tf += a.nf; }
    printf("G units write: BIT %llu SM %llu FR %llu; plan totals: w %u b %u s %u f %u q %u\n", (unsigned long long)tb, (unsigned long long)ts, (unsigned long long)tf, plan.total.w, plan.total.b, plan.total.s, plan.total.f, plan.total.q);
    uint64_t perms = 0; for (const SpongeDesc& s : plan.sponges) perms += s.n;
    printf("sponges %zu perms %llu; Keccak stored BIT per group: chain wires %llu, round arrays %llu\n", plan.sponges.size(), (unsigned long long)perms, (unsigned long long)(perms * AB_DIRECT), (unsigned long long)(perms * 24 * KR_BITS));
    return 0;
}
