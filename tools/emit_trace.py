#!/usr/bin/env python3
"""Emission of three production witnesses after a warm-up one (run under rocprofv3 --kernel-trace --memory-copy-trace), or, with a
rocpd database as argument, the timeline of the LAST witness: every D2H copy and the kernels of every window, per window.
    rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o e -- python tools/emit_trace.py run [reduced]
    python tools/emit_trace.py DB"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(reduced: bool):
    from proof_of_burn_amd import WitnessCalculator, inputs as G
    main = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
    b = G.synthetic_batch(64, depth=10, seed=5, distinct_keys=2)
    calc = WitnessCalculator(main, max_batch=64)
    calc.calculate(b.inputs)
    keep = None
    if reduced:
        from proof_of_burn_amd.circuit_model import keepmap
        keep, _ = keepmap.load(main)
    calc.emit_throughput(0, 1, keep=keep, window_wires=(16 << 20) if reduced else 0)
    sec, nb = calc.emit_throughput(1, 3, keep=keep, window_wires=(16 << 20) if reduced else 0)
    print(f"{nb / sec / 1e9:.2f} GB/s, {sec / 3 * 1e3:.2f} ms per witness")
    calc.close()


def show(path):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch')); ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    mc = next(t for t in tabs if t.startswith('rocpd_memory_copy'))
    cps = cur.execute(f"select start,end,size from {mc} where size >= 1000000 order by start").fetchall()
    n = len(cps) // 4                                   # four witnesses: the last one
    cps = cps[3 * n:]
    t0 = cps[0][0] - 6_000_000
    ker = cur.execute(f"select k.kernel_name,d.start,d.end from {kd} d join {ks} k on d.kernel_id=k.id where d.start>={t0} order by d.start").fetchall()
    print("# window: copy start, end, ms, GB/s | gap to the previous copy's end | kernels between the previous copy's start and this one's: busy ms (count), by kind")
    prev_end, prev_start = None, t0
    for s, e, sz in cps:
        ks_ = [(nm, a, b) for nm, a, b in ker if prev_start <= a < s]
        kinds = {}
        for nm, a, b in ks_:
            key = 'fill' if 'fill' in nm else 'bits' if 'emit_bits' in nm else 'gEMIT' if 'EmitP' in nm else nm[:12]
            kinds[key] = kinds.get(key, 0) + (b - a) / 1e6
        gap = (s - prev_end) / 1e6 if prev_end else 0.0
        print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:6.3f} {sz / (e - s):6.1f} | gap {gap:6.3f} | " + " ".join(f"{k} {v:.3f}" for k, v in sorted(kinds.items())))
        prev_end, prev_start = e, s
    print(f"# witness: {(cps[-1][1] - cps[0][0]) / 1e6:.2f} ms from the first copy's start to the last copy's end; copies busy {sum(e - s for s, e, _ in cps) / 1e6:.2f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(len(sys.argv) > 2 and sys.argv[2] == "reduced")
    else:
        show(sys.argv[1])
