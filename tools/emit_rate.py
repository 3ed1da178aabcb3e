#!/usr/bin/env python3
"""Emission rates of ONE production calculator: the O0 payload and the reduced (O1-style) one, steady state (POB_LIB_PATH selects the build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proof_of_burn_amd import WitnessCalculator, inputs as gen  # noqa: E402
from proof_of_burn_amd.circuit_model import keepmap  # noqa: E402

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
batch = gen.synthetic_batch(64, depth=10, seed=0xB0B, distinct_keys=4)
calc = WitnessCalculator(MAIN, max_batch=64)
res = calc.calculate(batch.inputs, check=True)
assert all(r.ok for r in res)
calc.emit_throughput(0, count=1)
sec, nbytes = calc.emit_throughput(1, count=3)
keep, _ = keepmap.load(MAIN)
calc.emit_throughput(0, count=1, keep=keep, window_wires=1 << 24)
out = []
for rep in range(3):
    rsec, rbytes = calc.emit_throughput(1, count=6, keep=keep, window_wires=1 << 24)
    out.append(round(rsec / 6 * 1e3, 2))
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'lib'}: O0 payload {nbytes / sec / 1e9:.1f} GB/s ({sec / 3 * 1e3:.1f} ms per witness); reduced {out} ms per witness")
calc.close()
