#!/usr/bin/env python3
"""One production batch of 1 024 witnesses generated and evaluated TWICE by a lone calculator (the second pass is the one to read: buffers, probe tables
and caches are set up): run under `rocprofv3 --pmc ... --kernel-trace` so that every kernel of the step has its SQ counters (tools/pmc_units.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from proof_of_burn_amd import WitnessCalculator, inputs as gen  # noqa: E402

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
NB = int(os.environ.get("POB_PMC_BATCH", "1024"))      # (8192: what every launch costs when it fills the machine -- its THROUGHPUT cost, not its latency)
batch = gen.synthetic_batch(NB, depth=10, seed=0xB0B, distinct_keys=16)
calc = WitnessCalculator(MAIN, max_batch=NB)
if os.environ.get("POB_PMC_INORDER", "7") != "0":
    calc.set_inorder(int(os.environ.get("POB_PMC_INORDER", "7")))          # 7: the schedule bench.py runs (in order, fused launch, evaluation riding with the generation); 3 / 1: without
for _ in range(2):
    res = calc.calculate(batch.inputs, check=True)
    assert all(r.ok and r.check_status == 0 for r in res)
calc.close()
