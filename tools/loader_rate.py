#!/usr/bin/env python3
"""input.json texts -> packed rows (byte form) through the native loader pool, for several pool widths: witnesses/s on this host (no GPU needed).
    python tools/loader_rate.py [batch] [threads,threads,...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proof_of_burn_amd import witness as W, inputs as gen  # noqa: E402

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
widths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8, 16, 32, 64, 128, 0]
try:
    quota = open("/sys/fs/cgroup/cpu.max").read().strip()
except Exception:
    quota = "?"
print(f"cpu_count {os.cpu_count()}  affinity {len(os.sched_getaffinity(0))}  cgroup cpu.max {quota}  LOCAL_WORLD_SIZE {os.environ.get('LOCAL_WORLD_SIZE')}")
batch = gen.synthetic_batch(min(B, 256), depth=10, seed=0xB0B, distinct_keys=4)
texts = W.TextBatch([json.dumps(batch.inputs[i % len(batch.inputs)]).encode() for i in range(B)])
print(f"batch {B}, {sum(len(t) for t in texts.raw) / B / 1024:.1f} KB of text per witness")
for t in widths:
    W.pack_json8(MAIN, texts, threads=t)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        W.pack_json8(MAIN, texts, threads=t)
    dt = (time.perf_counter() - t0) / reps
    print(f"threads {t:4d}: {dt * 1e3:8.3f} ms per batch  {B / dt:12.0f} witnesses/s")
