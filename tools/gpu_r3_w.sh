#!/bin/bash
# round 3, last call: witnesses per calculator on the final build (1024 / 2048 / 4096)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 1024 2048 4096; do
  timeout 200 python bench.py --gpus 1 --batch $b --steps $((40960 / b)) --warmup 4 --distinct-batches 2 --no-cpu-baseline --no-emission --no-single > gpurun_out/r3w_$b.json 2> gpurun_out/r3w_$b.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3w_$b.json").read().strip().splitlines()[-1])
    print("batch $b:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", round(d["ms_per_step"] * 1024 / $b, 3), "ms per 1024")
except Exception as e:
    print("batch $b FAILED", e, open("gpurun_out/r3w_$b.err").read()[-800:])
PY
done 2>&1 | tee gpurun_out/r3w_summary.txt
