#!/bin/bash
# round 3, call 18: evaluation families' streams at the low priority (the device has two levels: 0 and -1) -- five interleaved pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3 4 5; do for pr in 0 2; do
  POB_CHK_PRIO=$pr timeout 300 python bench.py --gpus 1 --steps 100 --warmup 6 --no-cpu-baseline --no-emission --no-single > gpurun_out/r3n_p${pr}_$rep.json 2> gpurun_out/r3n_p${pr}_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3n_p${pr}_$rep.json").read().strip().splitlines()[-1])
print("chk_prio=$pr rep=$rep", d["ms_per_step"], d["value"], "kchk", d["roofline"]["avg_ms"])
PY
done; done 2>&1 | tee gpurun_out/r3n_summary.txt
