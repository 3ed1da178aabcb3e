#!/bin/bash
# round 3, call 13: strictly phased schedule (evaluation kernel alone | G work of both batches | expansion alone) against the overlapped one
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for ph in 0 1; do
  POB_PHASES=$ph timeout 300 python bench.py --gpus 1 --steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single > gpurun_out/r3k_ph${ph}_$rep.json 2> gpurun_out/r3k_ph${ph}_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3k_ph${ph}_$rep.json").read().strip().splitlines()[-1])
print("phases=$ph rep=$rep", d["ms_per_step"], d["value"], "kchk", d["roofline"]["avg_ms"], "frac", d["roofline"]["frac"])
PY
done; done 2>&1 | tee gpurun_out/r3k_summary.txt
