#!/bin/bash
# round 3: the streaming stream at the high priority again, now that the G side is 40 % lighter -- four interleaved pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3 4; do for v in 0 1; do
  POB_K_HI=$v timeout 200 python bench.py --gpus 1 --steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single > gpurun_out/r3x_${v}_$rep.json 2> gpurun_out/r3x_${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3x_${v}_$rep.json").read().strip().splitlines()[-1])
print("k_hi=$v rep=$rep", d["ms_per_step"], d["value"], "kchk", d["roofline"]["avg_ms"])
PY
done; done 2>&1 | tee gpurun_out/r3x_summary.txt
