#!/bin/bash
# round 3: the sponge-chain evaluation on an evaluation stream instead of in order on the streaming stream -- three interleaved pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for v in 0 1; do
  POB_CHAIN_SIDE=$v timeout 100 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-emission --no-single > gpurun_out/r3y_${v}_$rep.json 2> gpurun_out/r3y_${v}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3y_${v}_$rep.json").read().strip().splitlines()[-1])
    print("chain_side=$v rep=$rep", d["ms_per_step"], d["value"], "kchk", d["roofline"]["avg_ms"])
except Exception as e:
    print("chain_side=$v rep=$rep FAILED", open("gpurun_out/r3y_${v}_$rep.err").read()[-600:])
PY
done; done 2>&1 | tee gpurun_out/r3y_summary.txt
