#!/bin/bash
# round 3, call 22: pob_generate / pob_constraint_check of a single handle captured as HIP graphs (per input / record buffer) and replayed -- interleaved pairs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3 4; do for g in 0 1; do
  POB_GRAPHS=$g timeout 300 python bench.py --gpus 1 --pipeline 0 --steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single > gpurun_out/r3q_g${g}_$rep.json 2> gpurun_out/r3q_g${g}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3q_g${g}_$rep.json").read().strip().splitlines()[-1])
    print("graphs=$g rep=$rep", d["ms_per_step"], d["value"], "validated", d["config"]["validated_witnesses"])
except Exception as e:
    print("graphs=$g rep=$rep FAILED", e); print(open("gpurun_out/r3q_g${g}_$rep.err").read()[-1500:])
PY
done; done 2>&1 | tee gpurun_out/r3q_summary.txt
