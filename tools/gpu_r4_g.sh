#!/bin/bash
# round 4 experiment: everything of a calculator on ONE stream (no cross-stream events), N calculators in flight: how far does it scale, where does it collapse?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label, args...
  L=$1; shift
  timeout 120 python bench.py --gpus 1 --steps 96 --warmup 12 --no-cpu-baseline --no-emission --no-single --no-extra-legs "$@" > $R/r4g_$L.json 2> $R/r4g_$L.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4g_$L.json").read().strip().splitlines()[-1])
    print("$L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", "K_CHK in step", d["roofline"]["avg_ms"], "frac", d["roofline"]["frac"], "host", d["config"]["host_ms_per_step"])
except Exception as e:
    print("$L FAILED", e, open("gpurun_out/r4g_$L.err").read()[-600:])
PY
}
export POB_SCHED_STREAMS=0
run ser6 --sched private --pipeline 6
run ser8 --sched private --pipeline 8
run ser10 --sched private --pipeline 10
run ser12 --sched private --pipeline 12
run ser16 --sched private --pipeline 16
run ser4_b2048 --sched private --pipeline 4 --batch 2048
run ser6b --sched private --pipeline 6
