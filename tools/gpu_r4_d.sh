#!/bin/bash
# round 4 experiment: is the 1024-witness step bound by the HOST's enqueue rate?  host time in pob_generate / pob_constraint_check / waiting per step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label, args...
  L=$1; shift
  timeout 120 python bench.py --gpus 1 --steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single "$@" > $R/r4d_$L.json 2> $R/r4d_$L.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4d_$L.json").read().strip().splitlines()[-1])
    print("$L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", "host ms per step", d["config"]["host_ms_per_step"])
except Exception as e:
    print("$L FAILED", e, open("gpurun_out/r4d_$L.err").read()[-600:])
PY
}
run pool2
run priv2 --sched private
run priv4s2 --sched private --pipeline 4
run pool2_2048 --batch 2048
run single --pipeline 0
