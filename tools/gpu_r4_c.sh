#!/bin/bash
# round 4 experiment: scheduling once the Keccak kernels are small -- pool + gate (round 3) vs own side streams per calculator, no gate, 2..4 calculators
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label, args...
  L=$1; shift
  timeout 120 python bench.py --gpus 1 --steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single "$@" > $R/r4c_$L.json 2> $R/r4c_$L.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4c_$L.json").read().strip().splitlines()[-1])
    print("$L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", "K_CHK in step", d["roofline"]["avg_ms"])
except Exception as e:
    print("$L FAILED", e, open("gpurun_out/r4c_$L.err").read()[-600:])
PY
}
run pool2
run priv2 --sched private
run priv3 --sched private --pipeline 3
POB_SCHED_STREAMS=3 run priv3s3 --sched private --pipeline 3
POB_SCHED_STREAMS=2 run priv3s2 --sched private --pipeline 3
POB_SCHED_STREAMS=2 run priv4s2 --sched private --pipeline 4
POB_SCHED_STREAMS=1 run priv4s1 --sched private --pipeline 4
GPU_MAX_HW_QUEUES=24 run priv3q24 --sched private --pipeline 3
run pool2b
