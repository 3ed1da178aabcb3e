#!/bin/bash
# round 4: after the derived BIT copies of the G side -- pooled two-calculator pipeline vs one in-order stream per calculator (6 in flight), unit times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label, args...
  L=$1; shift
  timeout 120 python bench.py --gpus 1 --steps 96 --warmup 12 --no-cpu-baseline --no-emission --no-single --no-extra-legs "$@" > $R/r4i_$L.json 2> $R/r4i_$L.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4i_$L.json").read().strip().splitlines()[-1])
    print("$L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", "K_CHK in step", d["roofline"]["avg_ms"], "frac", d["roofline"]["frac"], "resident", d["config"]["resident_bytes_per_witness"])
except Exception as e:
    print("$L FAILED", e, open("gpurun_out/r4i_$L.err").read()[-600:])
PY
}
run pool2
POB_SCHED_STREAMS=0 run ser6 --sched private --pipeline 6
POB_SCHED_STREAMS=0 run ser12 --sched private --pipeline 12
timeout 300 python tools/unit_times.py 1024 > $R/r4i_unit_times.txt 2>&1; tail -64 $R/r4i_unit_times.txt
