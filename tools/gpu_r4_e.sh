#!/bin/bash
# round 4 experiment: do independent calculators overlap at all?  timeline of two ungated calculators with own streams; everything of a calculator on ONE stream, 1..6 calculators
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label, args...
  L=$1; shift
  timeout 120 python bench.py --gpus 1 --steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single "$@" > $R/r4e_$L.json 2> $R/r4e_$L.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4e_$L.json").read().strip().splitlines()[-1])
    print("$L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", "host ms per step", d["config"]["host_ms_per_step"])
except Exception as e:
    print("$L FAILED", e, open("gpurun_out/r4e_$L.err").read()[-600:])
PY
}
export POB_SCHED_STREAMS=0
run ser1 --sched private --pipeline 0
run ser2 --sched private --pipeline 2
run ser3 --sched private --pipeline 3
run ser4 --sched private --pipeline 4
run ser6 --sched private --pipeline 6
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/r4e_prof -o r4e -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-emission --no-single --sched private --pipeline 4 > $R/r4e_prof.log 2>&1)
python tools/rocpd_step.py $R/r4e_prof/r4e_results.db 20 0.0 > $R/r4e_step_timeline_ser4.txt 2>&1; tail -2 $R/r4e_step_timeline_ser4.txt
rm -rf $R/r4e_prof
unset POB_SCHED_STREAMS
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/r4e_prof -o r4e -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-emission --no-single --sched private --pipeline 2 > $R/r4e_prof.log 2>&1)
python tools/rocpd_step.py $R/r4e_prof/r4e_results.db 20 0.02 > $R/r4e_step_timeline_priv2.txt 2>&1; tail -2 $R/r4e_step_timeline_priv2.txt
rm -rf $R/r4e_prof
