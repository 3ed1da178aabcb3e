#!/bin/bash
# round 4: the round evaluation of an in-order calculator on the device's high-priority stream vs on the calculator's own stream; the two failed tests again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label, args...
  L=$1; shift
  timeout 120 python bench.py --gpus 1 --steps 96 --warmup 12 --no-cpu-baseline --no-emission --no-single --no-extra-legs "$@" > $R/r4k_$L.json 2> $R/r4k_$L.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4k_$L.json").read().strip().splitlines()[-1])
    print("$L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", "K_CHK in step", d["roofline"]["avg_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("$L FAILED", e, open("gpurun_out/r4k_$L.err").read()[-600:])
PY
}
for n in 4 6 8 12; do run prio$n --pipeline $n; POB_X_NO_PRIO=1 run noprio$n --pipeline $n; done
run prio6_k20 --pipeline 6 --steps 20 --warmup 5
run prio12_k20 --pipeline 12 --steps 20 --warmup 5
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=800 -k "selfcheck or field_inversions" > $R/r4k_pytest.log 2>&1; tail -3 $R/r4k_pytest.log
