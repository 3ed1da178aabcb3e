#!/bin/bash
# round 4: in-order level-merged schedule -- calculators in flight sweep, the driver's command, a timeline, then the whole GPU test suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label, args...
  L=$1; shift
  timeout 120 python bench.py --gpus 1 --steps 96 --warmup 12 --no-cpu-baseline --no-emission --no-single --no-extra-legs "$@" > $R/r4j_$L.json 2> $R/r4j_$L.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4j_$L.json").read().strip().splitlines()[-1])
    print("$L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", "K_CHK in step", d["roofline"]["avg_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("$L FAILED", e, open("gpurun_out/r4j_$L.err").read()[-600:])
PY
}
for n in 2 3 4 6 8 12; do run io$n --pipeline $n; done
run tracks --schedule tracks
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r4j_bench_driver.json 2> $R/r4j_bench_driver.err; cut -c1-300 $R/r4j_bench_driver.json; tail -3 $R/r4j_bench_driver.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/r4j_prof -o r4j -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 24 --warmup 6 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/r4j_prof.log 2>&1)
python tools/rocpd_step.py $R/r4j_prof/r4j_results.db 40 0.0 > $R/r4j_step_timeline.txt 2>&1; tail -2 $R/r4j_step_timeline.txt
python tools/rocpd_summary.py $R/r4j_prof/r4j_results.db > $R/r4j_kernel_stats.txt 2>&1
rm -rf $R/r4j_prof
timeout 1800 python -m pytest tests -m gpu -q --timeout=1500 > $R/r4j_pytest.log 2>&1; echo "pytest rc=$?" >> $R/r4j_pytest.log; tail -6 $R/r4j_pytest.log
