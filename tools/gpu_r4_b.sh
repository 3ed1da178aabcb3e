#!/bin/bash
# round 4: the rest of the GPU tests on the compact layout, step timeline of the driver's command, witnesses per calculator
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/r4b_prof -o r4b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-emission --no-single > $R/r4b_prof.log 2>&1)
python tools/rocpd_summary.py $R/r4b_prof/r4b_results.db > $R/r4b_kernel_stats.txt 2>&1; head -40 $R/r4b_kernel_stats.txt
python tools/rocpd_step.py $R/r4b_prof/r4b_results.db 20 0.02 > $R/r4b_step_timeline.txt 2>&1; tail -3 $R/r4b_step_timeline.txt
rm -rf $R/r4b_prof
for b in 1024 2048 4096; do
  timeout 200 python bench.py --gpus 1 --batch $b --steps $((81920 / b)) --warmup 4 --distinct-batches 2 --no-cpu-baseline --no-emission --no-single > $R/r4b_w$b.json 2> $R/r4b_w$b.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4b_w$b.json").read().strip().splitlines()[-1])
    print("batch $b:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s", round(d["ms_per_step"] * 1024 / $b, 3), "ms per 1024")
except Exception as e:
    print("batch $b FAILED", e, open("gpurun_out/r4b_w$b.err").read()[-800:])
PY
done 2>&1 | tee $R/r4b_w_summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=1200 > $R/r4b_pytest.log 2>&1; echo "pytest rc=$?" >> $R/r4b_pytest.log; tail -5 $R/r4b_pytest.log
