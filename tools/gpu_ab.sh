#!/bin/bash
# Parametrised A/B runner for the GPU box: every argument is one bench.py variant ("label: args"), run in the given order, `rounds` times over (interleaved on ONE box:
# boxes differ by up to 5 %), one summary line per run.  Works on HEAD: variants are bench.py arguments and documented environment variables only.
#   gpurun -- 'ROUNDS=3 bash tools/gpu_ab.sh "io4: --pipeline 4" "io8: --pipeline 8" "tracks: --schedule tracks"'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
for r in $(seq 1 ${ROUNDS:-1}); do
  for v in "$@"; do
    L=${v%%:*}; A=${v#*:}
    timeout 180 python bench.py --gpus 1 --steps ${STEPS:-96} --warmup ${WARMUP:-12} --no-cpu-baseline --no-emission --no-single --no-extra-legs $A > $R/ab_${L}_$r.json 2> $R/ab_${L}_$r.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab_${L}_$r.json").read().strip().splitlines()[-1])
    print("round $r $L:", d["ms_per_step"], "ms/step", d["value"], "witnesses/s; round evaluation in step", d["roofline"]["avg_ms"], "ms = frac", d["roofline"]["frac"])
except Exception as e:
    print("round $r $L FAILED", e, open("gpurun_out/ab_${L}_$r.err").read()[-600:])
PY
  done
done 2>&1 | tee $R/ab_summary.txt
