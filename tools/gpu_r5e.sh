#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:---no-cpu-baseline} > $R/r5e_bench.json 2> $R/r5e_bench.err; tail -3 $R/r5e_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5e_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "h2d", d["config"]["h2d_bytes_per_step"], "json->packed", d["config"]["json_to_packed_witnesses_per_s"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_ms", "frac_alone", "avg_ms_alone", "bytes_per_launch")}, d["roofline"]["check_pass"]["ms"], d["roofline"]["gen_kernel"])
for k in ("e2e_from_json", "deeper_pipeline", "depth16", "strong_slice", "single_calculator", "tracks_pipeline", "kernel_pipeline_only"):
    print(k, {a: b for a, b in (d.get(k) or {}).items() if a != "what"})
PY
