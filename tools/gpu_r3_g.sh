#!/bin/bash
# round 3, pass g: streaming kernels of both batches in order on one stream -- pipeline tests, bench x2 + bare figure, timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x --timeout=500 -k "spend or pokes or reading or pipelined or two_ranks" > $R/r3g_pytest.log 2>&1; tail -4 $R/r3g_pytest.log
Q="--steps 60 --warmup 6 --no-cpu-baseline --no-emission"
one() { python -c "import json,sys; d=json.load(open('$1')); print(d['ms_per_step'], d['value'], 'kchk', d['roofline']['avg_ms'], 'bare', d['kernel_pipeline_only'], 'single', d['single_calculator']['ms_per_step'])" 2>&1 | tail -1; }
for rep in 1 2; do
  timeout 200 python bench.py $Q > $R/r3g_bench_$rep.json 2> $R/r3g_bench_$rep.err; echo "rep=$rep: $(one $R/r3g_bench_$rep.json)"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/r3g_prof -o r3g -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-emission --no-single > $R/r3g_prof.log 2>&1)
python tools/rocpd_step.py $R/r3g_prof/r3g_results.db 10 0.3 | tail -45
