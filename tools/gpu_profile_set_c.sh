#!/bin/bash
# round-2 profile set with the two-calculator pipeline: GPU tests, the driver's bench line, kernel stats + timeline of the same command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout=800 > $R/r2p_pytest.log 2>&1; tail -3 $R/r2p_pytest.log
timeout 600 python bench.py > $R/r2p_bench_default.json 2> $R/r2p_bench_default.err; cut -c1-200 $R/r2p_bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/r2p_prof -o r2p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-emission > $R/r2p_prof.log 2>&1)
for P in 0 1 0 1; do
  timeout 150 python bench.py --pipeline $P --steps 20 --warmup 4 --no-cpu-baseline --no-emission > $R/r2p_pipe$P.json 2> $R/r2p_pipe$P.err
  echo "pipeline=$P $(cut -c47-60 $R/r2p_pipe$P.json)"
done
