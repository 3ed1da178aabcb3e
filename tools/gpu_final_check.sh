#!/bin/bash
# final state of a round: every GPU test, then the driver's bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x --timeout=500 > $R/final_pytest.log 2>&1; tail -3 $R/final_pytest.log
timeout 300 python bench.py > $R/final_bench_default.json 2> $R/final_bench_default.err; cut -c1-160 $R/final_bench_default.json
