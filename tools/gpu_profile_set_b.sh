#!/bin/bash
# final round-2 profile set: default bench (driver's command line), kernel trace + stats, PMC passes (FETCH_SIZE / WRITE_SIZE separately,
# as MI355X_MICROARCH.md prescribes), the GPU test suite.   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/gpu_profile_set_b.sh'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py > $R/r2o_bench_default.json 2> $R/r2o_bench_default.err; cut -c1-160 $R/r2o_bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/r2o_prof -o r2o -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-emission > $R/r2o_prof.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/r2o_pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission > $R/r2o_pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/r2o_pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission > $R/r2o_pmc_write.log 2>&1)
timeout 900 python -m pytest tests -m gpu -q -x --timeout=800 > $R/r2o_pytest.log 2>&1; tail -2 $R/r2o_pytest.log
