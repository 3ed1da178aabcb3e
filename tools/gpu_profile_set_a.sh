#!/bin/bash
# round 2 profile set: default bench, kernel trace + stats, per-kind / per-family times, PMC passes (FETCH_SIZE, WRITE_SIZE separately)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py > $R/r2e_bench_default.json 2> $R/r2e_bench_default.err; cut -c1-200 $R/r2e_bench_default.json
timeout 300 python bench.py --halves 2 --no-cpu-baseline > $R/r2e_bench_h2.json 2> $R/r2e_bench_h2.err; cut -c1-200 $R/r2e_bench_h2.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/r2e_prof -o r2e -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/r2e_prof.log 2>&1)
timeout 300 python tools/unit_times.py 1024 > $R/r2e_unit_times.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/r2e_pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $R/r2e_pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/r2e_pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $R/r2e_pmc_write.log 2>&1)
ls $R/r2e_prof $R/r2e_pmc_fetch $R/r2e_pmc_write
