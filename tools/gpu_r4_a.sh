#!/bin/bash
# round 4, first GPU pass on the compact Keccak layout: parity tests, the driver's bench command, per-unit times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=1200 > $R/r4a_pytest.log 2>&1; echo "pytest rc=$?" >> $R/r4a_pytest.log; tail -5 $R/r4a_pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r4a_bench_driver.json 2> $R/r4a_bench_driver.err; cut -c1-300 $R/r4a_bench_driver.json; tail -3 $R/r4a_bench_driver.err
timeout 300 python tools/unit_times.py 1024 > $R/r4a_unit_times.txt 2>&1; tail -30 $R/r4a_unit_times.txt
