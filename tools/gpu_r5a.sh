#!/bin/bash
# round 5, call 1: evaluator parity on the multi-round walker, then the k_rounds_check variant sweep (one process), then the driver's command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -n 4 --timeout=800 -k "spend or constraint_evaluator or inorder_schedule or main_instantiation_batch or different_streams" > $R/r5a_tests.txt 2>&1; echo "pytest rc=$?" >> $R/r5a_tests.txt; tail -5 $R/r5a_tests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-emission --no-single --no-extra-legs --x-kchk-sweep 1200,4400,4408,4412,4416,4420,4316,4324,6412,6416,3412,3416,8416,2416,4508,4512 > $R/r5a_sweep.json 2> $R/r5a_sweep.err; grep SWEEP $R/r5a_sweep.err | cut -c1-200; tail -3 $R/r5a_sweep.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/r5a_bench_driver.json 2> $R/r5a_bench_driver.err; cut -c1-300 $R/r5a_bench_driver.json; tail -2 $R/r5a_bench_driver.err
