#!/bin/bash
# first GPU pass of round 2: parity tests, bench variants, per-family times, kernel trace
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=1200 > gpurun_out/r2_pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest1.log
for H in 2 1; do
  timeout 300 python bench.py --steps 10 --warmup 2 --halves $H --no-cpu-baseline > gpurun_out/r2_bench_h$H.json 2> gpurun_out/r2_bench_h$H.err
done
POB_CHECK_OVERLAP=1 timeout 300 python bench.py --steps 10 --warmup 2 --halves 2 --no-cpu-baseline > gpurun_out/r2_bench_h2_ovl.json 2> gpurun_out/r2_bench_h2_ovl.err
GPU_MAX_HW_QUEUES=4 timeout 300 python bench.py --steps 10 --warmup 2 --halves 2 --no-cpu-baseline > gpurun_out/r2_bench_h2_q4.json 2> gpurun_out/r2_bench_h2_q4.err
timeout 300 python bench.py --steps 10 --warmup 2 --halves 4 --no-cpu-baseline > gpurun_out/r2_bench_h4.json 2> gpurun_out/r2_bench_h4.err
timeout 300 python tools/unit_times.py 1024 > gpurun_out/r2_unit_times1.txt 2>&1
tail -3 gpurun_out/r2_pytest1.log; cat gpurun_out/r2_bench_h*.json | cut -c1-400
