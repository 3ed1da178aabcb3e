#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2c_$name.json 2> gpurun_out/r2c_$name.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2c_$name.json')); r=d['roofline']
    print('$name', d['value'], d['ms_per_step'], 'check_pass', r['check_pass']['ms'])
except Exception as e: print('$name ERR', e, open('gpurun_out/r2c_$name.err').read()[-300:])
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "spend or corruption or pokes or failure_sets" > gpurun_out/r2c_pytest.log 2>&1; tail -2 gpurun_out/r2c_pytest.log
for H in 1 2; do
run early_h$H POB_BENCH_HALVES=$H
run late_h$H POB_BENCH_HALVES=$H POB_CHECK_EARLY_K=0
run early_planB_h$H POB_BENCH_HALVES=$H "POB_CHECK_PLAN=1,2;7,5,3;4,6,0"
run early_planC_h$H POB_BENCH_HALVES=$H "POB_CHECK_PLAN=4,1,2;7,5,3;6,0"
done
run early_h2_q12 POB_BENCH_HALVES=2 GPU_MAX_HW_QUEUES=12
run early_h2_q16 POB_BENCH_HALVES=2 GPU_MAX_HW_QUEUES=16
run early_h1_q16 POB_BENCH_HALVES=1 GPU_MAX_HW_QUEUES=16
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c_prof -o r2c -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --halves 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2c_prof.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2c_prof2 -o r2c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --halves 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2c_prof2.log 2>&1)
