#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
B="timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-emission"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2i_$name.json 2> gpurun_out/r2i_$name.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2i_$name.json')); r=d['roofline']
    print('$name', d['value'], d['ms_per_step'], 'check_pass', r['check_pass']['ms'])
except Exception as e: print('$name ERR', e, open('gpurun_out/r2i_$name.err').read()[-300:])
PY
}
run long8 X=1
run long0 POB_LONG_SPONGE=0
run long8b X=1
run long0b POB_LONG_SPONGE=0
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "main_instantiation or failure_sets or fixture or proof_of_burn_and_wtns" > gpurun_out/r2i_pytest.log 2>&1; tail -2 gpurun_out/r2i_pytest.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2i_prof -o r2i -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-emission > $GRAFT_REPO_ROOT/gpurun_out/r2i_prof.log 2>&1)
