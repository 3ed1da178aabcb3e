#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
B="timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-emission"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2k_$name.json 2> gpurun_out/r2k_$name.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2k_$name.json')); r=d['roofline']
    print('$name', d['value'], d['ms_per_step'], 'check_pass', r['check_pass']['ms'])
except Exception as e: print('$name ERR', e, open('gpurun_out/r2k_$name.err').read()[-300:])
PY
}
run default X=1
run kfirst "POB_CHECK_PLAN=K,1,2;7,5,3,6,0;4"
run kfirst4 "POB_CHECK_PLAN=K;1,2,7,5;3,6,0;4"
run k_2nd "POB_CHECK_PLAN=1,K,2;7,5,3,6,0;4"
run default_b X=1
timeout 900 python -m pytest tests -m gpu -q -x --timeout=800 > gpurun_out/r2k_pytest.log 2>&1; tail -2 gpurun_out/r2k_pytest.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2k_prof -o r2k -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-emission > $GRAFT_REPO_ROOT/gpurun_out/r2k_prof.log 2>&1)
timeout 200 python tools/unit_times.py 1024 > gpurun_out/r2k_unit_times.txt 2>&1
