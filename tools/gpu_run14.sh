#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
B="timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-emission"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2l_$name.json 2> gpurun_out/r2l_$name.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2l_$name.json')); r=d['roofline']
    print('$name', d['value'], d['ms_per_step'], 'check_pass', r['check_pass']['ms'])
except Exception as e: print('$name ERR', e, open('gpurun_out/r2l_$name.err').read()[-300:])
PY
}
run default X=1
run old "POB_CHECK_PLAN=1,2,K;7,5,3,6,0;4"
run v2 "POB_CHECK_PLAN=K;7,5,3,2;4,6,0,1"
run v3 "POB_CHECK_PLAN=K;3,5,1;7,6,0,2;4"
run v4 "POB_CHECK_PLAN=K;3,1,2;7,5;4,6,0"
run default_b X=1
