#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
B="timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2h_$name.json 2> gpurun_out/r2h_$name.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2h_$name.json')); r=d['roofline']
    print('$name', d['value'], d['ms_per_step'], 'check_pass', r['check_pass']['ms'], 'emission GB/s', d['emission']['GB_per_s'], d['emission']['ms_per_witness'])
except Exception as e: print('$name ERR', e, open('gpurun_out/r2h_$name.err').read()[-300:])
PY
}
run A_default X=1
run B_kfirst4 "POB_CHECK_PLAN=K;1,2,7,5;3,6,0;4"
run D_sel_k4 "POB_CHECK_PLAN=2,K;1,7,5;3,6,0;4"
run E_4 "POB_CHECK_PLAN=1,2,K;7,5,3;6,0;4"
run G_noprobe POB_EMIT_PROBE=0
timeout 900 python -m pytest tests -m gpu -q -x --timeout=800 > gpurun_out/r2h_pytest.log 2>&1; tail -3 gpurun_out/r2h_pytest.log
