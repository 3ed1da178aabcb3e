#!/bin/bash
# round 2, GPU pass 2: kernel trace of one step sequence + evaluation stream plans
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2b_$name.json 2> gpurun_out/r2b_$name.err; python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r2b_$name.json')); r=d['roofline']
    print('$name', d['value'], d['ms_per_step'], 'check_pass', r['check_pass']['ms'])
except Exception as e: print('$name ERR', e)
PY
}
for H in 1 2; do
run default_h$H POB_BENCH_HALVES=$H
run strict_h$H POB_BENCH_HALVES=$H "POB_CHECK_PLAN=1,2;7,5,3;4,6,0;K"
run kfirst_h$H POB_BENCH_HALVES=$H "POB_CHECK_PLAN=K,1,2;7,5,3;4,6,0"
run allside_h$H POB_BENCH_HALVES=$H "POB_CHECK_PLAN=K;7,5,3,1;4,6,0,2"
run allside2_h$H POB_BENCH_HALVES=$H "POB_CHECK_PLAN=K;4,7,5,1;3,6,0,2"
run mix_h$H POB_BENCH_HALVES=$H "POB_CHECK_PLAN=2,K;4,7,1;3,5,6,0"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b_prof -o r2b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --halves 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2b_prof.log 2>&1)
ls gpurun_out/r2b_prof | head
