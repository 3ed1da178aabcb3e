#!/bin/bash
# round 3, second GPU pass: every GPU test with the new API (records after evaluation, async upload, reduced emission, RCCL world 1, differential),
# the service-loop bench, and the persistent-round-kernel A/B (interleaved on one box)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=1200 --durations=12 > $R/r3b_pytest.log 2>&1; tail -22 $R/r3b_pytest.log
timeout 400 python bench.py --no-cpu-baseline > $R/r3b_bench.json 2> $R/r3b_bench.err; cut -c1-180 $R/r3b_bench.json; tail -3 $R/r3b_bench.err
Q="--steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single"
for rep in 1 2; do
  for cfg in "0 0" "2 0" "2 3" "2 4" "0 3" "1 3"; do
    set -- $cfg
    POB_KR_PERSIST_CHECK=$1 POB_KR_PERSIST_GEN=$2 timeout 200 python bench.py $Q > $R/r3b_ab_c$1_g$2_$rep.json 2> $R/r3b_ab_c$1_g$2_$rep.err
    echo "persist check=$1 gen=$2 rep=$rep: $(python -c "import json,sys; d=json.load(open('$R/r3b_ab_c$1_g$2_$rep.json')); print(d['ms_per_step'], d['value'], d['roofline']['avg_ms'], d['roofline']['avg_ms_alone'], d['roofline']['gen_kernel']['avg_ms'])" 2>&1 | tail -1)"
  done
done
