#!/bin/bash
# round 6, GPU session 2: fused launches -- parity on hardware, the lone-batch launch traces of both plans, then the loop A/B (fused 0 / 1 x depths)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spend_wtns_bit_exact or reference_suite_spend or main_instantiation_batch or production_batch_payloads_beyond_group_0 or inorder_schedule_equals or different_streams or gadget_mains_payload_and_evaluator or failure_sets_match_oracle_at_the_production" 2>&1 | tail -5) > $R/s2_tests.txt 2>&1
cat $R/s2_tests.txt
for M in 3 1; do
  (cd /tmp && POB_PMC_INORDER=$M timeout 300 rocprofv3 --kernel-trace -d $R/s2_lone$M -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/s2_lone$M.log 2>&1)
  python tools/lone_batch_trace.py $R/s2_lone$M/s_results.db > $R/s2_lone_batch_mode$M.txt 2>&1; rm -rf $R/s2_lone$M; cat $R/s2_lone_batch_mode$M.txt | cut -c12-100
done
LIBS="new=" POINTS="4:0,8:0,12:0,4:1,8:1,12:1" ROUNDS=2 EXTRA="--alone" TAG=s2_fused bash tools/gpu_r6_ab.sh
