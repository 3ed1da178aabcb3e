#!/bin/bash
# quick look at a build: the lone-batch launch trace and the 96-step loop with 4 and 8 calculators in flight (ROUNDS interleaved rounds)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/q_lone -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/q_lone.log 2>&1)
python tools/lone_batch_trace.py $R/q_lone/s_results.db > $R/q_lone_batch.txt 2>&1; rm -rf $R/q_lone; cat $R/q_lone_batch.txt | cut -c12-90
ROUNDS=${ROUNDS:-2} STEPS=96 bash tools/gpu_ab.sh "p4: --pipeline 4" "p8: --pipeline 8"
