#!/bin/bash
# round 6, GPU session 4: per-launch THROUGHPUT cost: the lone-batch launch trace of a calculator pass over 8 192 witnesses (every launch fills the machine)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && POB_PMC_BATCH=8192 POB_PMC_INORDER=1 timeout 600 rocprofv3 --kernel-trace -d $R/s4_lone -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/s4_lone.log 2>&1)
python tools/lone_batch_trace.py $R/s4_lone/s_results.db > $R/s4_lone_batch_8192.txt 2>&1; rm -rf $R/s4_lone; cat $R/s4_lone_batch_8192.txt | cut -c12-100
