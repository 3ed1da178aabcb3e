#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out; T=${TAG:-r5f}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/${T}_lone -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/${T}_lone.log 2>&1)
python tools/lone_batch_trace.py $R/${T}_lone/s_results.db > $R/${T}_lone_batch.txt 2>&1; rm -rf $R/${T}_lone; cat $R/${T}_lone_batch.txt
