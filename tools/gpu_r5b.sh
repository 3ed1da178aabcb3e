#!/bin/bash
# SQ counters of one batch (in-order schedule), two passes of 4 counters each
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out; T=${TAG:-r5b}
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY --kernel-trace -d $R/${T}_sq1 -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/${T}_sq1.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/${T}_sq2 -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/${T}_sq2.log 2>&1)
python tools/pmc_units.py $R/${T}_sq1/s_results.db > $R/${T}_sq_a.txt 2>&1; python tools/pmc_units.py $R/${T}_sq2/s_results.db > $R/${T}_sq_b.txt 2>&1
rm -rf $R/${T}_sq1 $R/${T}_sq2
head -40 $R/${T}_sq_a.txt | cut -c1-160
