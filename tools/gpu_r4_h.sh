#!/bin/bash
# round 4: where do the wave cycles of a batch go?  SQ counters per kernel (each unit kind alone: tools/unit_times.py under rocprofv3 --pmc)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace -d $R/r4h_pmc -o u -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/r4h_pmc.log 2>&1)
python tools/pmc_units.py $R/r4h_pmc/u_results.db > $R/r4h_pmc_units.txt 2>&1; head -70 $R/r4h_pmc_units.txt | cut -c1-200
rm -rf $R/r4h_pmc
