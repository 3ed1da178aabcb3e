#!/bin/bash
# round 3, call 15: where the emission's time goes -- kernel + copy trace of full and reduced emission
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $R/r3l_full -o e -- python $GRAFT_REPO_ROOT/tools/emit_trace.py run > $R/r3l_full.log 2>&1)
python tools/emit_trace.py $(find $R/r3l_full -name "e_results.db" | head -1) > $R/r3l_full_timeline.txt 2>&1; tail -3 $R/r3l_full.log; head -40 $R/r3l_full_timeline.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $R/r3l_red -o e -- python $GRAFT_REPO_ROOT/tools/emit_trace.py run reduced > $R/r3l_red.log 2>&1)
python tools/emit_trace.py $(find $R/r3l_red -name "e_results.db" | head -1) > $R/r3l_red_timeline.txt 2>&1; tail -3 $R/r3l_red.log; cat $R/r3l_red_timeline.txt
