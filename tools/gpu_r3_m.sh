#!/bin/bash
# round 3, call 16: queued emission -- test, then rates (full / reduced), traces of both
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "queued_emission or spend_wtns or reduced_witness or failed_witness" 2>&1 | tail -5
timeout 300 python tools/emit_trace.py run 2>&1 | tail -1
timeout 300 python tools/emit_trace.py run reduced 2>&1 | tail -1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $R/r3m_full -o e -- python $GRAFT_REPO_ROOT/tools/emit_trace.py run > $R/r3m_full.log 2>&1)
python tools/emit_trace.py $(find $R/r3m_full -name "e_results.db" | head -1) > $R/r3m_full_timeline.txt 2>&1; tail -1 $R/r3m_full.log; head -8 $R/r3m_full_timeline.txt; tail -5 $R/r3m_full_timeline.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $R/r3m_red -o e -- python $GRAFT_REPO_ROOT/tools/emit_trace.py run reduced > $R/r3m_red.log 2>&1)
python tools/emit_trace.py $(find $R/r3m_red -name "e_results.db" | head -1) > $R/r3m_red_timeline.txt 2>&1; tail -1 $R/r3m_red.log; cat $R/r3m_red_timeline.txt
