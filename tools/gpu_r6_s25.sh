#!/bin/bash
# round 6, GPU session 25: the riding evaluation of a BATCH of small values as stores | loads | compares (sm_commit -> GenPT::put_batch: one memory round trip per batch instead of one
# per value) against one pending compare per value ("prev"); and the loop with the evaluation as a separate pass as a process of its own (bench.py --fused 1)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new prev; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new prev; do pt $L 12:3; pt $L 8:3; pt $L 16:3; done
  pt new 12:1
done
} 2>&1 | tee $R/ab_s25_batch.txt
