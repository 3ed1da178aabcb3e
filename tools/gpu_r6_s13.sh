#!/bin/bash
# round 6, GPU session 13: k_rounds_gc as store + load-back per gate (DP arrays between a store and its compare: the footprint stays in L2) against its first version (a whole
# round generated, then evaluated: "park") and against expansion + evaluation as two launches; builds with other DP / rounds per wavefront; the service loop with and without
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new dp4 dp12w3 kr4 dp16w3 park; do pt $L 4:3 48 --alone; done
for r in 1 2; do
  for p in 12:1 12:3 8:1 8:3 4:1 4:3 6:3 16:3; do pt new $p; done
  pt dp4 12:3; pt dp12w3 12:3; pt kr4 12:3; pt park 12:3
done
} 2>&1 | tee $R/ab_s13_gc.txt
