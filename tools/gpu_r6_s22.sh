#!/bin/bash
# round 6, GPU session 22: which RLP units ride: all but the leaf's head U_RL_A (in-tree) against none of the family ("rlplain": block 21's final form), 12 / 8 / 16 in flight, three rounds
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new rlplain; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new rlplain; do pt $L 12:3; pt $L 8:3; pt $L 16:3; done
done
} 2>&1 | tee $R/ab_s22_rl.txt
