#!/bin/bash
# round 6, GPU session 24: the riding evaluation without the scheduling barrier behind a load-back (everywhere: "nobar"; in the Poseidon blocks only: "nobarpos"): a lone generation, the loop
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new nobar nobarpos; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new nobar nobarpos; do pt $L 12:3; pt $L 8:3; done
done
} 2>&1 | tee $R/ab_s24_bar.txt
