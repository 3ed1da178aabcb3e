#!/bin/bash
# round 6, GPU session 17: how much of the device k_rounds_gc should occupy beside the other calculators: rounds per wavefront 8 / 12 / 24 (4 032 / 2 688 / 1 344 wavefronts per launch),
# wavefronts per SIMD 4 / 2 / 1; the service loop with 12 and 16 in flight, three interleaved rounds
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new kr24 kr12 w2 w1 kr24w2; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new kr24 kr12 w2 w1 kr24w2; do pt $L 12:3; done
  pt new 16:3; pt kr24 16:3; pt w2 16:3
done
} 2>&1 | tee $R/ab_s17_occ.txt
