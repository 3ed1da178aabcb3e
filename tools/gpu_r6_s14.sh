#!/bin/bash
# round 6, GPU session 14: where k_rounds_gc's loads come from (FETCH_SIZE / WRITE_SIZE / TCC hits of one lone batch with the round blocks evaluated by the launch that writes them),
# and the number of arrays between a store and its compare, non-temporal loads of the stored states, rounds per wavefront in the service loop
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
export POB_PMC_INORDER=7
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  T=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/s14_$T -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/s14_$T.log 2>&1)
  python tools/pmc_units.py $R/s14_$T/s_results.db 2>&1 | grep -i "kernel\|k_rounds\|k_chain" | cut -c1-160; rm -rf $R/s14_$T
done 2>&1 | tee $R/s14_pmc_gc.txt
unset POB_PMC_INORDER
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new dp2 nt kr24; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new dp8 dp2 dp6 nt kr24; do pt $L 12:3; done
  pt new 16:3; pt dp2 16:3; pt new 12:1
done
} 2>&1 | tee $R/ab_s14_gc.txt
