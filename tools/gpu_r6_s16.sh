#!/bin/bash
# round 6, GPU session 16: the midRound states inside a chunk of k_rounds_gc stored by that launch (and loaded back from L2) instead of by k_chain (and loaded from HBM):
# "midall" = a build whose k_chain stores every state as before; GPU tests of the riding evaluation first; PMC of the launch
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "riding or inorder_schedule or different_streams or detects_corruption" --durations=5 ) > $R/s16_tests.txt 2>&1; tail -6 $R/s16_tests.txt
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new midall; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new midall; do pt $L 12:3; pt $L 16:3; pt $L 8:3; done
done
} 2>&1 | tee $R/ab_s16_mid.txt
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/s16_$C -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/s16_$C.log 2>&1)
  python tools/pmc_units.py $R/s16_$C/s_results.db 2>&1 | grep -i "kernel\|k_rounds\|k_chain\|pos_chain" | cut -c1-160; rm -rf $R/s16_$C
done 2>&1 | tee $R/s16_pmc_gc.txt
