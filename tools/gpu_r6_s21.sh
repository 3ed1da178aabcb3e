#!/bin/bash
# round 6, GPU session 21: the evaluation of the G units riding with their generation, final form (RLP units on the plain policy: no spill; field elements compared at once):
# GPU tests of the riding evaluation, then the loop with it against without ("noride"), 12 / 8 / 16 in flight, three interleaved rounds
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "riding or inorder_schedule or different_streams or detects_corruption or failure_sets_match_oracle_at_the_production" --durations=6 ) > $R/s21_tests.txt 2>&1; tail -12 $R/s21_tests.txt
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new noride; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new noride; do pt $L 12:3; pt $L 8:3; pt $L 16:3; done
done
} 2>&1 | tee $R/ab_s21_rideg.txt
