#!/bin/bash
# round 6, GPU session 20: the evaluation of the G units riding with their generation (policy.hpp GenPT<true>, poseidon_wide.hpp PosWideT<true>): GPU tests of the in-order modes, then the
# loop with it (the merged generation kernel at 3 / 2 / 4 wavefronts per SIMD, field-element compares deferred or immediate) against without ("noride")
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "riding or inorder_schedule or different_streams or detects_corruption" --durations=5 ) > $R/s20_tests.txt 2>&1; tail -6 $R/s20_tests.txt
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new noride; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new noride w2 w2i w4i; do pt $L 12:3; done
  pt new 8:3; pt noride 8:3; pt new 16:3
done
} 2>&1 | tee $R/ab_s20_rideg.txt
