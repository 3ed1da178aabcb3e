#!/bin/bash
# round 6, GPU session 15: the evaluation of the input rows and of the sponge chains riding with their generation too (k_inputs / k_chain MODE 2), the Poseidon workgroups of four
# wavefronts sharing one LDS copy of the constants; first the GPU tests of the riding evaluation
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "riding or inorder_schedule or different_streams or detects_corruption" --durations=5 ) > $R/s15_tests.txt 2>&1; tail -12 $R/s15_tests.txt
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new old; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new old posw1 rounds nochain; do pt $L 12:3; done
  pt new 16:3; pt nochain 16:3; pt new 8:3; pt old 8:3
done
} 2>&1 | tee $R/ab_s15_ride.txt
