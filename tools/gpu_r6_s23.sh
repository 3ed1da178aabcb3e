#!/bin/bash
# round 6, GPU session 23: the Poseidon blocks' riding evaluation with a pending compare per store site (3 / 4 slots) against ONE (block 21's form, "slots1"): the launch alone, the loop
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "riding or inorder_schedule" --durations=3 ) > $R/s23_tests.txt 2>&1; tail -4 $R/s23_tests.txt
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new slots1 slots4; do pt $L 4:3 48 --alone; done
for r in 1 2 3; do
  for L in new slots1 slots4; do pt $L 12:3; pt $L 8:3; done
done
} 2>&1 | tee $R/ab_s23_slots.txt
