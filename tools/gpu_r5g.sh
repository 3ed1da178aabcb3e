#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
ROUNDS=1 STEPS=96 bash tools/gpu_ab.sh "base4: --pipeline 4 --dbg-no-fetch" "base8: --pipeline 8 --dbg-no-fetch"
for L in 1 4 8 16 512 20 2048; do POB_X_SKIP_LEVELS=$L ROUNDS=1 STEPS=96 bash tools/gpu_ab.sh "skipL${L}_4: --pipeline 4 --dbg-no-fetch" "skipL${L}_8: --pipeline 8 --dbg-no-fetch"; done
for S in 256 512 768; do POB_X_SKIP=$S ROUNDS=1 STEPS=96 bash tools/gpu_ab.sh "skip${S}_4: --pipeline 4 --dbg-no-fetch" "skip${S}_8: --pipeline 8 --dbg-no-fetch"; done
