#!/bin/bash
# round 3, final: the profile set (tools/gpu_r3_final.sh) of the committed state, then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_r3_final.sh
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r3z_gpu_tests.txt
