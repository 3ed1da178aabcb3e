#!/bin/bash
# round 6, GPU session 3: how much of the step is the SIZE of a launch?  the same loop with 4 096 / 8 192 witnesses per calculator pass
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
(timeout 400 python tools/ab_loop.py --label b4096 --batch 4096 --nbatches 2 --points "1:0,2:0,3:0,4:0" --steps 24 --rounds 1 --alone 2>&1 | grep "^round";
 timeout 500 python tools/ab_loop.py --label b8192 --batch 8192 --nbatches 2 --points "1:0,2:0,3:0" --steps 12 --rounds 1 --alone 2>&1 | grep "^round";
 timeout 300 python tools/ab_loop.py --label b2048 --batch 2048 --nbatches 2 --points "2:0,4:0,6:0" --steps 48 --rounds 1 2>&1 | grep "^round") | tee gpurun_out/ab_s3_sizes.txt
