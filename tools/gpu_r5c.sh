#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-emission --no-single --no-extra-legs --x-kchk-sweep "$1" > $R/r5c_sweep.json 2> $R/r5c_sweep.err; grep SWEEP $R/r5c_sweep.err | cut -c7-250; tail -3 $R/r5c_sweep.err | cut -c1-300
