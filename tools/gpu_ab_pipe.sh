#!/bin/bash
# single calculator against the two-calculator pipeline (bench.py --pipeline 0 / 1; POB_PIPE_CHAINS=0: narrow chains gated too), interleaved on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2 3; do for M in "0 1" "1 0" "1 1"; do set -- $M
  POB_PIPE_CHAINS=$2 timeout 150 python bench.py --pipeline $1 --steps 20 --warmup 4 --no-cpu-baseline --no-emission > $R/ab_p$1_c$2.json 2> $R/ab_p$1_c$2.err
  echo "pipeline=$1 chains_early=$2 $(cut -c47-60 $R/ab_p$1_c$2.json) $(grep -v amdgpu.ids $R/ab_p$1_c$2.err | tail -1 | cut -c1-100)"
done; done
