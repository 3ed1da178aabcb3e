#!/bin/bash
# single calculator against the two-calculator pipeline (bench.py --pipeline 0 / 1), interleaved on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2 3; do for P in 0 1; do
  timeout 150 python bench.py --pipeline $P --steps 20 --warmup 4 --no-cpu-baseline --no-emission > $R/ab_pipe$P.json 2> $R/ab_pipe$P.err
  echo "pipeline=$P $(cut -c47-60 $R/ab_pipe$P.json)"
done; done
