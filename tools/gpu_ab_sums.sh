#!/bin/bash
# planner switch POB_SC_SUMS_SPLIT (one SubstringCheck sums unit per 64 positions / per layer), interleaved on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
for rep in 1 2 3; do for S in 0 1; do
  POB_SC_SUMS_SPLIT=$S timeout 150 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-emission > $R/ab_sums$S.json 2> $R/ab_sums$S.err
  echo "sums_split=$S $(cut -c47-60 $R/ab_sums$S.json) $(grep -v amdgpu.ids $R/ab_sums$S.err | tail -1 | cut -c1-100)"
done; done
