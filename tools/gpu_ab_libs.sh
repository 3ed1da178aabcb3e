#!/bin/bash
# A/B of several BUILDS of the library on one box: LIBS="label=path ..." (a path relative to the repo; "new=" = the in-tree library), ROUNDS interleaved rounds of the
# 96-step loop with the pipeline depths in DEPTHS (default "4 8")
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for r in $(seq 1 ${ROUNDS:-2}); do
for LP in $LIBS; do
  L=${LP%%=*}; P=${LP#*=}
  if [ -n "$P" ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/$P; else unset POB_LIB_PATH; fi
  for N in ${DEPTHS:-4 8}; do
    timeout 180 python bench.py --gpus 1 --steps ${STEPS:-96} --warmup 12 --pipeline $N --no-cpu-baseline --no-emission --no-single --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('round $r $L N=$N', d['ms_per_step'], 'ms/step; round evaluation in the step', r['avg_ms'], 'ms = frac', r['frac'], '; whole evaluation alone', r['check_pass']['ms'], 'ms')"
  done
done; done 2>&1 | tee gpurun_out/ab_libs_${TAG:-x}.txt
