#!/bin/bash
# A/B of two BUILDS of the library on one box: proof_of_burn_amd/csrc/libpob_hip.so (label "new") against the build at $BASE_LIB (label "base", loaded through POB_LIB_PATH),
# ROUNDS interleaved rounds of the 96-step loop with 4 and 8 calculators in flight
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
BASE_LIB=${BASE_LIB:-$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_base.so}
for r in $(seq 1 ${ROUNDS:-2}); do
for L in base new; do
  if [ $L = base ]; then export POB_LIB_PATH=$BASE_LIB; else unset POB_LIB_PATH; fi
  for N in 4 8; do
    timeout 180 python bench.py --gpus 1 --steps 96 --warmup 12 --pipeline $N --no-cpu-baseline --no-emission --no-single --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('round $r $L N=$N', d['ms_per_step'], 'ms/step; round evaluation in the step', d['roofline']['avg_ms'], 'ms; whole evaluation alone', d['roofline']['check_pass']['ms'], 'ms')"
  done
done; done
