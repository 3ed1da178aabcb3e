cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for r in 1 2; do
for L in base w3; do
  if [ $L = w3 ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_w3.so; else unset POB_LIB_PATH; fi
  for N in 4 8; do
    timeout 180 python bench.py --gpus 1 --steps 96 --warmup 12 --pipeline $N --no-cpu-baseline --no-emission --no-single --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('round $r $L N=$N', d['ms_per_step'], 'kchk', d['roofline']['avg_ms'], 'check_pass', d['roofline']['check_pass']['ms'])"
  done
done; done
