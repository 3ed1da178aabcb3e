cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
for r in 1 2 3; do for d in 6 8 10 12 16; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --pipeline $d --no-cpu-baseline --no-emission --no-extra-legs --no-single 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d steps 20:', d['ms_per_step'], 'ms/step', d['value'])"
done; done 2>&1 | tee gpurun_out/s28_depth20.txt
