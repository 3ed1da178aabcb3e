#!/bin/bash
# round 6, GPU session 9: loader pool threads at nice 10 (the GPU-driving thread first) against equal priority, 12 in flight, 96 steps so that the CPU quota is in force
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label env...
  L=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 96 --warmup 5 --no-cpu-baseline --no-emission --no-extra-legs > $R/s9_$L.json 2> $R/s9_$L.err
  python -c "
import json
d = json.loads(open('gpurun_out/s9_$L.json').read().strip().splitlines()[-1]); e = d['e2e_from_json']
print('$L: packed ahead', d['ms_per_step'], 'ms/step; e2e', e['ms_per_step'], 'ms, loader', e['loader_ms_per_batch'], 'ms per batch, host waited', e['host_waited_for_loader_ms_per_step'], 'ms;', e['bound'][:40])
" 2>&1 | tail -1
}
for r in 1 2 3; do
  run nice_$r X=1; run flat_$r POB_X_NO_NICE=1
done 2>&1 | tee $R/s9_loader.txt
