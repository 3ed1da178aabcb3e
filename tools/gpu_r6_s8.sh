#!/bin/bash
# round 6, GPU session 8: the loop from input.json TEXT at 12 in flight: loader width against the 16-CPU quota, blocking event waits of the host
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
run() { # label env...
  L=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-emission --no-extra-legs > $R/s8_$L.json 2> $R/s8_$L.err
  python -c "
import json
d = json.loads(open('gpurun_out/s8_$L.json').read().strip().splitlines()[-1]); e = d['e2e_from_json']
print('$L: packed ahead', d['ms_per_step'], 'ms/step; e2e', e['ms_per_step'], 'ms, loader', e['loader_ms_per_batch'], 'ms per batch, host waited', e['host_waited_for_loader_ms_per_step'], 'ms;', e['bound'][:40])
" 2>&1 | tail -1
}
for r in 1 2; do
  run base_$r X=1; run t14_$r POB_LOADER_THREADS=14; run t12_$r POB_LOADER_THREADS=12; run blk_$r POB_X_BLOCKING=1; run blk14_$r POB_X_BLOCKING=1 POB_LOADER_THREADS=14; run t20_$r POB_LOADER_THREADS=20
done 2>&1 | tee $R/s8_loader.txt
