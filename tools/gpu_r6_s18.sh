#!/bin/bash
# round 6, GPU session 18: the loader pool's width in the loop from JSON text now that the device needs 1.02-1.06 ms per batch (the box shows 256 CPUs and grants 16 by cgroup quota):
# POB_LOADER_THREADS 16 (default) / 20 / 24 / 32, 12 in flight, 96 steps, three interleaved rounds; the pool alone first
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 200 python tools/loader_rate.py 1024 1,8,12,16,20,24,32,48 2>&1 | tail -10 | tee $R/s18_loader_rate.txt
run() { # label env...
  L=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 96 --warmup 5 --no-cpu-baseline --no-emission --no-extra-legs > $R/s18_$L.json 2> $R/s18_$L.err
  python -c "
import json
d = json.loads(open('gpurun_out/s18_$L.json').read().strip().splitlines()[-1]); e = d['e2e_from_json']
print('$L: packed ahead', d['ms_per_step'], 'ms/step; e2e', e['ms_per_step'], 'ms, loader', e['loader_ms_per_batch'], 'ms per batch, host waited', e['host_waited_for_loader_ms_per_step'], 'ms;', e['bound'][:40])
" 2>&1 | tail -1
}
for r in 1 2 3; do
  run t16_$r X=1; run t20_$r POB_LOADER_THREADS=20; run t24_$r POB_LOADER_THREADS=24; run t32_$r POB_LOADER_THREADS=32
done 2>&1 | tee $R/s18_loader.txt
