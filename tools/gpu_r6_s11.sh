#!/bin/bash
# round 6, GPU session 11: the device PARTITIONED between the calculators (every calculator's stream restricted to a share of the compute units, hipExtStreamCreateWithCUMask),
# one process per point, 12 and 8 in flight; and the loader with block-wise validation in the loop from JSON text
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
pt() { timeout 200 python tools/ab_loop.py --label new --points "$1" --steps 96 --rounds 1 --cumask $2 2>&1 | grep "^round"; }
for r in 1 2; do
  pt 12:1 none; pt 12:1 parity; pt 12:1 quarters; pt 12:1 halves; pt 8:1 none; pt 8:1 parity; pt 4:1 none; pt 4:1 quarters
done 2>&1 | tee $R/ab_s11_cumask.txt
for r in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 96 --warmup 5 --no-cpu-baseline --no-emission --no-extra-legs > $R/s11_e2e_$r.json 2> $R/s11_e2e_$r.err
  python -c "
import json
d = json.loads(open('gpurun_out/s11_e2e_$r.json').read().strip().splitlines()[-1]); e = d['e2e_from_json']
print('run $r: packed ahead', d['ms_per_step'], 'ms/step; e2e', e['ms_per_step'], 'ms, loader', e['loader_ms_per_batch'], 'ms per batch, host waited', e['host_waited_for_loader_ms_per_step'], 'ms;', e['bound'][:40], '; json->packed', d['config']['json_to_packed_witnesses_per_s'])
" 2>&1 | tail -1
done 2>&1 | tee $R/s11_loader.txt
timeout 200 python tools/loader_rate.py 1024 1,8,16,32 2>&1 | tail -5 | tee $R/s11_loader_rate.txt
