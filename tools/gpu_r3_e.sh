#!/bin/bash
# round 3, pass e: what the service loop costs over the bare kernel pipeline -- interleaved variants on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
Q="--steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single"
one() { python -c "import json,sys; d=json.load(open('$1')); print(d['ms_per_step'], d['value'], 'kchk', d['roofline']['avg_ms'], d['roofline']['measured'][:12])" 2>&1 | tail -1; }
for rep in 1 2; do
  i=0
  for V in "" "--probe-in-timed-region" "--dbg-no-upload" "--dbg-no-fetch" "--dbg-no-upload --dbg-no-fetch"; do
    i=$((i+1))
    timeout 200 python bench.py $Q $V > $R/r3e_v${i}_$rep.json 2> $R/r3e_v${i}_$rep.err
    echo "[$V] rep=$rep: $(one $R/r3e_v${i}_$rep.json)"
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/r3e_prof -o r3e -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-emission --no-single > $R/r3e_prof.log 2>&1)
DB=$(ls $R/r3e_prof/*.db $R/r3e_prof/*/*.db 2>/dev/null | head -1)
python tools/rocpd_timeline.py $DB 8 0.1 > $R/r3e_timeline.txt 2>&1; tail -60 $R/r3e_timeline.txt
