#!/bin/bash
# round 3, pass d: the pipelined service loop after the stream-count fix (hardware-queue sweep), then the persistent-round-kernel A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
Q="--steps 40 --warmup 6 --no-cpu-baseline --no-emission --no-single"
one() { python -c "import json,sys; d=json.load(open('$1')); print(d['ms_per_step'], d['value'], 'kchk in step', d['roofline']['avg_ms'], 'alone', d['roofline']['avg_ms_alone'], 'gen', d['roofline']['gen_kernel']['avg_ms'], 'json/s', d['config']['json_to_packed_witnesses_per_s'])" 2>&1 | tail -1; }
for HWQ in 16 24 8; do
  GPU_MAX_HW_QUEUES=$HWQ timeout 200 python bench.py $Q > $R/r3d_hwq$HWQ.json 2> $R/r3d_hwq$HWQ.err
  echo "hwq=$HWQ: $(one $R/r3d_hwq$HWQ.json)"
done
OK=$(python -c "import json; print(1 if json.load(open('$R/r3d_hwq16.json'))['ms_per_step'] < 30 else 0)" 2>/dev/null || echo 0)
if [ "$OK" = "1" ]; then
  for rep in 1 2; do
    for cfg in "0 0" "2 0" "2 3" "2 4" "0 3" "1 3" "2 2"; do
      set -- $cfg
      POB_KR_PERSIST_CHECK=$1 POB_KR_PERSIST_GEN=$2 timeout 200 python bench.py $Q > $R/r3d_ab_c$1_g$2_$rep.json 2> $R/r3d_ab_c$1_g$2_$rep.err
      echo "persist check=$1 gen=$2 rep=$rep: $(one $R/r3d_ab_c$1_g$2_$rep.json)"
    done
  done
else
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $R/r3d_prof -o r3d -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-emission --no-single > $R/r3d_prof.log 2>&1)
  ls -la $R/r3d_prof* | head
fi
