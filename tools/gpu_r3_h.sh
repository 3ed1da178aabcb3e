#!/bin/bash
# round 3, pass h: A/B (interleaved, one box): evaluation kernel occupancy, streaming-stream priority; batch 2048 / 4096 per calculator; full GPU tests; R1CS referee on a GPU-emitted production witness
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
Q="--steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single"
one() { python -c "import json,sys; d=json.load(open('$1')); print(d['ms_per_step'], d['value'], 'kchk', d['roofline']['avg_ms'], 'alone', d['roofline']['avg_ms_alone'])" 2>&1 | tail -1; }
hipprio=$(python -c "print(0)")
for rep in 1 2; do
  for cfg in "2 x" "3 x" "4 x" "2 0" "2 -1" "4 0"; do
    set -- $cfg
    if [ "$2" = "x" ]; then P=""; else P="POB_KSTREAM_PRIO=$2"; fi
    env POB_KCHK_WAVES=$1 $P timeout 200 python bench.py $Q > $R/r3h_w$1_p$2_$rep.json 2> $R/r3h_w$1_p$2_$rep.err
    echo "waves=$1 kprio=$2 rep=$rep: $(one $R/r3h_w$1_p$2_$rep.json)"
  done
done
for B in 2048 4096; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 4 --no-cpu-baseline --no-emission --no-single --distinct-batches 2 > $R/r3h_batch$B.json 2> $R/r3h_batch$B.err
  echo "batch=$B: $(one $R/r3h_batch$B.json)"
done
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=1200 --durations=6 > $R/r3h_pytest.log 2>&1; tail -12 $R/r3h_pytest.log
timeout 600 python tools/gpu_r1cs_main_check.py /tmp > $R/r3h_r1cs_main_check_gpu.txt 2>&1; tail -4 $R/r3h_r1cs_main_check_gpu.txt
