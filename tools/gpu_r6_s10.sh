#!/bin/bash
# round 6, GPU session 10: serial per-wire loops unrolled by 4 (variant build) -- unit times and the loop; reduced-emission rate, round 5's build against this one
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
for LP in new= unr4=ab/libpob_unr4.so; do
  L=${LP%%=*}; P=${LP#*=}
  if [ -n "$P" ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/$P; else unset POB_LIB_PATH; fi
  echo "== $L"; timeout 300 python tools/unit_times.py 1024 2>&1 | awk '$3>0.035 || $2>0.05 || NR==1 || /F_|all/'
  timeout 200 python tools/ab_loop.py --label $L --points "12:1" --steps 96 --rounds 1 --alone 2>&1 | grep "^round"
  timeout 200 python tools/ab_loop.py --label $L --points "4:1" --steps 96 --rounds 1 2>&1 | grep "^round"
done 2>&1 | tee $R/s10_unroll.txt
for LP in r5=ab/libpob_r5.so new= r5=ab/libpob_r5.so new=; do
  L=${LP%%=*}; P=${LP#*=}
  if [ -n "$P" ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/$P; else unset POB_LIB_PATH; fi
  timeout 300 python tools/emit_rate.py $L 2>&1 | tail -1
done 2>&1 | tee $R/s10_emit.txt
