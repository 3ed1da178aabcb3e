#!/bin/bash
# round 6: library builds (LIBS="label=path ..."; "new=" = the in-tree build) x service-loop points on ONE box, interleaved ROUNDS times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
for r in $(seq 1 ${ROUNDS:-2}); do
for LP in $LIBS; do
  L=${LP%%=*}; P=${LP#*=}
  if [ -n "$P" ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/$P; else unset POB_LIB_PATH; fi
  timeout 300 python tools/ab_loop.py --label $L --points "${POINTS:-4:0,8:0}" --steps ${STEPS:-96} --rounds 1 ${EXTRA} 2>&1 | grep -v "^$" | sed "s/^round 0/round $r/"
done; done 2>&1 | tee gpurun_out/ab_${TAG:-x}.txt
