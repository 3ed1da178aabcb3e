#!/bin/bash
# round 6, GPU session 6: clean A/B, ONE process per point (later points of a process run 6-10 % slower whatever they are): fused launch on / off, non-temporal expansion stores
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
pt() { # label lib point
  if [ -n "$2" ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/$2; else unset POB_LIB_PATH; fi
  timeout 200 python tools/ab_loop.py --label $1 --points "$3" --steps 96 --rounds 1 2>&1 | grep "^round"
}
for r in 1 2; do
  pt new "" 8:1; pt new "" 8:0; pt nt ab/libpob_kgennt.so 8:1; pt new "" 12:0; pt new "" 12:1; pt nt ab/libpob_kgennt.so 12:1; pt new "" 4:1; pt new "" 4:0; pt new "" 6:1; pt new "" 10:1
done 2>&1 | tee gpurun_out/ab_s6_clean.txt
