#!/bin/bash
# round 6, GPU session 12: the Keccak round blocks evaluated by the launch that writes them (k_rounds_gc, pob_set_inorder bit 2): the kernel alone against expansion + evaluation
# as two launches, builds with other rounds per wavefront / wavefronts per SIMD / loads ahead, and the service loop with and without it (fused 1 = Poseidon + chain launch only, 3 = + gc)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for L in new w3 kr8 kr2w3 dp0w3 dp24; do pt $L 4:3 48 --alone; done
pt new 4:1 48 --alone
for r in 1 2; do
  for p in 12:1 12:3 8:1 8:3 4:1 4:3 16:3; do pt new $p; done
  pt w3 12:3; pt kr8 12:3; pt w3 8:3
done
} 2>&1 | tee $R/ab_s12_gc.txt
