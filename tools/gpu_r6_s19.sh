#!/bin/bash
# round 6, GPU session 19: KNOCK-OUT -- what the loop would gain if the rest of the evaluation (the G families; the sponge chains) cost nothing: builds that skip those launches
# (their records are not evaluated: no measurement of the product, an upper bound for "all of the evaluation rides with the generation"); and smoke()
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $R/s19_smoke.txt
pt() { L=$1; if [ "$L" != new ]; then export POB_LIB_PATH=$GRAFT_REPO_ROOT/ab/libpob_$L.so; else unset POB_LIB_PATH; fi
       timeout 200 python tools/ab_loop.py --label $L --points "$2" --steps ${3:-96} --rounds 1 $4 2>&1 | grep "^round\|Error\|error" | head -5; }
{
for r in 1 2 3; do
  for L in new nogeval noeval; do pt $L 12:3; done
done
} 2>&1 | tee $R/ab_s19_knockout.txt
