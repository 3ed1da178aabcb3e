#!/bin/bash
# round 3: this build (libpob_hip_dv.so) against the one before (libpob_hip_base.so): five interleaved pairs, then the unit times of the new one
# (two-build A/B: needs libpob_hip_base.so / libpob_hip_<variant>.so copied next to libpob_hip.so and, in witness.py for the run only,
#  LIB_PATH = os.environ.get("POB_LIB_EXPERIMENT") or ...; the product reads no such variable)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-r3u}
for rep in 1 2 3 4 5; do for v in base dv; do
  POB_LIB_EXPERIMENT=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_$v.so timeout 300 python bench.py --gpus 1 --steps 100 --warmup 6 --no-cpu-baseline --no-emission --no-single > $R/${T}_${v}_$rep.json 2> $R/${T}_${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("$R/${T}_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v rep=$rep", d["ms_per_step"], d["value"], "kchk", d["roofline"]["avg_ms"])
PY
done; done 2>&1 | tee $R/${T}_summary.txt
POB_LIB_EXPERIMENT=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_dv.so timeout 300 python tools/unit_times.py 1024 > $R/${T}_units_dv.txt 2>&1
grep -E "^sum|all G check" $R/${T}_units_dv.txt
