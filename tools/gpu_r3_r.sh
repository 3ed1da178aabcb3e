#!/bin/bash
# round 3, call 25: the SB class (operands of the Keccak output selectors' IsEqual gadgets) as DERIVED wires (not stored) -- unit times and interleaved pairs
# (two-build A/B: needs libpob_hip_base.so / libpob_hip_<variant>.so copied next to libpob_hip.so and, in witness.py for the run only,
#  LIB_PATH = os.environ.get("POB_LIB_EXPERIMENT") or ...; the product reads no such variable)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_base.so; D=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_sbd.so
for v in base sbd; do
  POB_LIB_EXPERIMENT=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_$v.so timeout 300 python tools/unit_times.py 1024 > $R/r3r_units_$v.txt 2>&1
  echo "== $v"; grep -E "U_KB_SELROW|F_SELROW|^sum|all G check" $R/r3r_units_$v.txt
done
for rep in 1 2 3 4 5; do for v in base sbd; do
  POB_LIB_EXPERIMENT=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_$v.so timeout 300 python bench.py --gpus 1 --steps 100 --warmup 6 --no-cpu-baseline --no-emission --no-single > $R/r3r_${v}_$rep.json 2> $R/r3r_${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("$R/r3r_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v rep=$rep", d["ms_per_step"], d["value"], "kchk", d["roofline"]["avg_ms"])
PY
done; done 2>&1 | tee $R/r3r_summary.txt
