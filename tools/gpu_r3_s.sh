#!/bin/bash
# round 3, call 27: derived operand wires, this build against the one before -- tests, unit times, interleaved pairs against the build before
# (two-build A/B: needs libpob_hip_base.so / libpob_hip_<variant>.so copied next to libpob_hip.so and, in witness.py for the run only,
#  LIB_PATH = os.environ.get("POB_LIB_EXPERIMENT") or ...; the product reads no such variable)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "spend_wtns or proof_of_burn_and_wtns or main_instantiation_batch or gadget_mains_payload or failure_sets or corruption_sweep" 2>&1 | tail -4
for v in base dv; do
  POB_LIB_EXPERIMENT=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_$v.so timeout 300 python tools/unit_times.py 1024 > $R/r3t_units_$v.txt 2>&1
  echo "== $v"; grep -E "U_KB_RANGE|U_LD_SELR|U_POB_LASTLAYER_RANGE|U_RL_SLROW|U_SC_RANGE|^sum|all G check|F_LD|F_RANGE|F_RL |F_SC" $R/r3t_units_$v.txt
done
for rep in 1 2 3 4 5; do for v in base dv; do
  POB_LIB_EXPERIMENT=$GRAFT_REPO_ROOT/proof_of_burn_amd/csrc/libpob_hip_$v.so timeout 300 python bench.py --gpus 1 --steps 100 --warmup 6 --no-cpu-baseline --no-emission --no-single > $R/r3t_${v}_$rep.json 2> $R/r3t_${v}_$rep.err
  python - <<PY
import json
d=json.loads(open("$R/r3t_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v rep=$rep", d["ms_per_step"], d["value"], "kchk", d["roofline"]["avg_ms"])
PY
done; done 2>&1 | tee $R/r3t_summary.txt
