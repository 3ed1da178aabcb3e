#!/bin/bash
# round 3, final profile set of the committed state: the driver's bench command, kernel stats + step timeline of the same command, per-unit times,
# PMC passes over the Keccak round kernels (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only), emission window sweep
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r3z_bench_driver.json 2> $R/r3z_bench_driver.err; cut -c1-200 $R/r3z_bench_driver.json
timeout 600 python bench.py > $R/r3z_bench_default.json 2> $R/r3z_bench_default.err; cut -c1-200 $R/r3z_bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/r3z_prof -o r3z -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-emission --no-single > $R/r3z_prof.log 2>&1)
python tools/rocpd_summary.py $R/r3z_prof/r3z_results.db > $R/r3z_kernel_stats.txt 2>&1; head -12 $R/r3z_kernel_stats.txt
python tools/rocpd_step.py $R/r3z_prof/r3z_results.db 20 0.1 > $R/r3z_step_timeline.txt 2>&1; tail -3 $R/r3z_step_timeline.txt
timeout 300 python tools/unit_times.py 1024 > $R/r3z_unit_times.txt 2>&1; tail -14 $R/r3z_unit_times.txt
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/r3z_pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single > $R/r3z_pmc_fetch.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/r3z_pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single > $R/r3z_pmc_write.log 2>&1)
python tools/pmc_summary.py $R/r3z_pmc_fetch/f_results.db $R/r3z_pmc_write/w_results.db $R/r3z_pmc_k_rounds.json > $R/r3z_pmc.log 2>&1; tail -5 $R/r3z_pmc.log
python - > $R/r3z_emission_windows.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from proof_of_burn_amd import WitnessCalculator, inputs as gen
from proof_of_burn_amd.circuit_model import keepmap
MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
b = gen.synthetic_batch(8, depth=10, seed=0xB0B, distinct_keys=2)
c = WitnessCalculator(MAIN, max_batch=8)
assert all(r.ok for r in c.calculate(b.inputs, check=True))
keep, _ = keepmap.load(MAIN)
for ww in (4 << 20, 8 << 20, 16 << 20, 32 << 20):
    c.emit_throughput(0, 1, window_wires=ww)
    s, n = c.emit_throughput(1, 3, window_wires=ww)
    print(f"O0 payload, windows of {ww >> 20} Mi wires: {n / s / 1e9:.2f} GB/s, {s / 3 * 1e3:.1f} ms per witness")
for ww in (4 << 20, 8 << 20, 16 << 20, 24 << 20):
    c.emit_throughput(0, 1, window_wires=ww, keep=keep)
    s, n = c.emit_throughput(1, 4, window_wires=ww, keep=keep)
    print(f"reduced payload ({keep.size} wires), windows of {ww >> 20} Mi kept wires: {n / s / 1e9:.2f} GB/s, {s / 4 * 1e3:.2f} ms per witness")
c.close()
PY
cat $R/r3z_emission_windows.txt
