#!/bin/bash
# short verification of the committed state when GPU minutes are scarce: the driver's bench command, the whole -m gpu suite spread over ${NPROC:-4}
# pytest-xdist workers (the one timing test runs alone afterwards), then per-unit times, kernel stats + timeline of the driver's command, a batch sweep.
# (tools/gpu_final.sh is the full profile set with the PMC passes and the suite in one process, as the driver runs it.)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out; T=${TAG:-r4v}
python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"      # build the checker once, before the workers race for it
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/${T}_bench_driver.json 2> $R/${T}_bench_driver.err; cut -c1-220 $R/${T}_bench_driver.json; tail -2 $R/${T}_bench_driver.err
STALL=tests/test_gpu_parity.py::test_reading_a_batch_does_not_stall_the_partner
timeout 1500 python -m pytest tests -m gpu -q -n ${NPROC:-4} --deselect $STALL --timeout=1200 --durations=12 > $R/${T}_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $R/${T}_gpu_tests.txt; tail -20 $R/${T}_gpu_tests.txt
timeout 600 python -m pytest $STALL -q --timeout=500 > $R/${T}_gpu_tests_stall.txt 2>&1; echo "pytest rc=$?" >> $R/${T}_gpu_tests_stall.txt; tail -3 $R/${T}_gpu_tests_stall.txt
timeout 300 python tools/unit_times.py 1024 > $R/${T}_unit_times.txt 2>&1; tail -14 $R/${T}_unit_times.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/${T}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 24 --warmup 6 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_prof.log 2>&1)
python tools/rocpd_summary.py $R/${T}_prof/p_results.db > $R/${T}_kernel_stats.txt 2>&1; head -8 $R/${T}_kernel_stats.txt
python tools/rocpd_step.py $R/${T}_prof/p_results.db 40 0.0 > $R/${T}_step_timeline.txt 2>&1; tail -2 $R/${T}_step_timeline.txt
rm -rf $R/${T}_prof
# batch sweep: ms/step against the batch size = the per-step constant and the per-witness slope (4 in-order calculators in flight; 1 024 is the driver's command above)
: > $R/${T}_batch_sweep.txt
for B in 2048 4096; do timeout 300 python bench.py --steps 40 --warmup 6 --batch $B --no-cpu-baseline --no-emission --no-single --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('4 in flight, batch %5d  %.3f ms/step  %.0f witnesses/s' % ($B, d['ms_per_step'], d['value']))" >> $R/${T}_batch_sweep.txt; done
cat $R/${T}_batch_sweep.txt
