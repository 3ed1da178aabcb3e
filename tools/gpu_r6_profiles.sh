#!/bin/bash
# round 6's profile set of the committed state (everything lands in gpurun_out/${TAG}_*, copy what is to be kept into profiles/):
#   the driver's bench command | kernel stats + memory copies + step timeline of the service loop | every launch of ONE batch on a lone calculator |
#   per-unit-kind times | SQ counters of one batch (two passes of four counters) | FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out; T=${TAG:-round6}
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/${T}_bench_driver_cmd.json 2> $R/${T}_bench_driver.err; cut -c1-200 $R/${T}_bench_driver_cmd.json; tail -2 $R/${T}_bench_driver.err
if [ -n "$DEFAULT_RUN" ]; then timeout 900 python bench.py --no-cpu-baseline > $R/${T}_bench_default.json 2> $R/${T}_bench_default.err; cut -c1-200 $R/${T}_bench_default.json; fi
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/${T}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 24 --warmup 8 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_prof.log 2>&1)
python tools/rocpd_summary.py $R/${T}_prof/p_results.db > $R/${T}_kernel_stats.txt 2>&1; head -8 $R/${T}_kernel_stats.txt | cut -c1-150
python tools/rocpd_step.py $R/${T}_prof/p_results.db 30 0.0 > $R/${T}_step_timeline.txt 2>&1; tail -1 $R/${T}_step_timeline.txt
rm -rf $R/${T}_prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/${T}_lone -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/${T}_lone.log 2>&1)
python tools/lone_batch_trace.py $R/${T}_lone/s_results.db > $R/${T}_lone_batch.txt 2>&1; rm -rf $R/${T}_lone; tail -1 $R/${T}_lone_batch.txt
timeout 300 python tools/unit_times.py 1024 > $R/${T}_unit_times.txt 2>&1; tail -5 $R/${T}_unit_times.txt
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY --kernel-trace -d $R/${T}_sq1 -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/${T}_sq1.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/${T}_sq2 -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/${T}_sq2.log 2>&1)
{ python tools/pmc_units.py $R/${T}_sq1/s_results.db; echo; python tools/pmc_units.py $R/${T}_sq2/s_results.db; } > $R/${T}_sq_counters.txt 2>&1; rm -rf $R/${T}_sq1 $R/${T}_sq2; head -6 $R/${T}_sq_counters.txt | cut -c1-150
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/${T}_pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_pmc_fetch.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/${T}_pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_pmc_write.log 2>&1)
python tools/pmc_summary.py $R/${T}_pmc_fetch/f_results.db $R/${T}_pmc_write/w_results.db $R/${T}_pmc_k_rounds.json > $R/${T}_pmc.log 2>&1; grep traffic_over $R/${T}_pmc.log
python tools/pmc_all_kernels.py $R/${T}_pmc_fetch/f_results.db $R/${T}_pmc_write/w_results.db > $R/${T}_pmc_all_kernels.txt 2>&1
rm -rf $R/${T}_pmc_fetch $R/${T}_pmc_write
if [ -n "$TESTS" ]; then python -c "import sys; sys.path.insert(0, 'tests'); import oracle_ffi; oracle_ffi.lib()"
  # the suite the way the driver runs it: one process, serial
  ( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 ) > $R/${T}_gpu_tests_serial.txt 2>&1; echo "pytest rc=$?" >> $R/${T}_gpu_tests_serial.txt; tail -25 $R/${T}_gpu_tests_serial.txt; fi
