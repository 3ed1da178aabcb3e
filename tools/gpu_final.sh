#!/bin/bash
# final profile set of the committed state: the driver's bench command, the default run, kernel stats + timeline of the driver's command, per-unit times,
# PMC passes over the Keccak round kernels (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only), then the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out; T=${TAG:-r4z}
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/${T}_bench_driver.json 2> $R/${T}_bench_driver.err; cut -c1-220 $R/${T}_bench_driver.json; tail -2 $R/${T}_bench_driver.err
timeout 600 python bench.py > $R/${T}_bench_default.json 2> $R/${T}_bench_default.err; cut -c1-220 $R/${T}_bench_default.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/${T}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 24 --warmup 6 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_prof.log 2>&1)
python tools/rocpd_summary.py $R/${T}_prof/p_results.db > $R/${T}_kernel_stats.txt 2>&1; head -8 $R/${T}_kernel_stats.txt
python tools/rocpd_step.py $R/${T}_prof/p_results.db 40 0.0 > $R/${T}_step_timeline.txt 2>&1; tail -2 $R/${T}_step_timeline.txt
rm -rf $R/${T}_prof
timeout 300 python tools/unit_times.py 1024 > $R/${T}_unit_times.txt 2>&1; tail -14 $R/${T}_unit_times.txt
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/${T}_pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_pmc_fetch.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/${T}_pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_pmc_write.log 2>&1)
python tools/pmc_summary.py $R/${T}_pmc_fetch/f_results.db $R/${T}_pmc_write/w_results.db $R/${T}_pmc_k_rounds.json > $R/${T}_pmc.log 2>&1; tail -14 $R/${T}_pmc.log
python tools/pmc_all_kernels.py $R/${T}_pmc_fetch/f_results.db $R/${T}_pmc_write/w_results.db > $R/${T}_pmc_all_kernels.txt 2>&1
rm -rf $R/${T}_pmc_fetch $R/${T}_pmc_write
if [ -z "$NO_TESTS" ]; then timeout 2400 python -m pytest tests -m gpu -q --timeout=1500 > $R/${T}_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $R/${T}_gpu_tests.txt; tail -5 $R/${T}_gpu_tests.txt; fi
