#!/bin/bash
# round 4: the new parity tests, the driver's bench command on the rebuilt bench.py, PMC passes over the compact Keccak round kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=1200 -k "beyond_group or across_depths or production_gpu_witness or field_inversions or stall or rccl or queued" > $R/r4f_pytest.log 2>&1; echo "pytest rc=$?" >> $R/r4f_pytest.log; tail -5 $R/r4f_pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r4f_bench_driver.json 2> $R/r4f_bench_driver.err; cut -c1-300 $R/r4f_bench_driver.json; tail -3 $R/r4f_bench_driver.err
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/r4f_pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/r4f_pmc_fetch.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/r4f_pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/r4f_pmc_write.log 2>&1)
python tools/pmc_summary.py $R/r4f_pmc_fetch/f_results.db $R/r4f_pmc_write/w_results.db $R/r4f_pmc_k_rounds.json > $R/r4f_pmc.log 2>&1; tail -8 $R/r4f_pmc.log
python tools/pmc_all_kernels.py $R/r4f_pmc_fetch/f_results.db $R/r4f_pmc_write/w_results.db > $R/r4f_pmc_all_kernels.txt 2>&1; head -40 $R/r4f_pmc_all_kernels.txt
rm -rf $R/r4f_pmc_fetch $R/r4f_pmc_write
