#!/bin/bash
# round 6, GPU session 5: byte-form input kernel, LeafDetector / PublicCommitment copy units, the Poseidon + chain launch: parity, the lone-batch traces, loop A/B, the driver's command at three depths
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=12 -k "spend_wtns_bit_exact or reference_suite or main_instantiation_batch or production_batch_payloads_beyond_group_0 or inorder_schedule_equals or different_streams or gadget_mains_payload_and_evaluator or failure_sets or seeded_differential or max_depth_and_ragged or corruption_sweep or sm_sb_fr_pokes" 2>&1 | tail -22) > $R/s5_tests.txt 2>&1
cat $R/s5_tests.txt
for M in 3 1; do
  (cd /tmp && POB_PMC_INORDER=$M timeout 300 rocprofv3 --kernel-trace -d $R/s5_lone$M -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/s5_lone$M.log 2>&1)
  python tools/lone_batch_trace.py $R/s5_lone$M/s_results.db > $R/s5_lone_batch_mode$M.txt 2>&1; rm -rf $R/s5_lone$M; cat $R/s5_lone_batch_mode$M.txt | cut -c12-100
done
LIBS="new= r5=ab/libpob_r5.so" POINTS="4:0,8:0,12:0,4:1,8:1,12:1" ROUNDS=2 EXTRA="--alone" TAG=s5_loop bash tools/gpu_r6_ab.sh
for N in 8 12 16; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --pipeline $N --no-cpu-baseline > $R/s5_bench_driver_p$N.json 2> $R/s5_bench_driver_p$N.err
  python -c "
import json
d = json.loads(open('gpurun_out/s5_bench_driver_p$N.json').read().strip().splitlines()[-1])
r = d['roofline']
print('driver cmd, pipeline $N:', d['ms_per_step'], 'ms/step', d['value'], 'w/s; K_CHK in step', r['avg_ms'], 'frac', r['frac'], 'alone', r['frac_alone'], '; check_pass', r['check_pass']['ms'], '; depth16', (d['depth16'] or {}).get('ms_per_step'), 'strong_slice', (d['strong_slice'] or {}).get('ms_per_step'), 'e2e', (d['e2e_from_json'] or {}).get('ms_per_step'), 'other', {k: v['ms_per_step'] for k, v in (d.get('other_depths') or {}).items()})
" 2>&1 | tail -3
done 2>&1 | tee $R/s5_driver_cmd.txt
