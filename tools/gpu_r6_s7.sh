#!/bin/bash
# round 6, GPU session 7: finer Poseidon segments / batched byte strings / folded asserts on hardware; the limb-per-lane product; C-ABI gather
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 -k "c_abi_record or spend_wtns_bit_exact or reference_suite or main_instantiation_batch or production_batch_payloads_beyond_group_0 or inorder_schedule_equals or gadget_mains_payload_and_evaluator or failure_sets or corruption_sweep or sm_sb_fr_pokes or two_ranks or max_depth_config5 or evaluator_detects" 2>&1 | tail -16) > $R/s7_tests.txt 2>&1
cat $R/s7_tests.txt
hipcc --offload-arch=gfx950 -O3 -w -I proof_of_burn_amd/csrc tools/ubench/fr_mul_lanes.hip -o /tmp/fr_mul_lanes && /tmp/fr_mul_lanes 2>&1 | tee $R/s7_fr_mul_lanes.txt
(cd /tmp && POB_PMC_INORDER=3 timeout 300 rocprofv3 --kernel-trace -d $R/s7_lone -o s -- python $GRAFT_REPO_ROOT/tools/pmc_one_batch.py > $R/s7_lone.log 2>&1)
python tools/lone_batch_trace.py $R/s7_lone/s_results.db > $R/s7_lone_batch.txt 2>&1; rm -rf $R/s7_lone; cat $R/s7_lone_batch.txt | cut -c12-100
timeout 300 python tools/unit_times.py 1024 > $R/s7_unit_times.txt 2>&1; awk '$3>0.04 || $2>0.04 || NR==1 || /F_|k_|all/' $R/s7_unit_times.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/s7_bench_driver.json 2> $R/s7_bench_driver.err
python -c "
import json
d = json.loads(open('gpurun_out/s7_bench_driver.json').read().strip().splitlines()[-1])
r = d['roofline']
print('driver cmd:', d['ms_per_step'], 'ms/step', d['value'], 'w/s; K_CHK in step', r['avg_ms'], 'frac', r['frac'], 'alone', r['frac_alone'], '; check_pass', r['check_pass']['ms'], r['check_pass']['frac'], 'tracks', r['check_pass']['track_schedule'], '; depth16', (d['depth16'] or {}).get('ms_per_step'), 'strong_slice', (d['strong_slice'] or {}).get('ms_per_step'), 'e2e', (d['e2e_from_json'] or {}).get('ms_per_step'), (d['e2e_from_json'] or {}).get('host_waited_for_loader_ms_per_step'), 'other', {k: v['ms_per_step'] for k, v in (d.get('other_depths') or {}).items()})
" 2>&1 | tail -3
