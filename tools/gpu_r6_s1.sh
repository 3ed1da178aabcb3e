#!/bin/bash
# round 6, GPU session 1: the transposition build on hardware (smoke + the payload comparisons that exercise every decomposition), then the A/B of builds x depths
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out; ulimit -c 0
R=gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spend_wtns_bit_exact or reference_suite_spend or main_instantiation_batch or production_batch_payloads_beyond_group_0 or inorder_schedule_equals or different_streams or gadget_mains_payload_and_evaluator" 2>&1 | tail -5) > $R/s1_tests.txt 2>&1
cat $R/s1_tests.txt
LIBS="r5=ab/libpob_r5.so new= nw3=ab/libpob_nw3.so nw4=ab/libpob_nw4.so" POINTS="4:0,8:0,12:0" ROUNDS=2 EXTRA="--alone" TAG=s1_libs bash tools/gpu_r6_ab.sh
LIBS="new=" POINTS="4:1,8:1,12:1,16:1,8:0" ROUNDS=2 TAG=s1_stream bash tools/gpu_r6_ab.sh
