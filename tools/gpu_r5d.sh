#!/bin/bash
# kernel trace + step timeline of the service loop with $1 calculators in flight
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out; T=${TAG:-r5d}; N=${1:-4}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/${T}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 24 --warmup 8 --pipeline $N --no-cpu-baseline --no-emission --no-single --no-extra-legs > $R/${T}_prof.log 2>&1)
python tools/rocpd_summary.py $R/${T}_prof/p_results.db > $R/${T}_kernel_stats_$N.txt 2>&1; head -24 $R/${T}_kernel_stats_$N.txt | cut -c1-150
python tools/rocpd_step.py $R/${T}_prof/p_results.db 30 0.0 > $R/${T}_step_timeline_$N.txt 2>&1; tail -2 $R/${T}_step_timeline_$N.txt
rm -rf $R/${T}_prof
