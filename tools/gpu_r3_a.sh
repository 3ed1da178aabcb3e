#!/bin/bash
# round 3, first GPU pass: parity tests with the lane-spread Poseidon generation, the driver's bench line, per-unit times, kernel trace of a short bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout=800 > $R/r3a_pytest.log 2>&1; tail -3 $R/r3a_pytest.log
timeout 300 python bench.py --no-cpu-baseline > $R/r3a_bench.json 2> $R/r3a_bench.err; cut -c1-200 $R/r3a_bench.json
timeout 300 python tools/unit_times.py 1024 > $R/r3a_unit_times.txt 2>&1; grep -E "POS|BAH|POW|N2B|sum" $R/r3a_unit_times.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/r3a_prof -o r3a -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-emission > $R/r3a_prof.log 2>&1)
DB=$(ls $R/r3a_prof/*/*.db $R/r3a_prof/*.db 2>/dev/null | head -1)
python tools/rocpd_timeline.py $DB 6 0.1 > $R/r3a_timeline.txt 2>&1; tail -50 $R/r3a_timeline.txt
python tools/rocpd_summary.py $DB > $R/r3a_kernel_stats.txt 2>&1
