#!/bin/bash
# final round-2 profile set
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python bench.py > $R/r2m_bench_default.json 2> $R/r2m_bench_default.err; cut -c1-160 $R/r2m_bench_default.json
timeout 300 python bench.py --no-cpu-baseline --no-emission > $R/r2m_bench_b.json 2> $R/r2m_bench_b.err; cut -c1-160 $R/r2m_bench_b.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/r2m_prof -o r2m -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-emission > $R/r2m_prof.log 2>&1)
timeout 300 python tools/unit_times.py 1024 > $R/r2m_unit_times.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
