#!/bin/bash
# round 3, pass i: SM class as int16 rows -- bench x2 first, then every GPU test
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
Q="--steps 60 --warmup 6 --no-cpu-baseline --no-emission"
one() { python -c "import json,sys; d=json.load(open('$1')); print(d['ms_per_step'], d['value'], 'kchk', d['roofline']['avg_ms'], 'alone', d['roofline']['avg_ms_alone'], 'pass', d['roofline']['check_pass']['ms'], 'single', d['single_calculator']['ms_per_step'], 'bytes/witness', d['config']['resident_bytes_per_witness'])" 2>&1 | tail -1; }
for rep in 1 2; do
  timeout 200 python bench.py $Q > $R/r3i_bench_$rep.json 2> $R/r3i_bench_$rep.err; echo "rep=$rep: $(one $R/r3i_bench_$rep.json)"; tail -2 $R/r3i_bench_$rep.err
done
timeout 300 python tools/unit_times.py 1024 > $R/r3i_unit_times.txt 2>&1; tail -14 $R/r3i_unit_times.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=1200 > $R/r3i_pytest.log 2>&1; tail -5 $R/r3i_pytest.log
