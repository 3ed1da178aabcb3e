#!/bin/bash
# round 3, pass j: batched Selector evaluation -- unit times, SQ counters of every unit kind alone, bench x2
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/unit_times.py 1024 > $R/r3j_unit_times.txt 2>&1; grep -E "LD_SELR|LASTLAYER_RANGE|LASTLEN|SC_RANGE|KB_SELROW|F_|sum|all G" $R/r3j_unit_times.txt
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace -d $R/r3j_pmc -o u -- python $GRAFT_REPO_ROOT/tools/unit_times.py 1024 > $R/r3j_pmc.log 2>&1)
python tools/pmc_units.py $R/r3j_pmc/u_results.db > $R/r3j_pmc_units.txt 2>&1; head -40 $R/r3j_pmc_units.txt
Q="--steps 60 --warmup 6 --no-cpu-baseline --no-emission --no-single"
for rep in 1 2; do
  timeout 200 python bench.py $Q > $R/r3j_bench_$rep.json 2> $R/r3j_bench_$rep.err; python -c "import json; d=json.load(open('$R/r3j_bench_$rep.json')); print(d['ms_per_step'], d['value'], 'kchk', d['roofline']['avg_ms'], 'pass', d['roofline']['check_pass']['ms'])"
done
