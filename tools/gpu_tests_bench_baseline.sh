#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(free -g; nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > gpurun_out/r2d_box.txt 2>&1; head -3 gpurun_out/r2d_box.txt
timeout 1700 python -m pytest tests -m gpu -q -x --timeout=1500 > gpurun_out/r2d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log; tail -3 gpurun_out/r2d_pytest.log
free -g | head -2
for H in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 2 --halves $H --no-cpu-baseline > gpurun_out/r2d_bench_h$H.json 2> gpurun_out/r2d_bench_h$H.err
  python -c "
import json; d=json.load(open('gpurun_out/r2d_bench_h$H.json')); r=d['roofline']; print('h$H', d['value'], d['ms_per_step'], 'check_pass', r['check_pass']['ms'], 'emission', d['emission'])"
done
timeout 600 python bench.py > gpurun_out/r2d_bench_default.json 2> gpurun_out/r2d_bench_default.err; cut -c1-300 gpurun_out/r2d_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r2d_bench_default.json')); print(d['cpu_baseline'])"
