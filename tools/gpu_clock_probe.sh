#!/bin/bash
# does the shader clock drop while the HBM-streaming kernels run?  sample sclk / power from sysfs beside a long bench run
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
R=$GRAFT_REPO_ROOT/gpurun_out
ls /sys/class/drm/ > $R/clk_sysfs.txt 2>&1
for d in /sys/class/drm/card*/device; do echo "== $d" >> $R/clk_sysfs.txt; ls $d/hwmon/*/ >> $R/clk_sysfs.txt 2>&1; cat $d/pp_dpm_sclk >> $R/clk_sysfs.txt 2>&1; cat $d/pp_dpm_mclk >> $R/clk_sysfs.txt 2>&1; done
python - > $R/clk_samples.txt 2>&1 <<'PY' &
import glob, time
fs = glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input') + glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') + glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input') + glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq2_input')
print(fs)
t0 = time.time()
while time.time() - t0 < 40:
    row = []
    for f in fs:
        try: row.append(open(f).read().strip())
        except Exception as e: row.append('x')
    print(f"{time.time()-t0:.3f}", *row, flush=True)
    time.sleep(0.05)
PY
sleep 3
timeout 100 python bench.py --steps 600 --warmup 3 --no-cpu-baseline --no-emission > $R/clk_bench.json 2> $R/clk_bench.err
sleep 3
(rocm-smi --showclocks --showpower > $R/clk_smi.txt 2>&1)
wait
cut -c1-120 $R/clk_bench.json
