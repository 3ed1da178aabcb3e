#!/usr/bin/env python3
"""Per-launch durations of ONE batch on a lone in-order calculator (nothing else on the device), in launch order, from a rocprofv3 --kernel-trace rocpd database of
tools/pmc_one_batch.py: what every launch of the step costs when it has the machine to itself.
    python tools/lone_batch_trace.py DB"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_step import short  # noqa: E402


def main(path):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch')); ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    rows = cur.execute(f"select k.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y from {kd} d join {ks} k on d.kernel_id=k.id order by d.start").fetchall()
    col = [i for i, r in enumerate(rows) if 'k_collect' in r[0]]
    a, b = col[-3], col[-1]          # the last batch: behind the collect of the previous evaluation up to its own evaluation's collect
    t0 = rows[a][2]; tot = 0.0; gen = 0.0; prev_end = t0
    for n, s, e, gx, gy in rows[a + 1:b + 1]:
        d = (e - s) / 1e3; tot += d
        print(f"{(s - t0) / 1e3:9.1f} us  +{d:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {short(n)} {gx // 64}x{gy}")
        prev_end = e
    print(f"sum of kernel times {tot:.1f} us, span {(rows[b][2] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
