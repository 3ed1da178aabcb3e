#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`) as the per-kernel
statistics table that `--stats` prints: calls, total / average / min / max duration, share of GPU time."""
import collections
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    sfx = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    rows = cur.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, k.arch_vgpr_count, k.sgpr_count "
                       f"from rocpd_kernel_dispatch{sfx} d join rocpd_info_kernel_symbol{sfx} k on d.kernel_id = k.id order by d.start").fetchall()
    agg = collections.OrderedDict()
    for n, s, e, gx, gy, vg, sg in rows:
        a = agg.setdefault(n, {"calls": 0, "tot": 0, "min": 1 << 62, "max": 0, "vgpr": vg, "sgpr": sg})
        a["calls"] += 1; a["tot"] += e - s; a["min"] = min(a["min"], e - s); a["max"] = max(a["max"], e - s)
    tot = sum(a["tot"] for a in agg.values())
    print(f"{'kernel':64s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s}")
    for k, a in sorted(agg.items(), key=lambda x: -x[1]["tot"]):
        print(f"{k[:64]:64s} {a['calls']:6d} {a['tot'] / 1e6:10.3f} {a['tot'] / a['calls'] / 1e3:10.1f} {a['min'] / 1e3:10.1f} {a['max'] / 1e3:10.1f} "
              f"{100 * a['tot'] / tot:6.2f} {a['vgpr'] or 0:5d} {a['sgpr'] or 0:5d}")


if __name__ == "__main__":
    main(sys.argv[1])
