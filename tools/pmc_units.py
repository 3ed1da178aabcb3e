#!/usr/bin/env python3
"""Per-kernel SQ counters of a rocprofv3 --pmc rocpd database, grouped by (kernel, grid): where the wave cycles of each G kernel go.
    python tools/pmc_units.py DB"""
import collections
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    sfx = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    q = f"""select k.kernel_name, d.grid_size_x, d.grid_size_y, i.name, p.value, d.id from rocpd_pmc_event{sfx} p
            join rocpd_kernel_dispatch{sfx} d on p.event_id = d.event_id join rocpd_info_kernel_symbol{sfx} k on d.kernel_id = k.id
            join rocpd_info_pmc{sfx} i on p.pmc_id = i.id"""
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for name, gx, gy, ctr, val, did in cur.execute(q):
        key = (name[:48], gx // 64, gy)
        agg[key][ctr] += val; calls[key].add(did)
    ctrs = sorted({c for v in agg.values() for c in v})
    print(f"{'kernel':50s} {'grid':>10s} {'n':>3s} " + " ".join(f"{c[3:][:14]:>14s}" for c in ctrs))
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        n = len(calls[key])
        print(f"{key[0]:50s} {key[1]:>6d}x{key[2]:<3d} {n:3d} " + " ".join(f"{v.get(c, 0) / n:14.0f}" for c in ctrs))


if __name__ == "__main__":
    main(sys.argv[1])
