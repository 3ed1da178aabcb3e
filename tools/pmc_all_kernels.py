#!/usr/bin/env python3
"""HBM-side traffic of EVERY kernel of a run from two rocprofv3 `--pmc` rocpd databases (FETCH_SIZE and WRITE_SIZE in separate passes): calls, read MB
(raw counter and doubled per the gfx950 note of MI355X_MICROARCH.md), written MB, per kernel and in total.
    python tools/pmc_all_kernels.py fetch.db write.db"""
import collections
import sqlite3
import sys


def table(path, counter):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    sfx = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    q = f"""select k.kernel_name, p.value from rocpd_pmc_event{sfx} p join rocpd_kernel_dispatch{sfx} d on p.event_id = d.event_id
            join rocpd_info_kernel_symbol{sfx} k on d.kernel_id = k.id join rocpd_info_pmc{sfx} i on p.pmc_id = i.id where i.name = ?"""
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, val in cur.execute(q, (counter,)):
        a = agg[name[:60]]; a[0] += 1; a[1] += val
    return agg


def main(fetch_db, write_db):
    f, w = table(fetch_db, "FETCH_SIZE"), table(write_db, "WRITE_SIZE")
    names = sorted(set(f) | set(w), key=lambda n: -(2 * f.get(n, [0, 0])[1] + w.get(n, [0, 0])[1]))
    print(f"{'kernel':60s} {'calls':>6s} {'read MB raw':>12s} {'read MB x2':>11s} {'write MB':>10s}")
    tf = tw = 0.0
    for n in names:
        cf, vf = f.get(n, [0, 0.0]); cw, vw = w.get(n, [0, 0.0])
        print(f"{n:60s} {max(cf, cw):6d} {vf / 1024:12.1f} {2 * vf / 1024:11.1f} {vw / 1024:10.1f}")
        tf += vf; tw += vw
    print(f"{'total':60s} {'':6s} {tf / 1024:12.1f} {2 * tf / 1024:11.1f} {tw / 1024:10.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
