#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 rocpd database: kernels longer than a threshold, per queue."""
import sqlite3, sys
def main(path, which=3, thr=0.15):
    db = sqlite3.connect(path); cur = db.cursor()
    sfx = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace("rocpd_kernel_dispatch", "")
    rows = cur.execute(f"select k.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.queue_id, d.stream_id from rocpd_kernel_dispatch{sfx} d join rocpd_info_kernel_symbol{sfx} k on d.kernel_id = k.id order by d.start").fetchall()
    idx = [i for i, r in enumerate(rows) if 'k_collect' in r[0]]
    names = {'_Z8k_roundsILb1EEv5KArgs.kd': 'K_CHK', '_Z8k_roundsILb0EEv5KArgs.kd': 'K_GEN', '_Z7k_chainILb0EEv5KArgs.kd': 'chainG', '_Z13k_chain_check5KArgs.kd': 'chainC'}
    def short(n):
        if n in names: return names[n]
        if 'g_units' in n: return ('gCHK' if 'CheckP' in n else 'gGEN') + '[' + n.split('Lj')[1].split('E')[0] + ']'
        if 'k_poseidon_wide' in n: return 'posWide'
        return n[:20]
    sel = rows[idx[which - 1]:idx[which] + 1]
    t0 = sel[0][1]
    for n, s, e, gx, gy, q, st in sel:
        if (e - s) > thr * 1e6 or 'K_' in short(n):
            print(f"{(s-t0)/1e6:8.3f} {(e-t0)/1e6:8.3f} {(e-s)/1e6:7.3f} q{q} s{st} {short(n):12s} grid {gx//64}x{gy}")
    print("collect-to-collect:", (rows[idx[which]][1] - rows[idx[which-1]][1]) / 1e6, "ms")
if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, float(sys.argv[3]) if len(sys.argv) > 3 else 0.15)
