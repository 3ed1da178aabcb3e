#!/usr/bin/env python3
"""One steady-state step of a pipelined bench run from a rocprofv3 rocpd database (--kernel-trace [--memory-copy-trace]): every kernel longer
than `thr` ms and every memory copy between two result-collect kernels, with stream ids.
    python tools/rocpd_step.py DB [first_collect_index] [thr_ms]"""
import sqlite3
import sys


def short(n):
    for key, lab in (('k_gen_level', 'gGEN+K_GEN'), ('k_pos_chain', 'pos+chainG'), ('k_check_wide', 'K_CHK+wide'), ('k_check_narrow', 'narrow+chainC')):      # fused launches (round 6)
        if key in n:
            return lab
    if 'k_rounds' in n:
        return 'K_GC' if 'k_rounds_gc' in n else 'K_CHK' if 'k_rounds_check' in n else 'K_GEN'
    if 'g_units' in n:
        return ('gCHK' if 'CheckP' in n else 'gEMIT' if 'EmitP' in n else 'gGEN') + '[' + n.split('Lj')[1].split('E')[0] + ']'
    for key, lab in (('poseidon', 'posWide'), ('k_chain_check', 'chainC'), ('k_chain', 'chainG'), ('k_inputs', 'inputs'), ('k_collect', 'collect')):
        if key in n:
            return lab
    return n[:18]


def main(path, first=10, thr=0.06):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch')); ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    mc = next((t for t in tabs if t.startswith('rocpd_memory_copy')), None)
    rows = cur.execute(f"select k.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y,d.stream_id from {kd} d join {ks} k on d.kernel_id=k.id order by d.start").fetchall()
    col = [i for i, r in enumerate(rows) if 'k_collect' in r[0]]
    a, b = col[first], col[first + 4]
    t0 = rows[a][1]
    ev = [(s, e, f"s{st} {short(n)} {gx // 64}x{gy}") for n, s, e, gx, gy, st in rows[a:b + 1]]
    if mc:
        ev += [(s, e, f"s{st} COPY {sz} B {sa}->{da}") for s, e, sz, sa, da, st in cur.execute(f"select start,end,size,src_agent_id,dst_agent_id,stream_id from {mc} where start>={t0} and start<={rows[b][2]}")]
    for s, e, lab in sorted(ev):
        if e - s > thr * 1e6 or 'COPY' in lab or 'collect' in lab:
            print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:6.3f} {lab}")
    print("two steps:", (rows[b][1] - t0) / 1e6, "ms")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10, float(sys.argv[3]) if len(sys.argv) > 3 else 0.06)
