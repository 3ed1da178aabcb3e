#!/usr/bin/env python3
"""Extract per-launch FETCH_SIZE / WRITE_SIZE of the Keccak round kernels from rocprofv3 `--pmc` rocpd databases
(separate passes, as MI355X_MICROARCH.md prescribes) and write profiles/<name>.json, which bench.py reads for `traffic`.

gfx950 correction (MI355X_MICROARCH.md "HBM"): FETCH_SIZE counts 64 B per 128-B request of a coalesced stream, i.e. exactly
half the bytes -> doubled here (calibrated on this kernel family in round 3: the full-layout kernel streamed 27 GB that cannot sit in the 256 MB L3
and read 0.51 x raw); WRITE_SIZE is used as reported (it matched the algorithmic write bytes to 0.2 %).  Round 4: the compact round blocks (76 stored
arrays of 64 wires per block instead of 1 604).  Round 5: a wavefront covers several consecutive rounds; every resident array is counted once."""
import json
import sqlite3
import sys


def per_launch(path, counter, kernel_like):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    sfx = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    q = f"""select d.grid_size_x, d.grid_size_y, p.value from rocpd_pmc_event{sfx} p
            join rocpd_kernel_dispatch{sfx} d on p.event_id = d.event_id join rocpd_info_kernel_symbol{sfx} k on d.kernel_id = k.id
            join rocpd_info_pmc{sfx} i on p.pmc_id = i.id where k.kernel_name like ? and i.name = ? order by d.start"""
    rows = [r for r in cur.execute(q, (kernel_like, counter)).fetchall()]
    if not rows:
        return None
    full = max(r[0] for r in rows)
    vals = [r[2] for r in rows if r[0] == full]
    return {"launches": len(vals), "grid_x_threads": full, "groups": rows[0][1], "avg_kb": sum(vals) / len(vals)}


def main(fetch_db, write_db, out):
    chk = per_launch(fetch_db, "FETCH_SIZE", "%k_rounds_check%")
    gen = per_launch(write_db, "WRITE_SIZE", "%k_rounds_gen%")
    groups = chk["groups"]
    KCHK, KGEN = 4, 8                          # rounds per wavefront (keccak_kernels.hpp POB_KCHK_ROUNDS / POB_KGEN_ROUNDS)
    chunks = chk["grid_x_threads"] // 64       # (permutation, chunk) items of the evaluation launch
    res = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --batch %d" % (groups * 64),
        "groups": groups,
        "k_rounds_check": {"fetch_size_kb_raw": chk["avg_kb"], "hbm_read_bytes_per_launch": chk["avg_kb"] * 1024 * 2,
                           # every resident array of a chunk once: midRound[r0] + per round the 76 stored gate outputs and the stored midRound[r+1]
                           "algorithmic_bytes_per_launch": chunks * groups * (101 * KCHK + 25) * 64 * 8, "rounds_per_wavefront": KCHK},
        "k_rounds_gen": {"write_size_kb_raw": gen["avg_kb"], "hbm_write_bytes_per_launch": gen["avg_kb"] * 1024,
                         "algorithmic_bytes_per_launch": (gen["grid_x_threads"] // 64) * groups * KGEN * 76 * 64 * 8, "rounds_per_wavefront": KGEN},      # the 76 stored arrays of each round
    }
    for k in ("k_rounds_check", "k_rounds_gen"):
        d = res[k]
        d["traffic_over_algorithmic"] = round((d.get("hbm_read_bytes_per_launch") or d.get("hbm_write_bytes_per_launch")) / d["algorithmic_bytes_per_launch"], 4)
    # round 6: the launch that expands AND evaluates the round blocks (k_rounds_gc, keccak_kernels.hpp): algorithmic = the expansion's stores + the evaluation's loads; the loads of
    # what the wavefront has just stored are served by L2, so the traffic is BELOW the algorithmic bytes -- by how much is what these passes measure
    gcf, gcw = per_launch(fetch_db, "FETCH_SIZE", "%k_rounds_gc%"), per_launch(write_db, "WRITE_SIZE", "%k_rounds_gc%")
    if gcf and gcw:
        items = gcf["grid_x_threads"] // 64
        KGC = 24 * (chunks // (24 // KCHK)) // items            # rounds per wavefront: permutations x 24 rounds / items
        store, load = items * groups * KGC * 76 * 64 * 8, items * groups * (101 * KGC + 25) * 64 * 8
        rd, wr = gcf["avg_kb"] * 1024 * 2, gcw["avg_kb"] * 1024
        res["k_rounds_gc"] = {"fetch_size_kb_raw": gcf["avg_kb"], "write_size_kb_raw": gcw["avg_kb"], "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                              "algorithmic_store_bytes_per_launch": store, "algorithmic_load_bytes_per_launch": load, "rounds_per_wavefront": KGC,
                              "stored_states_bytes_per_launch": load - store,         # midRound[r0] + every midRound[r+1]: loads that no store of the launch precedes
                              "loads_of_own_stores_served_by_cache": round(1 - (rd - (load - store)) / store, 4),
                              "traffic_over_algorithmic": round((rd + wr) / (store + load), 4)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
