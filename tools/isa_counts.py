#!/usr/bin/env python3
"""Memory-instruction counts of the Keccak round kernels in the gfx950 ISA of k_keccak.hip (no GPU needed): that the fused generation + evaluation kernel (k_rounds_gc)
really holds the evaluation's 101 array loads per round beside the generation's 76 stores -- i.e. that no load was replaced by the value stored before it.

    python tools/isa_counts.py [> profiles/roundN_isa_counts.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proof_of_burn_amd import build as B


def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        flags = [f for f in B.FLAGS if f not in ("-Xarch_host", "-O1")]
        subprocess.check_call(["hipcc", *flags, *extra, "--cuda-device-only", "-S", os.path.join(B.CSRC, "k_keccak.hip"), "-o", asm])
        s = open(asm).read()
    print(f"{'kernel':58s} {'ld.x2':>6s} {'st.x2':>6s} {'ds_wr':>6s} {'ds_rd':>6s} {'bperm':>6s} {'scratch':>7s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s}")
    for m in re.finditer(r"^(_Z\w+):\s.*?\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "k_rounds" not in name and "k_chain" not in name:
            continue
        c = lambda pat: len(re.findall(pat, body))
        g = lambda key: (re.search(rf"\.amdhsa_{key} (\d+)", body) or [0, "?"])[1]
        n = [c(pat) for pat in ("global_load_dwordx2", "global_store_dwordx2", "ds_write[0-9a-z]*_b64", "ds_read[0-9a-z]*_b64", "ds_bpermute_b32", "scratch_")]
        print(f"{name:58s} {n[0]:6d} {n[1]:6d} {n[2]:6d} {n[3]:6d} {n[4]:6d} "
              f"{n[5]:7d} {g('next_free_vgpr'):>5s} {g('next_free_sgpr'):>5s} {g('group_segment_fixed_size'):>6s}")
    print("# per round of the walk: 76 stores (generation), 101 loads (evaluation: 76 stored arrays + midRound[r+1]); + 25 loads of midRound[r0] per wavefront")


if __name__ == "__main__":
    main()
