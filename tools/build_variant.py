#!/usr/bin/env python3
"""A variant build of libpob_hip.so for an A/B on one GPU box: the in-tree objects with SOME translation units recompiled under extra -D switches, linked into ab/<name>.so
(git-ignored like every .so, but it travels to the box; POB_LIB_PATH=ab/<name>.so selects it, tools/gpu_ab_libs.sh interleaves the builds).

    python tools/build_variant.py nw4 g_check_narrow.hip:-DPOB_NARROW_WAVES=4 [more.hip:-DX=1,-DY=2 ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proof_of_burn_amd import build as B


def main():
    name, specs = sys.argv[1], sys.argv[2:]
    B.build(verbose=False)                               # the in-tree objects are current
    out_dir = os.path.join(ROOT, "ab")
    os.makedirs(out_dir, exist_ok=True)
    objs = {u: os.path.join(B.CSRC, u.replace(".hip", ".o")) for u in B.UNITS}
    procs = []
    for spec in specs:
        tu, _, defs = spec.partition(":")
        obj = os.path.join(out_dir, f"{name}_{tu.replace('.hip', '.o')}")
        flags = [f for f in B.FLAGS if f not in ("-Xarch_host", "-O1")] + ["-O3"] if tu in B.HOST_O3 else B.FLAGS
        cmd = ["hipcc", *flags, *[d for d in defs.split(",") if d], "-c", os.path.join(B.CSRC, tu), "-o", obj]
        procs.append((tu, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        objs[tu] = obj
    for tu, p in procs:
        err = p.communicate()[1]
        if p.returncode:
            raise SystemExit(f"hipcc failed for {tu}:\n{err[-3000:]}")
    lib = os.path.join(out_dir, f"libpob_{name}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs.values(), "-o", lib])
    for tu, _ in procs:
        os.remove(objs[tu])
    print(lib)


if __name__ == "__main__":
    main()
