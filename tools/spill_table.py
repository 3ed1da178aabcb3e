#!/usr/bin/env python3
"""Register / scratch usage of every gfx950 kernel in libpob_hip.so's objects, from the code-object metadata
(no GPU needed): kernel, VGPRs, AGPRs, SGPRs, VGPR spills, SGPR spills, scratch bytes, LDS bytes.

    python tools/spill_table.py [> profiles/roundN_spill_table.txt]
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = [".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size"]


def kernels_of(obj: str):
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb"), os.path.join(td, "co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], capture_output=True)
        if r.returncode != 0 or not os.path.exists(fb):
            return []
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*(\.\w+):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == ".agpr_count" or (k == ".args" and cur is None):
            pass
        if k == ".name" and not v.endswith(".kd") and v.startswith("_Z") or (k == ".name" and v.startswith("k_")):
            cur = {"name": v}
            out.append(cur)
        elif cur is not None and k in KEYS:
            cur[k] = int(v)
    return out


def demangle(n: str) -> str:
    try:
        return subprocess.run([f"{LLVM}/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def main():
    rows = []
    for obj in sorted(glob.glob(os.path.join(ROOT, "proof_of_burn_amd", "csrc", "*.o"))):
        for k in kernels_of(obj):
            if ".vgpr_count" in k:
                rows.append((os.path.basename(obj), demangle(k["name"]), k))
    print(f"{'object':22s} {'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>7s} {'sspill':>7s} {'scratch':>8s} {'lds':>6s}")
    bad = 0
    for obj, name, k in rows:
        name = re.sub(r"^void ", "", name)[:58]
        print(f"{obj:22s} {name:58s} {k.get('.vgpr_count', 0):5d} {k.get('.agpr_count', 0):5d} {k.get('.sgpr_count', 0):5d} "
              f"{k.get('.vgpr_spill_count', 0):7d} {k.get('.sgpr_spill_count', 0):7d} {k.get('.private_segment_fixed_size', 0):8d} {k.get('.group_segment_fixed_size', 0):6d}")
        bad += k.get(".vgpr_spill_count", 0) > 0 or k.get(".private_segment_fixed_size", 0) > 0
    print(f"# {len(rows)} kernels, {bad} with VGPR spills or scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
