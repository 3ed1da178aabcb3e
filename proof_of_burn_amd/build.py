"""Build libpob_hip.so in-tree: hipcc --offload-arch=gfx950, one object per translation unit (compiled in
parallel), linked -shared.  No CMake, no JIT cache: the .so travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
UNITS = ["g_gen_poswide.hip", "g_emit_light.hip", "g_emit_heavy.hip", "g_check_pos.hip", "g_check_n2b.hip", "g_gen_n2b.hip", "g_gen_light.hip", "g_gen_all.hip", "g_gen_all_ride.hip", "g_check_narrow.hip", "g_check_rl.hip", "g_check_ld.hip", "g_check_misc.hip", "g_check_sc.hip", "g_gen_sc.hip", "g_check_range.hip", "g_check_selrow.hip", "k_keccak.hip", "pob_host.hip", "pack_json.hip", "g_gen_gm.hip", "g_check_gm.hip", "g_emit_gm.hip"]
HEADERS = ["fr_dev.hpp", "policy.hpp", "gadgets.hpp", "circuits.hpp", "kernels_common.hpp", "g_units.hpp", "keccak_kernels.hpp", "poseidon_wide.hpp", "gadget_mains.hpp",
           "poseidon_consts.h", os.path.join("..", "..", "include", "pob_hip.h")]
# (host pass at -O1: the only host code of any size is the layout planner, which runs once per pob_open)
FLAGS = ["--offload-arch=gfx950", "-O3", "-Xarch_host", "-O1", "-std=c++17", "-fPIC", "-Wno-unused-result"]
HOST_O3 = {"pack_json.hip"}          # host-only translation units whose speed matters (the input.json loader): host pass at -O3 too
LIB = os.path.join(CSRC, "libpob_hip.so")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(src: str, obj: str):
    """the headers a translation unit really includes (hipcc -MD dependency file of its last build); every header if there is none"""
    dfile = obj[:-2] + ".d"
    if os.path.exists(dfile) and os.path.exists(obj):
        with open(dfile) as f:
            toks = f.read().replace("\\\n", " ").split()
        found = [t for t in toks[1:] if os.path.exists(t) and (t.startswith(CSRC) or "/include/pob_hip.h" in t)]
        if found:
            return found
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    if _newer(obj, _deps(src, obj)):
        t0 = time.time()
        flags = [f for f in FLAGS if f not in ("-Xarch_host", "-O1")] + ["-O3"] if src in HOST_O3 else FLAGS
        cmd = ["hipcc", *flags, "-MD", "-MF", obj[:-2] + ".d", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"  hipcc {src}: {time.time() - t0:.0f}s", flush=True)
    return obj


def build(force: bool = False, verbose: bool = True, jobs: int | None = None) -> str:
    if force:
        for u in UNITS:
            o = os.path.join(CSRC, u.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda u: _compile(u, verbose), UNITS))
    if _newer(LIB, objs):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"  linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
