"""Build libpob_hip.so in-tree: hipcc --offload-arch=gfx950, one object per translation unit (compiled in
parallel), linked -shared.  No CMake, no JIT cache: the .so travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
UNITS = ["g_gen_pos.hip", "g_emit_light.hip", "g_emit_heavy.hip", "g_check_pos.hip", "g_check_n2b.hip", "g_gen_n2b.hip", "g_gen_light.hip", "g_check_rl.hip", "g_check_ld.hip", "g_check_misc.hip", "g_check_sc.hip", "g_gen_sc.hip", "g_check_range.hip", "g_check_selrow.hip", "k_keccak.hip", "pob_host.hip"]
HEADERS = ["fr_dev.hpp", "policy.hpp", "gadgets.hpp", "circuits.hpp", "kernels_common.hpp", "g_units.hpp", "keccak_kernels.hpp",
           "poseidon_consts.h", os.path.join("..", "..", "include", "pob_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
LIB = os.path.join(CSRC, "libpob_hip.so")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, verbose: bool) -> str:
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if _newer(obj, deps):
        t0 = time.time()
        cmd = ["hipcc", *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
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
