"""TEST INFRASTRUCTURE: builds tests/hostsim/libpob_hostsim.so -- the unchanged sources of libpob_hip.so compiled for the host CPU
against the "HIP on fibers" shim (hip/hip_runtime.h here), so that `pytest -m "not gpu"` can run the product's own kernels and
host scheduler on small circuits.  Never loaded by the product."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from proof_of_burn_amd import build as pb  # noqa: E402

CXX = "/opt/rocm/lib/llvm/bin/clang++"
OBJ = os.path.join(HERE, "obj")
LIB = os.path.join(HERE, "libpob_hostsim.so")
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-DPOB_HOSTSIM",  "-D__HIPCC__", "-I", HERE, "-I", pb.CSRC,
         "-Wno-unused-value", "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-undefined-inline", "-fbracket-depth=1024"]


SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]


def _compile(src: str, san: bool = False) -> str:
    obj = os.path.join(OBJ + ("_san" if san else ""), os.path.basename(src).rsplit(".", 1)[0] + ".o")
    deps = [src, os.path.join(HERE, "hip", "hip_runtime.h")] + [os.path.join(pb.CSRC, h) for h in pb.HEADERS]
    if not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
        r = subprocess.run([CXX, *FLAGS, *(SAN_FLAGS if san else []), "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hostsim: {os.path.basename(src)} failed:\n{r.stderr[-6000:]}")
    return obj


def build(san: bool | None = None) -> str:
    """san (default: POB_HOSTSIM_SAN=1 in the environment): the AddressSanitizer + UndefinedBehaviorSanitizer build, libpob_hostsim_san.so; the
    process that loads it must have the clang sanitizer runtime preloaded (tools/run_sanitizers.py)"""
    if san is None:
        san = os.environ.get("POB_HOSTSIM_SAN") == "1"
    lib = LIB.replace(".so", "_san.so") if san else LIB
    os.makedirs(OBJ + ("_san" if san else ""), exist_ok=True)
    srcs = [os.path.join(pb.CSRC, u) for u in pb.UNITS] + [os.path.join(HERE, "hostsim_rt.cpp")]
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s_: _compile(s_, san), srcs))
    if not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        r = subprocess.run([CXX, "-shared", "-fPIC", *(SAN_FLAGS + ["-shared-libasan"] if san else []), *objs, "-o", lib], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hostsim link failed:\n{r.stderr[-4000:]}")
    return lib


if __name__ == "__main__":
    print(build())
