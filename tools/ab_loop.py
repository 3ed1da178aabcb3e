#!/usr/bin/env python3
"""Service-loop A/B on ONE box, one process per library build: the production batch is synthesised once, then every (depth, streaming-stream) point is measured ROUNDS times,
interleaved -- throughput, the round evaluation kernel's in-step duration, and (once) the whole evaluation alone.  The library build comes from POB_LIB_PATH
(tools/build_variant.py), so a shell loop over builds interleaves them too (tools/gpu_r6_ab.sh).

    python tools/ab_loop.py --label nw4 --points 4:0,8:0,12:0,8:1 --steps 96 --rounds 2
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BM


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--label", default="new")
    ap.add_argument("--points", default="4:0,8:0,12:0", help="depth:fused, ...")
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--nbatches", type=int, default=4)
    ap.add_argument("--cumask", default="none", help="none | parity: calculator c on the CUs with index parity c % 2 | quarters: c % 4 of every 4 CUs | halves: lower / upper half of the CU indices")
    ap.add_argument("--alone", action="store_true", help="also time the whole evaluation / generation of one calculator alone and the family kernels")
    a = ap.parse_args()
    import numpy as np
    from proof_of_burn_amd import PinnedInputs, inputs as gen
    args = BM.parse_args(["--gpus", "1", "--batch", str(a.batch)])
    job = BM.Job(args)
    B = job.B
    batches = [gen.synthetic_batch(B, depth=a.depth, seed=0xB0B, distinct_keys=16, first=b * B, pow_device=job.dev if a.depth > 12 else None) for b in range(a.nbatches)]
    expect = [BM._expect(np, bt) for bt in batches]
    pinned = None
    points = [tuple(int(x) for x in p.split(":")) for p in a.points.split(",")]
    for r in range(a.rounds):
        for depth, ss in points:
            args.fused = ss
            lp = BM.ServiceLoop(job, BM.MAIN, depth, True, ss)
            if pinned is None:
                pinned = [PinnedInputs(lp.calcs[0], B) for _ in batches]
                for pin, bt in zip(pinned, batches):
                    lp.calcs[0].pack_json([json.dumps(i).encode() for i in bt.inputs], out=pin)
            if a.cumask != "none":          # partition the device between the calculators: every calculator's stream restricted to a share of the compute units
                import ctypes
                from proof_of_burn_amd import witness as W
                lib = W.load_library()
                lib.pob_debug_stream_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
                ncu, words = 256, 8
                class Raw:
                    def __init__(self, ptr): self.cuda_stream = ptr
                for c in range(depth):
                    if a.cumask == "parity": cus = [i for i in range(ncu) if i % 2 == c % 2]
                    elif a.cumask == "quarters": cus = [i for i in range(ncu) if i % 4 == c % 4]
                    else: cus = [i for i in range(ncu) if (i < ncu // 2) == (c % 2 == 0)]
                    m = (ctypes.c_uint32 * words)()
                    for i in cus: m[i >> 5] |= 1 << (i & 31)
                    p = ctypes.c_void_p()
                    assert lib.pob_debug_stream_create(job.dev, m, words, ctypes.byref(p)) == 0
                    lp.streams[c] = Raw(p.value)
            lp.set_inputs(pinned, expect)
            lp.run(max(depth, 1) + 2); job.fence()
            lp.probe(True)
            s, _, _ = lp.timed(a.steps, 2, k0=max(depth, 1) + 2)
            k = float(np.mean(lp.kchk_ms)) if lp.kchk_ms else 0.0
            lp.probe(False)
            extra = ""
            if a.alone and r == 0 and depth == points[0][0]:
                c0, st0 = lp.calcs[0], lp.streams[0]
                cuda = job.cuda
                def span(fn, n=5):
                    e0, e1 = cuda.Event(enable_timing=True), cuda.Event(enable_timing=True)
                    cuda.synchronize(); e0.record(st0)
                    for _ in range(n):
                        fn()
                    e1.record(st0); cuda.synchronize()
                    return e0.elapsed_time(e1) / n
                t_eval = span(lambda: c0.constraint_check(st0.cuda_stream))
                t_gen = span(lambda: c0.generate(st0.cuda_stream))
                fams = {f: round(c0.time_kernel(300 + f, iters=5, stream=st0.cuda_stream), 4) for f in range(8)}
                extra = f" | alone: evaluation {t_eval:.3f} ms generation {t_gen:.3f} ms K_CHK {c0.time_kernel(1, iters=5, stream=st0.cuda_stream):.4f} K_GEN {c0.time_kernel(0, iters=5, stream=st0.cuda_stream):.4f} K_GC {c0.time_kernel(6, iters=5, stream=st0.cuda_stream):.4f} families {fams}"
            print(f"round {r} {a.label} cumask {a.cumask} depth {depth} fused {ss}: {s / a.steps * 1e3:.3f} ms/step = {s / a.steps * 1e3 * 1024 / B:.3f} ms per 1024  {B * a.steps / s:.0f} w/s  K_CHK in step {k:.4f} ms{extra}", flush=True)
            lp.close()
            del lp
    for pin in pinned:
        pin.free()


if __name__ == "__main__":
    main()
