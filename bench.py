#!/usr/bin/env python3
"""bench.py -- proof_of_burn witnesses/s on N MI355X GPUs (BASELINE.json metric).

A "step" = one pass of the hot path over one batch: witness generation (every wire of the O0 witness, resident in
HBM in the compact typed layout) + the per-gate constraint evaluation over that resident vector + the RCCL gather
of the per-witness results.  Workload at N=1: BASELINE.json configs[2] -- batch = 1024 proof_of_burn witnesses of
the production instantiation ProofOfBurn(16,4,16,50,31,2,1e19,1e20) on synthetic 10-layer MPT proofs; for N > 1
every rank gets its own 1024 (weak scaling), one slice per GPU, no data-path collective except the result gather.
Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"      # circuits/main_proof_of_burn.circom:27
HBM_PEAK_GBS = 8000.0                                                # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="witnesses per GPU")
    ap.add_argument("--depth", type=int, default=10, help="MPT proof depth of the synthetic inputs")
    ap.add_argument("--distinct-keys", type=int, default=16, help="distinct PoW burn keys tiled over the batch")
    ap.add_argument("--cpu-samples", type=int, default=2, help="witnesses timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from proof_of_burn_amd import WitnessCalculator, inputs as gen
    from proof_of_burn_amd import distributed as D

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    rank, local_rank, world = D.init()
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    dev_index = int(os.environ.get("POB_FORCE_DEVICE", local_rank))     # (test hook: several ranks on one GPU with POB_DIST_BACKEND=gloo)
    torch.cuda.set_device(dev_index)
    B = args.batch

    # ---- synthetic inputs (seeded; rank r generates witnesses [r*B, (r+1)*B) of the global batch)
    t0 = time.time()
    batch = gen.synthetic_batch(B, depth=args.depth, seed=0xB0B + rank * B, distinct_keys=args.distinct_keys)
    calc = WitnessCalculator(MAIN, max_batch=B, device=dev_index)
    fr, sm, forced = calc.pack(batch.inputs)
    calc.upload_packed(fr, sm, forced)                  # H2D happens here, outside the timed region
    setup_s = time.time() - t0
    stream = torch.cuda.current_stream().cuda_stream
    st_dev, out_dev = D.device_results(calc, B)

    def step():
        calc.generate(stream)
        calc.constraint_check(stream)
        return D.gather_results(st_dev, out_dev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st_all, out_all = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- the work was real: every witness valid, commitments equal the host-side formula, evaluator clean
    res = calc.results(with_check=True)
    assert all(r.ok for r in res), [r.message() for r in res if not r.ok][:3]
    assert [r.outputs[0] for r in res] == batch.commitments, "commitment mismatch"
    assert all(r.check_status == 0 and r.bad_wire is None for r in res), "constraint evaluator flagged a witness"
    assert int(st_all.shape[0]) == world * B and int((st_all != 0).sum().item()) == 0
    got0 = int.from_bytes(bytes(out_all[rank * B].cpu().numpy().tobytes()), "little")
    assert got0 == batch.commitments[0]

    info = calc.info
    groups = (B + 63) // 64
    # ---- roofline of the dominant kernel: Keccak round constraint evaluation, HBM-read bound.
    # algorithmic bytes per launch = every wire of every KeccakfRound block as resident (8 B per BIT wire per 64 witnesses)
    #                                + the round input/output states it is checked against.
    t_chk = calc.time_kernel(1, iters=5, stream=stream)
    t_gen = calc.time_kernel(0, iters=5, stream=stream)
    round_bytes = (102656 + 2 * 1600) * 8
    launch_bytes = info.n_perms * 24 * round_bytes * groups
    achieved = launch_bytes / (t_chk * 1e-3) / 1e9
    traffic = None                                        # HBM bytes per launch from the committed PMC passes (tools/pmc_summary.py)
    try:
        with open(os.path.join(ROOT, "profiles", "round1_pmc_k_rounds.json")) as f:
            pmc = json.load(f)
        traffic = int(pmc["k_rounds_check"]["hbm_read_bytes_per_launch"] * groups / pmc["groups"])
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "k_rounds<CHECK> (Keccak-f round constraint evaluation)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": launch_bytes, "avg_ms": round(t_chk, 4),
                "gen_kernel": {"kernel": "k_rounds<GEN>", "achieved": round(info.n_perms * 24 * (102656 + 1600) * 8 * groups / (t_gen * 1e-3) / 1e9, 1),
                               "avg_ms": round(t_gen, 4)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests import oracle_ffi as O
        O.run(MAIN, batch.inputs[0])                     # first run pays the page faults of a fresh 6.9 GB mapping
        t0 = time.perf_counter()
        for i in range(args.cpu_samples):
            r = O.run(MAIN, batch.inputs[1 + i])
            assert not r.failed and r.outputs() == [batch.commitments[1 + i]]
        cdt = time.perf_counter() - t0
        cpu = {"value": round(args.cpu_samples / cdt, 4), "unit": "witnesses/s", "cores": 1, "kind": "port",
               "sample": f"{args.cpu_samples} witnesses of the same synthetic batch on the C oracle (oracle/pob_oracle.c, a restatement: "
                         f"the circom-emitted calculator is not buildable here), canonical 32 B x {info.n_witness} wires each, single thread",
               "host_cores": os.cpu_count()}

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            "metric": "proof_of_burn witnesses/sec", "value": round(value, 1), "unit": "witnesses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 bit-sliced lanes + BN254 Fr (8x32-bit Montgomery)", "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU proof_of_burn witnesses, {MAIN}, synthetic {args.depth}-layer MPT proofs "
                                   f"({batch.distinct_keys} distinct PoW burn keys tiled), generate + per-gate constraint evaluation + result gather",
                       "wires_per_witness": int(info.n_witness), "resident_bytes_per_witness": int(info.group_bytes // 64),
                       "canonical_bytes_per_witness": int(info.n_witness) * 32, "parallelism": f"one slice per GPU x{world}", "setup_s": round(setup_s, 1)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    calc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
