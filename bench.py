#!/usr/bin/env python3
"""bench.py -- proof_of_burn witnesses/s on N MI355X GPUs (BASELINE.json metric).

A "step" = one pass of the hot path over one batch: witness generation (every wire of the O0 witness, resident in
HBM in the compact typed layout) + the per-gate constraint evaluation over that resident vector + the RCCL gather
of the per-witness result records.  Workload at N=1: BASELINE.json configs[2] -- batch = 1024 proof_of_burn witnesses of
the production instantiation ProofOfBurn(16,4,16,50,31,2,1e19,1e20) on synthetic 10-layer MPT proofs; for N > 1
every rank gets its own 1024 (weak scaling), one slice per GPU, no data-path collective except ONE all-gather of the
36-byte result records.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

--halves H (default 1): the rank's batch may be held by H calculators (B/H witnesses each, own streams) whose passes are enqueued
interleaved, so that the latency-bound stages of one part run beside the HBM-streaming Keccak kernels of the other.  Measured
(profiles/round2_*): H = 2 gains nothing -- both kinds of kernel wait on the same memory system -- so the default is one calculator.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # every stream of the job needs its own hardware queue (ROCm default: 4): 2 callers + gather + 7 of the library (+ RCCL's at N > 1)

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"      # circuits/main_proof_of_burn.circom:27
HBM_PEAK_GBS = 8000.0                                                # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def _cpu_worker(args):
    """one CPU-baseline process: k witnesses on the C oracle (a restatement of the circom-emitted calculator), single thread"""
    main, inputs, commitments = args
    sys.path.insert(0, ROOT)
    from tests import oracle_ffi as O
    t0 = time.perf_counter()
    for inp, c in zip(inputs, commitments):
        r = O.run(main, inp)
        assert not r.failed and r.outputs() == [c]
    return time.perf_counter() - t0


def cpu_baseline(batch, info, single_samples: int, budget_s: float = 25.0):
    import multiprocessing as mp
    from tests import oracle_ffi as O
    O.run(MAIN, batch.inputs[0])                     # first run pays the page faults of a fresh 6.9 GB mapping
    t0 = time.perf_counter()
    for i in range(single_samples):
        r = O.run(MAIN, batch.inputs[1 + i])
        assert not r.failed and r.outputs() == [batch.commitments[1 + i]]
    t_single = (time.perf_counter() - t0) / single_samples
    # N processes in parallel on the host cores, memory-capped: one canonical witness is 6.9 GB + the oracle's scratch
    per_proc_gb = 9.0
    try:
        with open("/proc/meminfo") as f:
            avail_gb = next(int(line.split()[1]) for line in f if line.startswith("MemAvailable")) / 1e6
    except Exception:
        avail_gb = 32.0
    cores = os.cpu_count() or 1
    nproc = max(1, min(cores, int(avail_gb * 0.4 / per_proc_gb), 32))
    per = max(1, min(4, int(budget_s / max(t_single * 1.5, 0.1))))
    jobs = [(MAIN, [batch.inputs[(p * per + j) % len(batch.inputs)] for j in range(per)],
             [batch.commitments[(p * per + j) % len(batch.inputs)] for j in range(per)]) for p in range(nproc)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(nproc) as pool:
        pool.map(_cpu_worker, jobs)
    t_par = time.perf_counter() - t0
    return {"value": round(nproc * per / t_par, 3), "unit": "witnesses/s", "cores": nproc, "kind": "port",
            "single_thread": round(1.0 / t_single, 4), "host_cores": cores, "mem_available_gb": round(avail_gb, 1), "mem_cap_gb_per_process": per_proc_gb,
            "sample": f"{nproc} processes x {per} witnesses of the same synthetic batch on the C oracle (oracle/pob_oracle.c, a restatement: the circom-emitted "
                      f"calculator is not buildable here), canonical 32 B x {info.n_witness} wires each, one thread per process, spawn + first-touch included; "
                      f"single_thread = {single_samples} witnesses on one warm process"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1024, help="witnesses per GPU")
    ap.add_argument("--halves", type=int, default=int(os.environ.get("POB_BENCH_HALVES", "1")), help="calculators per GPU whose passes are interleaved")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("POB_BENCH_PIPELINE", "1")),
                    help="1: two calculators work on consecutive batches (pob_set_partner): batch k+1's latency-bound generation stages run beside batch k's evaluation")
    ap.add_argument("--depth", type=int, default=10, help="MPT proof depth of the synthetic inputs (16 = BASELINE config 5)")
    ap.add_argument("--distinct-keys", type=int, default=16, help="distinct PoW burn keys tiled over the global batch")
    ap.add_argument("--cpu-samples", type=int, default=2, help="witnesses timed single-threaded on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-emission", action="store_true", help="skip the .wtns emission throughput measurement")
    ap.add_argument("--dump-results", default=None, help="rank 0 writes the gathered records (uint8 [N*B, 36]) to this .npy")
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
    B, H = args.batch, max(1, args.halves)
    PIPE = bool(args.pipeline)
    if PIPE:
        H = 1
    while H > 1 and (B % H or (B // H) % 64):
        H -= 1                                                          # parts are whole 64-witness groups
    Bh = B // H

    # ---- synthetic inputs (seeded; rank r holds witnesses [r*B, (r+1)*B) of the global batch: witness g depends only on (seed, g))
    t0 = time.time()
    batch = gen.synthetic_batch(B, depth=args.depth, seed=0xB0B, distinct_keys=args.distinct_keys, first=rank * B,
                                pow_device=dev_index if args.depth > 12 else None)
    t_synth = time.time() - t0
    NC = 2 if PIPE else H                                 # pipeline: two calculators, each holds a whole batch (consecutive batches of the job)
    calcs = [WitnessCalculator(MAIN, max_batch=Bh, device=dev_index) for _ in range(NC)]
    t0 = time.time()
    packed = [calcs[h].pack(batch.inputs[h * Bh:(h + 1) * Bh]) for h in range(H)]
    t_pack = time.time() - t0
    t0 = time.time()
    for c in range(NC):
        calcs[c].upload_packed(*packed[c if not PIPE else 0])             # H2D happens here, outside the timed region
    t_h2d = time.time() - t0
    streams = [torch.cuda.Stream(device=dev_index) for _ in range(NC)]     # (not the legacy default stream: it synchronises with every blocking stream)
    recs = [D.device_records(calcs[c], Bh) for c in range(NC)]
    if PIPE:
        calcs[0].set_partner(calcs[1]); calcs[1].set_partner(calcs[0])

    gs = torch.cuda.Stream(device=dev_index)              # result records: packed at the end of generation, gathered beside the evaluation

    def step():
        for h in range(H):
            streams[h].wait_stream(gs)                    # the previous step's gather has read the records this generation overwrites
            calcs[h].generate(streams[h].cuda_stream)
        for h in range(H):
            gs.wait_stream(streams[h])
        with torch.cuda.stream(gs):
            rec = recs[0] if H == 1 else torch.cat(recs)
            out = D.gather_records(rec)
        for h in range(H):
            calcs[h].constraint_check(streams[h].cuda_stream)
        return out

    gathered = [None, None]
    used = set() if PIPE else set(range(NC))              # calculators that have generated a batch (a one-step pipelined run uses one)

    def run_pipelined(nsteps):
        """nsteps batches through the two-calculator pipeline, fill and drain included: batch k is generated by calculator k % 2 while
        batch k-1 is evaluated by the other one; every batch is generated AND evaluated inside the call"""
        out, prev = None, None
        for k in range(nsteps):
            cur = k % 2
            used.add(cur)
            if prev is not None:
                calcs[prev].constraint_check(streams[prev].cuda_stream)
            if gathered[cur] is not None:
                streams[cur].wait_event(gathered[cur])    # the gather of THIS calculator's previous batch has read its records
            calcs[cur].generate(streams[cur].cuda_stream)
            gs.wait_stream(streams[cur])
            with torch.cuda.stream(gs):
                out = D.gather_records(recs[cur])
                gathered[cur] = torch.cuda.Event(); gathered[cur].record(gs)
            prev = cur
        if prev is not None:
            calcs[prev].constraint_check(streams[prev].cuda_stream)
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if PIPE:
        if args.warmup:
            run_pipelined(args.warmup)
        fence()
        t0 = time.perf_counter()
        rec_all = run_pipelined(args.steps)
        fence()
        dt = time.perf_counter() - t0
    else:
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rec_all = step()
        fence()
        dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- the work was real: every witness valid, commitments equal the host-side formula, evaluator clean
    for h in sorted(used):
        res = calcs[h].results(with_check=True)
        assert all(r.ok for r in res), [r.message() for r in res if not r.ok][:3]
        assert [r.outputs[0] for r in res] == (batch.commitments if PIPE else batch.commitments[h * Bh:(h + 1) * Bh]), "commitment mismatch"
        assert all(r.check_status == 0 and r.bad_wire is None for r in res), "constraint evaluator flagged a witness"
    st_all, out_all = D.unpack_records(rec_all.cpu())
    assert int(st_all.shape[0]) == world * B and int((st_all != 0).sum().item()) == 0
    mine = out_all[rank * B:(rank + 1) * B].numpy()
    assert [int.from_bytes(bytes(mine[i].tobytes()), "little") for i in range(B)] == batch.commitments, "gathered records differ from this rank's commitments"
    if rank == 0 and args.dump_results:
        np.save(args.dump_results, rec_all.cpu().numpy())

    # ---- the same job without the pipeline (one calculator, generate -> gather -> evaluate per batch), 10 steps: reported beside `value`
    # so that both are measured in the same run on the same box
    single = None
    if PIPE and 0 in used:
        calcs[0].set_partner(None)

        def step1():
            streams[0].wait_stream(gs)
            calcs[0].generate(streams[0].cuda_stream)
            gs.wait_stream(streams[0])
            with torch.cuda.stream(gs):
                D.gather_records(recs[0])
            calcs[0].constraint_check(streams[0].cuda_stream)

        for _ in range(2):
            step1()
        fence()
        t1 = time.perf_counter()
        for _ in range(10):
            step1()
        fence()
        dt1 = time.perf_counter() - t1
        if world > 1:
            t1max = torch.tensor([dt1], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(t1max, op=dist.ReduceOp.MAX)
            dt1 = float(t1max.item())
        single = {"what": "one calculator per GPU, no pipeline (bench.py --pipeline 0), 10 steps after the timed region", "value": round(world * B * 10 / dt1, 1),
                  "ms_per_step": round(dt1 / 10 * 1e3, 3)}
        calcs[0].set_partner(calcs[1])

    info = calcs[0].info
    groups_h = (Bh + 63) // 64
    stream0 = streams[0].cuda_stream
    # ---- roofline of the dominant kernel: Keccak round constraint evaluation, HBM-read bound.
    # algorithmic bytes per launch = every wire of every KeccakfRound block as resident (8 B per BIT wire per 64 witnesses)
    #                                + the round input/output states it is checked against.  A launch covers one calculator's part.
    t_chk = calcs[0].time_kernel(1, iters=5, stream=stream0)
    t_gen = calcs[0].time_kernel(0, iters=5, stream=stream0)
    round_bytes = (102656 + 2 * 1600) * 8
    launch_bytes = info.n_perms * 24 * round_bytes * groups_h
    achieved = launch_bytes / (t_chk * 1e-3) / 1e9
    # whole evaluation pass and whole step against the resident vector (write once, read once)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record(streams[0])
    for _ in range(5):
        for h in range(H):
            calcs[h].constraint_check(streams[h].cuda_stream)
    for h in range(1, H):
        streams[0].wait_stream(streams[h])
    ev1.record(streams[0])
    torch.cuda.synchronize()
    t_check_pass = ev0.elapsed_time(ev1) / 5
    resident = int(info.group_bytes) * groups_h * H
    traffic, pmc_file = None, None                        # HBM bytes per launch of the dominant kernel from the committed PMC passes (not measured in this run)
    try:
        pmc_file = next(p for p in ("round2_pmc_k_rounds.json", "round1_pmc_k_rounds.json") if os.path.exists(os.path.join(ROOT, "profiles", p)))
        with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
            pmc = json.load(f)
        traffic = int(pmc["k_rounds_check"]["hbm_read_bytes_per_launch"] * groups_h / pmc["groups"])
    except Exception:
        pass
    ms_step = dt / args.steps * 1e3
    roofline = {"bound": "hbm", "kernel": "k_rounds<CHECK> (Keccak-f round constraint evaluation)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": f"profiles/{pmc_file} (separate rocprofv3 --pmc FETCH_SIZE pass over this kernel, scaled to this launch's groups; not measured in this run)" if traffic else None,
                "bytes_per_launch": launch_bytes, "avg_ms": round(t_chk, 4),
                "gen_kernel": {"kernel": "k_rounds<GEN>", "achieved": round(info.n_perms * 24 * (102656 + 1600) * 8 * groups_h / (t_gen * 1e-3) / 1e9, 1),
                               "avg_ms": round(t_gen, 4)},
                "check_pass": {"what": "whole pob_constraint_check over the resident vector (all G families + Keccak rounds + chains)", "bytes": resident,
                               "ms": round(t_check_pass, 3), "achieved": round(resident / (t_check_pass * 1e-3) / 1e9, 1),
                               "frac": round(resident / (t_check_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "step": {"what": "generate (write the resident vector once) + evaluate (read it once)", "bytes": 2 * resident,
                         "achieved": round(2 * resident / (ms_step * 1e-3) / 1e9, 1), "frac": round(2 * resident / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}

    # ---- .wtns emission (the step after the path): 2 witnesses back to back through the window pipeline into pinned host memory
    emission = None
    if rank == 0 and not args.no_emission:
        sec, nbytes = calcs[0].emit_throughput(0, count=2)
        emission = {"what": "canonical 32 B/wire payload expanded on the GPU in 256 MiB windows, D2H double-buffered into pinned memory, 2 witnesses back to back",
                    "GB_per_s": round(nbytes / sec / 1e9, 2), "ms_per_witness": round(sec / 2 * 1e3, 1), "bytes_per_witness": nbytes // 2}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(batch, info, args.cpu_samples)

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            "metric": "proof_of_burn witnesses/sec", "value": round(value, 1), "unit": "witnesses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 bit-sliced lanes + BN254 Fr (8x32-bit Montgomery)", "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU proof_of_burn witnesses, {MAIN}, synthetic {args.depth}-layer MPT proofs "
                                   f"({args.distinct_keys} distinct PoW burn keys tiled), generate + per-gate constraint evaluation + result gather",
                       "wires_per_witness": int(info.n_witness), "resident_bytes_per_witness": int(info.group_bytes // 64),
                       "canonical_bytes_per_witness": int(info.n_witness) * 32, "parallelism": f"one slice per GPU x{world}, " + (f"two calculators pipelined over consecutive batches of {Bh} (fill and drain inside the timed region)" if PIPE else f"{H} interleaved parts of {Bh} per GPU"),
                       "input_synthesis_s": round(t_synth, 2), "json_to_packed_witnesses_per_s": round(B / max(t_pack, 1e-9), 1),
                       "h2d_s": round(t_h2d, 3)},
            "roofline": roofline, "cpu_baseline": cpu, "emission": emission, "single_calculator": single,
        }
        print(json.dumps(line))
    for c in calcs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
