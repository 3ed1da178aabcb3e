#!/usr/bin/env python3
"""bench.py -- proof_of_burn witnesses/s on N MI355X GPUs (BASELINE.json metric).

A "step" = one batch through the hot path as a SERVICE LOOP would run it: the batch's packed inputs go H2D from pinned memory
(pob_upload_inputs8_async), every wire of the O0 witness is generated (resident in HBM in the compact typed layout), the per-gate
constraint evaluation reads that resident vector back, the per-witness result records {status, evaluator verdict, commitment} are packed
AFTER the evaluation, copied to pinned host memory and VALIDATED on the host (every record of every batch: status 0, evaluator clean,
commitment equal to the host-side formula) -- all of it inside the timed region.  Consecutive batches carry DIFFERENT inputs
(--distinct-batches of them, cycled).  The loop keeps --pipeline IN-ORDER calculators in flight, each on a stream of its own, on consecutive
batches (pob_set_inorder: a calculator's whole batch in dependency order on one stream); --schedule tracks is round 3's
two-calculator pipeline.  Workload at N=1: BASELINE.json configs[2] -- batch = 1024 proof_of_burn witnesses of the
production instantiation ProofOfBurn(16,4,16,50,31,2,1e19,1e20) on synthetic 10-layer MPT proofs; for N > 1 every rank gets its own 1024
(weak scaling; --total-batch B splits ONE global batch over the ranks instead: BASELINE config 4 as written), one slice per GPU, no
data-path collective except ONE all-gather of the 44-byte result records per batch.

Launch forms (both are tested):
    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                      WORLD_SIZE unset: bench.py starts its N ranks itself (one process per GPU, rendezvous on 127.0.0.1:free port)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Structure: `ServiceLoop` is the loop (calculators, streams, pinned inputs, counters); every leg of the report is a function that takes it.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # every stream of the job needs its own hardware queue (ROCm default: 4); libpob_hip.so refuses the pipeline below 12

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"      # circuits/main_proof_of_burn.circom:27
HBM_PEAK_GBS = 8000.0                                                # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
DEFAULT_DEPTH = 12                                                   # in-order calculators in flight: the fastest depth measured (4 / 6 / 8 / 10 / 12 / 16: profiles/round6_experiments.txt)
PMC_FILES = ("round6_pmc_k_rounds.json", "round5_pmc_k_rounds.json")


# ------------------------------------------------------------------------------------------------ CPU baseline (the oracle, test infrastructure: this leg only)
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


def _cpu_quota():
    """CPUs the container's cgroup grants this process (cpu.max), None = no quota"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        return None


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


# ------------------------------------------------------------------------------------------------ command line, self-launch
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=180, help="timed batches (180 x ~1.5 ms = a 0.27 s timed region)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="witnesses per GPU per step (weak scaling)")
    ap.add_argument("--total-batch", type=int, default=0, help="strong scaling: ONE global batch of this many witnesses per step, split over the ranks (BASELINE config 4: 8192)")
    ap.add_argument("--schedule", choices=["inorder", "tracks"], default="inorder",
                    help="inorder: every calculator enqueues its whole batch in dependency order on ONE stream (pob_set_inorder) and --pipeline of them are in flight; "
                         "tracks: round 2-3's schedule, two linked calculators (pob_set_partner) whose tracks run on the device's side streams")
    ap.add_argument("--pipeline", type=int, default=-1, help=f"calculators in flight, each on consecutive batches (default: {DEFAULT_DEPTH} in-order ones / 2 linked track ones; 0 = one calculator, no pipeline)")
    ap.add_argument("--fused", type=int, default=3, choices=[0, 1, 2, 3],
                    help="default 3.  bit 0: in-order calculators with the fused Poseidon + sponge-chain launch (pob_set_inorder bit 1); bit 1: the evaluation rides with the generation "
                         "(pob_set_inorder bit 2: the Keccak round blocks, the input rows and the G units' wires are loaded back from L2 and compared by the launches that store them; of the "
                         "evaluation pass the sponge chains' and the RLP family's kernels are left); 0: one launch per kernel, round 5's schedule")
    ap.add_argument("--depth", type=int, default=10, help="MPT proof depth of the synthetic inputs (16 = BASELINE config 5)")
    ap.add_argument("--distinct-keys", type=int, default=16, help="distinct PoW burn keys tiled over a global batch")
    ap.add_argument("--distinct-batches", type=int, default=4, help="different input batches cycled through the steps (every one is uploaded anew each time)")
    ap.add_argument("--cpu-samples", type=int, default=2, help="witnesses timed single-threaded on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-emission", action="store_true", help="skip the .wtns emission throughput measurements")
    ap.add_argument("--no-single", action="store_true", help="skip the single-calculator / e2e / bare-pipeline legs after the timed region")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the other-depths / depth16 / strong_slice legs after the timed region")
    ap.add_argument("--other-depths", default="4,8", help="pipeline depths measured beside the default one after the timed region (rank 0 of a 1-GPU run)")
    ap.add_argument("--probe-after", action="store_true", help="record the HIP events around the Keccak round evaluation kernel in 16 extra steps after the timed region instead of inside it")
    ap.add_argument("--dbg-no-upload", action="store_true", help="experiment: upload each calculator's inputs once, not per batch")
    ap.add_argument("--dbg-no-fetch", action="store_true", help="experiment: no per-batch record fetch / validation inside the loop")
    ap.add_argument("--main", choices=["proof_of_burn", "spend"], default="proof_of_burn", help="spend: the same service loop on Spend(31) (tests: small enough for the CPU shim)")
    ap.add_argument("--shim", action="store_true", help="TESTS ONLY: run the loop on the CPU shim of the kernels (tests/hostsim) -- no GPU, no timing claims, no roofline; "
                                                        "exercises the multi-rank plumbing (slices, pinned buffers, loader width, the records' all-gather) under gloo")
    ap.add_argument("--dump-results", default=None, help="rank 0 writes the gathered records of the LAST batch (uint8 [N*B, 44]) to this .npy")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks here, one process per GPU, with the environment torch.distributed.run would give
    them (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT on 127.0.0.1 and a port the kernel says is free).  Rank 0 inherits stdout, so the ONE
    JSON line is this command's output; the other ranks' stdout goes to stderr.  Any rank failing ends the job with its exit code (the others are terminated)."""
    from proof_of_burn_amd.distributed import free_port
    n = args.gpus
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), POB_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=None if r == 0 else sys.stderr) for r in range(n)]
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------ the job's environment: device API, ranks
class _ShimCuda:
    """tests/hostsim: the product's kernels and host scheduler on CPU fibers; streams and events do not exist there (a launch runs synchronously)"""
    class Stream:
        cuda_stream = 0
        def __init__(self, *a, **k): pass
        def wait_event(self, e): pass
        def wait_stream(self, s): pass
    class Event:
        def __init__(self, *a, **k): self.t = 0.0
        def record(self, s=None): self.t = time.perf_counter()
        def elapsed_time(self, o): return (o.t - self.t) * 1e3
    is_available = staticmethod(lambda: True)
    set_device = staticmethod(lambda d: None)
    synchronize = staticmethod(lambda: None)

    @staticmethod
    def stream(s):
        import contextlib
        return contextlib.nullcontext()


class Job:
    """ranks, device, the device API (torch.cuda or the tests' shim) and the job-wide reductions"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from proof_of_burn_amd import distributed as D
        from proof_of_burn_amd import witness as W
        self.args, self.torch, self.dist, self.D, self.W = args, torch, dist, D, W
        if args.shim:
            from tests.hostsim import build as hb
            W.LIB_PATH, W._lib = hb.build(), None
            os.environ.setdefault("POB_DIST_BACKEND", "gloo")
            self.cuda = _ShimCuda
        else:
            self.cuda = torch.cuda
        assert self.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
        self.rank, self.local_rank, self.world = D.init()
        assert self.world == args.gpus, f"WORLD_SIZE={self.world} but --gpus {args.gpus}"
        self.dev = int(os.environ.get("POB_FORCE_DEVICE", self.local_rank))     # (test hook: several ranks on one GPU with POB_DIST_BACKEND=gloo)
        self.cuda.set_device(self.dev)
        # this rank's host threads next to its GPU, before any pinned buffer is allocated or the loader pool starts (a lone rank keeps the whole host)
        self.bound_cpus = D.bind_rank_to_gpu_numa(self.local_rank, None if args.shim else self.dev) if (self.world > 1 or os.environ.get("POB_BIND_NUMA") == "1") else None
        self.strong = args.total_batch > 0
        if self.strong:
            lo, hi = D.shard_bounds(args.total_batch, self.rank, self.world)
            self.B, self.first0, self.GB = hi - lo, lo, args.total_batch          # this rank's slice of every global batch
        else:
            self.B, self.first0, self.GB = args.batch, self.rank * args.batch, self.world * args.batch
        self.backend = dist.get_backend() if dist.is_initialized() else None

    def fence(self):
        self.cuda.synchronize()
        if self.world > 1:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.dev])         # (this rank's GPU, said explicitly: without it the barrier guesses the device from the rank)
            else:
                self.dist.barrier()
        self.cuda.synchronize()

    def _tensor(self, vals):
        return self.torch.tensor(vals, dtype=self.torch.float64, device="cpu" if self.backend == "gloo" else "cuda")

    def max_over_ranks(self, seconds: float) -> float:
        if self.world == 1:
            return seconds
        t = self._tensor([seconds])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(self, value: float):
        """the value of every rank, in rank order"""
        if self.world == 1:
            return [value]
        out = self._tensor([0.0] * self.world)
        self.dist.all_gather_into_tensor(out, self._tensor([value]))
        return [float(x) for x in out.cpu()]


# ------------------------------------------------------------------------------------------------ the service loop
class ServiceLoop:
    """`depth` calculators in flight on consecutive batches.  Per batch, all inside run(): H2D of the packed inputs from pinned memory, generate, constraint evaluation, records into
    pinned host memory, every record validated on the host; at N > 1 one all-gather of the device records.  Pipeline: batch k is generated by calculator k % depth while batch k-1 is
    evaluated by the one before; the host validates the oldest batch after it has enqueued the newest, so the device never waits for the host."""

    @staticmethod
    def inorder_mode(fused: int) -> int:
        """pob_set_inorder's argument for --fused: bit 0 of `fused` = the Poseidon + sponge-chain launch, bit 1 = the round blocks evaluated by the launch that writes them"""
        return 1 | (2 if int(fused) & 1 else 0) | (4 if int(fused) & 2 else 0)

    def __init__(self, job: Job, main: str, depth: int, inorder: bool = True, fused: int = 1):
        import numpy as np
        from proof_of_burn_amd import WitnessCalculator
        self.job, self.np, self.main = job, np, main
        self.B, self.depth, self.inorder = job.B, max(1, depth), inorder
        self.link = (not inorder) and depth > 1
        cuda, D = job.cuda, job.D
        self.calcs = [WitnessCalculator(main, max_batch=self.B, device=job.dev) for _ in range(self.depth)]
        if inorder:
            for c in self.calcs:
                c.set_inorder(self.inorder_mode(fused))
        # (high priority: the callers' streams carry the main track of the generation, whose chain bounds the read phase; -0.3 % on the step)
        self.streams = [cuda.Stream(device=job.dev, priority=-1) for _ in self.calcs]     # (not the legacy default stream: it synchronises with every blocking stream)
        records_of = D.host_records if job.args.shim else D.device_records
        self.recs = [records_of(c, self.B) for c in self.calcs]
        if self.link:
            self.calcs[0].set_partner(self.calcs[1]); self.calcs[1].set_partner(self.calcs[0])
        self.gs = cuda.Stream(device=job.dev)              # the record gather of the multi-GPU job: after the batch's evaluation, beside the next batch's work
        self.gathered_ev = [None] * self.depth
        self.gather_bad = job.torch.zeros(1, dtype=job.torch.int64, device="cpu" if job.args.shim else f"cuda:{job.dev}")      # witnesses of OTHER ranks with a non-clean record
        self.last_gather = None
        self.uploaded = [False] * self.depth
        self.pinned, self.expect = [], []                   # the input batches the loop cycles through (set_inputs)
        self.active = self.depth                            # calculators the loop uses (the single / tracks legs use fewer)
        self.upload_each, self.fetch_each, self.probing = not job.args.dbg_no_upload, not job.args.dbg_no_fetch, False
        self.validated = self.h2d_bytes = 0
        self.kchk_ms = []
        self.host_s = {"t_gen": 0.0, "t_chk": 0.0, "t_wait": 0.0}

    # ---- inputs
    def set_inputs(self, pinned, expect):
        self.pinned, self.expect = list(pinned), list(expect)

    def reset_counters(self):
        self.validated = self.h2d_bytes = 0
        self.kchk_ms = []
        for k in self.host_s:
            self.host_s[k] = 0.0

    def probe(self, on: bool):
        """HIP events around the dominant kernel (Keccak round evaluation) of every evaluation from here on"""
        self.probing = on
        for c in self.calcs:
            c.probe_check_kernel(on)

    # ---- one batch
    def start(self, c: int, pin):
        calc, st = self.calcs[c], self.streams[c]
        if self.upload_each or not self.uploaded[c]:
            # H2D from pinned memory on the device's upload stream (byte form: 11.4 KB per witness), right in front of the generation that reads it.  (Sending a calculator's
            # NEXT batch on its way right after its generate -- the inputs are double-buffered -- was measured: 1.74 against 1.61-1.63 ms per step; round-5 experiments 14)
            self.h2d_bytes += calc.upload_pinned_async(pin)
            self.uploaded[c] = True
        if self.gathered_ev[c] is not None:
            st.wait_event(self.gathered_ev[c])                             # the gather of THIS calculator's previous batch has read its records
        _t = time.perf_counter()
        calc.generate(st.cuda_stream)
        self.host_s["t_gen"] += time.perf_counter() - _t

    def finish(self, c: int):
        """evaluation of calculator c's batch, its records (now with the verdict) to the host and to the other ranks"""
        job, D, cuda = self.job, self.job.D, self.job.cuda
        calc, st = self.calcs[c], self.streams[c]
        _t = time.perf_counter()
        calc.constraint_check(st.cuda_stream)
        self.host_s["t_chk"] += time.perf_counter() - _t
        if self.fetch_each:
            calc.fetch_records()
        if job.world > 1:
            self.gs.wait_stream(st)
            with cuda.stream(self.gs):
                out = D.gather_records(self.recs[c], total=job.GB if job.strong else None)
                status, _ = D.unpack_records(out)
                cs, bw = D.unpack_verdicts(out)
                self.gather_bad.add_(((status != 0) | (cs != D.CLEAN) | (bw != D.CLEAN)).sum())
                self.gathered_ev[c] = cuda.Event(); self.gathered_ev[c].record(self.gs)
                self.last_gather = out

    def validate(self, c: int, b: int):
        """every record of the batch calculator c has just finished: host-visible, checked before the clock stops"""
        np, W = self.np, self.job.W
        if not self.fetch_each:
            self.validated += self.B
            return
        _t = time.perf_counter()
        rec = self.calcs[c].wait_records()
        self.host_s["t_wait"] += time.perf_counter() - _t
        assert rec.shape[0] == self.B
        assert not rec["status"].any(), ("a witness failed", np.nonzero(rec["status"])[0][:4], rec["status"][np.nonzero(rec["status"])[0][:4]])
        assert (rec["check_status"] == W.CLEAN).all() and (rec["bad_wire"] == W.CLEAN).all(), "the constraint evaluator flagged a witness (or did not run)"
        assert np.array_equal(rec["commitment"], self.expect[b]), "commitment mismatch"
        self.validated += self.B
        if self.probing:
            self.kchk_ms.append(self.calcs[c].probe_check_kernel(True, read=True))

    # ---- the loop
    def run(self, nsteps: int, k0: int = 0, source=None):
        """nsteps batches, fill and drain included: every batch is uploaded, generated, evaluated, fetched and validated inside the call.
        source(k) -> the pinned inputs of batch k (default: the loop's own, cycled); it may block (the e2e leg waits for its loader there) and returns its stall"""
        nc, nb = self.active, max(1, len(self.pinned))
        pend, prev, stall = [], None, 0.0           # pend: (calculator, distinct batch) enqueued but not yet validated, oldest first
        for k in range(k0, k0 + nsteps):
            c, b = k % nc, k % nb
            if nc > 1:
                if prev is not None:
                    self.finish(prev[0]); pend.append(prev)
                if source is not None:
                    _t = time.perf_counter(); pin = source(k); stall += time.perf_counter() - _t
                else:
                    pin = self.pinned[b]
                self.start(c, pin)
                prev = (c, b)
                while len(pend) > nc - 1:
                    self.validate(*pend.pop(0))
            else:
                self.start(c, source(k) if source is not None else self.pinned[b])
                self.finish(c); pend.append((c, b))
                self.validate(*pend.pop(0))
        if prev is not None:
            self.finish(prev[0]); pend.append(prev)
        while pend:
            self.validate(*pend.pop(0))
        return stall

    def timed(self, nsteps: int, warm: int, k0: int = 0, source=None):
        """warm untimed steps, fence, nsteps timed steps, fence: (seconds: the max over the ranks, this rank's seconds, source stall)"""
        if warm:
            self.run(warm, k0=k0, source=source)
        self.job.fence()
        self.reset_counters()
        t0 = time.perf_counter()
        stall = self.run(nsteps, k0=k0 + warm, source=source)
        self.job.fence()
        mine = time.perf_counter() - t0
        assert self.validated == nsteps * self.B, "not every batch was validated inside the timed region"
        return self.job.max_over_ranks(mine), mine, stall

    def close(self):
        for c in self.calcs:
            c.close()


def _rate(job, loop, seconds, steps, **extra):
    return dict({"ms_per_step": round(seconds / steps * 1e3, 3), "value": round(job.GB * steps / seconds, 1), "unit": "witnesses/s", "steps": steps, "batch": loop.B,
                 "validated_witnesses": steps * loop.B}, **extra)


def _expect(np, batch):
    return np.array([list(c.to_bytes(32, "little")) for c in batch.commitments], dtype=np.uint8)


# ------------------------------------------------------------------------------------------------ legs after the timed region
def leg_e2e_from_json(job, loop, texts, nsteps, ms_step_packed_ahead=None):
    """the path from input.json TEXT, in the clock (reference Makefile:4-5: the calculator's unit of work starts at the JSON file): the same service loop, but every batch is
    parsed from its texts by the native loader (pob_pack_json_batch8 on the persistent loader pool) INSIDE the timed region -- a host thread packs batch k + 2 into a ring of
    pinned buffers while the device works on batches k, k - 1, ...; the loop waits for the packer only if it has fallen behind"""
    from concurrent.futures import ThreadPoolExecutor
    from proof_of_burn_amd import PinnedInputs
    NC, NB, calc0 = loop.active, len(texts), loop.calcs[0]
    ring = [PinnedInputs(calc0, loop.B) for _ in range(NC + 3)]
    ex = ThreadPoolExecutor(1)
    busy = [0.0]
    fut = {}

    def pack(k):
        _t = time.perf_counter()
        calc0.pack_json(texts[k % NB], out=ring[k % len(ring)])
        busy[0] += time.perf_counter() - _t
        return ring[k % len(ring)]

    def make_source(k_end):
        def source(k):
            if k not in fut:                                 # the first batches of a run: nothing was packed ahead
                fut[k] = ex.submit(pack, k)
                if k + 1 < k_end:
                    fut[k + 1] = ex.submit(pack, k + 1)
            pin = fut.pop(k).result()
            if k + 2 < k_end and k + 2 not in fut:
                fut[k + 2] = ex.submit(pack, k + 2)          # its ring slot held batch k - NC - 1: validated before the loop returns to it
            return pin
        return source
    warm = NC + 2
    loop.run(warm, k0=0, source=make_source(warm)); job.fence()
    loop.reset_counters(); busy[0] = 0.0
    t1 = time.perf_counter()
    stall = loop.run(nsteps, k0=warm, source=make_source(warm + nsteps)); job.fence()
    dte = job.max_over_ranks(time.perf_counter() - t1)
    assert loop.validated == nsteps * loop.B
    ex.shutdown()
    for pin in ring:
        pin.free()
    loader_ms = busy[0] / nsteps * 1e3
    return {"what": "the same service loop with the loader inside the timed region: input.json TEXT -> pob_pack_json_batch8 (persistent host thread pool, byte form straight into pinned "
                    "memory) -> H2D -> generate -> evaluate -> records validated; a host thread parses batch k + 2 while the device works on batch k",
            "steps": nsteps, "ms_per_step": round(dte / nsteps * 1e3, 3), "value": round(job.GB * nsteps / dte, 1), "unit": "witnesses/s", "validated_witnesses": nsteps * loop.B,
            "loader_ms_per_batch": round(loader_ms, 3), "loader_witnesses_per_s": round(loop.B / max(loader_ms, 1e-9) * 1e3, 1), "host_waited_for_loader_ms_per_step": round(stall / nsteps * 1e3, 3),
            "bound": ("device (the loader keeps ahead: within 5 % of the step with the inputs packed ahead)" if (ms_step_packed_ahead and dte / nsteps * 1e3 <= 1.05 * ms_step_packed_ahead)
                      else "loader: the device loop waited for the packer -- host CPU time, see loader_cpu" if stall > 0.3 * dte else "device (the loader keeps ahead)"),
            "loader_cpu": {"host_cpus_visible": os.cpu_count(), "cgroup_cpu_quota": _cpu_quota(), "what": "a batch of 1 024 production input.json texts (40 KB, 10 900 values each) is "
                           "loader_ms_per_batch x the loader's threads of CPU time; a host that grants this process Q CPUs packs at most Q / that many batches per second"}}


def leg_kernel_pipeline_only(job, loop):
    """the bare kernel pipeline (what round 2's bench timed): the same calculators and batches, but the inputs stay resident (no per-batch H2D), no records are read per batch
    and nothing is validated inside the loop -- the results are checked once afterwards.  Same run, same box: the difference to `value` is what the service loop costs."""
    loop.upload_each = loop.fetch_each = False
    loop.run(loop.active); job.fence()
    loop.reset_counters()
    t1 = time.perf_counter()
    loop.run(20, k0=loop.active); job.fence()
    dtb = job.max_over_ranks(time.perf_counter() - t1)
    loop.upload_each, loop.fetch_each = not job.args.dbg_no_upload, not job.args.dbg_no_fetch
    for c in loop.calcs[:loop.active]:
        res = c.results(with_check=True)
        assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in res)
    return {"what": "round 2's timed loop on this build: inputs resident, no per-batch upload / record read-back / host validation; 20 batches after the timed region",
            "value": round(job.GB * 20 / dtb, 1), "ms_per_step": round(dtb / 20 * 1e3, 3)}


def legs_track_schedule(job, loop):
    """the same service loop on ONE calculator with the track schedule (the lowest latency for a lone batch), and on round 3's pipeline (two linked track calculators)"""
    def timed(n):
        s, _, _ = loop.timed(n, 2)
        return {"value": round(job.GB * n / s, 1), "ms_per_step": round(s / n * 1e3, 3)}
    if loop.link:
        loop.calcs[0].set_partner(None)
    for c in loop.calcs[:2]:
        c.set_inorder(0)
    loop.active = 1
    single = dict(timed(10), what="one calculator per GPU, track schedule, no pipeline (bench.py --schedule tracks --pipeline 0): 10 batches of the same service loop after the timed region")
    loop.calcs[0].set_partner(loop.calcs[1]); loop.calcs[1].set_partner(loop.calcs[0])
    loop.active = 2
    tracks = dict(timed(20), what="round 3's schedule on this build (bench.py --schedule tracks): two linked track calculators (pob_set_partner), 20 batches after the timed region")
    if not loop.link:
        loop.calcs[0].set_partner(None)
        for c in loop.calcs[:2]:
            c.set_inorder(ServiceLoop.inorder_mode(job.args.fused))
    loop.active = loop.depth
    return single, tracks


def leg_other_depth(job, main, depth, pinned, expect, nsteps=96, fused=None, what=None):
    """the same service loop with another number of in-order calculators in flight (or another --fused mode): throughput and the round evaluation kernel's in-step duration"""
    np = __import__("numpy")
    lp = ServiceLoop(job, main, depth, True, job.args.fused if fused is None else fused)
    lp.set_inputs(pinned, expect)
    lp.run(depth + 2); job.fence()
    lp.probe(True)
    s, _, _ = lp.timed(nsteps, 2, k0=depth + 2)
    k = float(np.mean(lp.kchk_ms)) if lp.kchk_ms else None
    lp.probe(False)
    lp.close()
    return _rate(job, lp, s, nsteps, what=what or f"{depth} in-order calculators in flight, {nsteps} steps of the same service loop after the timed region",
                 calculators_in_flight=depth, round_evaluation_ms_in_step=(round(k, 4) if k else None))


def leg_other_inputs(job, loop, texts_of, batches, nsteps, what):
    """the same loop on other input batches.  They are packed into the loop's OWN pinned buffers (the ones the timed region read: round 5's leg allocated fresh ones and measured
    17 % slower on the driver's box, pinning artefact or not), warmed with depth + 4 steps, and the original batches are packed back afterwards."""
    np = loop.np
    calc0 = loop.calcs[0]
    keep_expect = loop.expect
    n = min(len(batches), len(loop.pinned))
    for pin, bt in zip(loop.pinned[:n], batches[:n]):
        calc0.pack_json([json.dumps(inp).encode() for inp in bt.inputs], out=pin)
    keep_pinned = loop.pinned
    loop.set_inputs(keep_pinned[:n], [_expect(np, bt) for bt in batches[:n]])
    s, _, _ = loop.timed(nsteps, loop.depth + 4)
    loop.set_inputs(keep_pinned, keep_expect)
    for pin, tx in zip(loop.pinned, texts_of):
        calc0.pack_json(tx, out=pin)
    return _rate(job, loop, s, nsteps, what=what)


def leg_roofline(job, loop, info, kchk_in_step, ms_step, probe_steps):
    """roofline of the dominant kernel, HBM bound.  --fused bit 1 (default): k_rounds_gc, the launch that expands the Keccak round blocks AND evaluates them (the probe events of
    the timed region bracket it); else k_rounds_check, the round evaluation as a launch of its own.  Algorithmic bytes per launch: what the work has to store / to load, stated
    in `bytes_convention`; `traffic`: what reached HBM (PMC passes committed under profiles/)."""
    cuda, args = job.cuda, job.args
    gc = bool(args.fused & 2) and loop.inorder
    groups = (loop.B + 63) // 64
    calc0, st0 = loop.calcs[0], loop.streams[0]
    t_chk = calc0.time_kernel(1, iters=5, stream=st0.cuda_stream)
    t_gen = calc0.time_kernel(0, iters=5, stream=st0.cuda_stream)
    t_gc = calc0.time_kernel(6, iters=5, stream=st0.cuda_stream)
    arr = 64 * 8                                          # one `signal x[64]` array of 64 witnesses: 512 B
    # k_rounds_check: a wavefront covers k = info.kchk_rounds consecutive rounds of a permutation: midRound[r0] once, then per round the 76 stored gate-output arrays + the stored
    # midRound[r+1] (which stays in registers as the next round's input): 101 k + 25 arrays, every resident array of the chunk counted ONCE
    kr = int(info.kchk_rounds)
    chk_bytes = info.n_perms * (24 // kr) * (101 * kr + 25) * arr * groups
    launch_bytes_r4 = info.n_perms * 24 * 126 * arr * groups
    # k_rounds_gen: 8 rounds per wavefront: 8 x 76 arrays stored, midRound[r0] loaded.  k_rounds_gc: g rounds per wavefront: per round 76 arrays stored (the expansion) and
    # 101 loaded (the evaluation: the 76 back + the stored midRound[r+1]), midRound[r0] once
    gen_bytes = info.n_perms * 3 * (8 * 76 + 25) * arr * groups
    g = int(info.kgc_rounds)
    gc_store, gc_load = info.n_perms * 24 * 76 * arr * groups, info.n_perms * (24 // g) * (101 * g + 25) * arr * groups
    gc_once = info.n_perms * (24 // g) * (101 * g + 25) * arr * groups        # every resident array of the chunk once, whether written, evaluated or both
    launch_bytes, t_alone = (gc_store + gc_load, t_gc) if gc else (chk_bytes, t_chk)
    alone = launch_bytes / (t_alone * 1e-3) / 1e9
    in_step = launch_bytes / (kchk_in_step * 1e-3) / 1e9 if kchk_in_step else None
    # the whole evaluation of a resident vector as a pass of its own (round kernel included: after a first, untimed pass nothing stands for "evaluated with the expansion"),
    # and one lone batch generated + evaluated the way the loop does it
    def span(fn, n=5):
        ev0, ev1 = cuda.Event(enable_timing=True), cuda.Event(enable_timing=True)
        cuda.synchronize()
        ev0.record(st0)
        for _ in range(n):
            fn()
        ev1.record(st0)
        cuda.synchronize()
        return ev0.elapsed_time(ev1) / n
    calc0.constraint_check(st0.cuda_stream)
    t_check_pass = span(lambda: calc0.constraint_check(st0.cuda_stream))
    t_lone = span(lambda: (calc0.generate(st0.cuda_stream), calc0.constraint_check(st0.cuda_stream)))
    resident = int(info.group_bytes) * groups
    traffic, pmc_file, pmc_key = None, None, "k_rounds_gc" if gc else "k_rounds_check"      # HBM bytes per launch of the dominant kernel from the committed PMC passes (not measured in this run)
    try:
        for cand in PMC_FILES:
            path = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(path):
                with open(path) as f:
                    pmc = json.load(f)
                if pmc_key in pmc:
                    pmc_file = cand
                    traffic = int((pmc[pmc_key]["hbm_read_bytes_per_launch"] + (pmc[pmc_key].get("hbm_write_bytes_per_launch", 0) if gc else 0)) * groups / pmc["groups"])
                    break
    except Exception:
        pass
    # `achieved` / `frac`: the kernel as it runs IN the timed loop (HIP events on its own stream around every launch, averaged over the timed
    # steps), i.e. beside the other batches' kernels; `frac_alone`: the same kernel alone on an idle device (5 back-to-back launches)
    out = {"bound": "hbm",
           "kernel": "k_rounds_gc (Keccak-f round blocks: expansion + constraint evaluation in one launch)" if gc else "k_rounds<CHECK> (Keccak-f round constraint evaluation)",
           "achieved": round(in_step if in_step else alone, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round((in_step if in_step else alone) / HBM_PEAK_GBS, 4),
           "measured": ((f"in {probe_steps} more batches of the same pipelined service loop right after the timed region" if args.probe_after else "in the timed region, every step")
                        + f": HIP events on the kernel's own stream around each of its launches, mean over them (pob_probe_check_kernel); {loop.depth} calculators in flight") if in_step else "alone",
           "avg_ms": round(kchk_in_step if kchk_in_step else t_alone, 4),
           "frac_alone": round(alone / HBM_PEAK_GBS, 4), "achieved_alone": round(alone, 1), "avg_ms_alone": round(t_alone, 4),
           "traffic": traffic,
           "traffic_source": (f"profiles/{pmc_file} (separate rocprofv3 --pmc FETCH_SIZE" + (" and WRITE_SIZE passes" if gc else " pass") + " over this kernel, scaled to this launch's groups; not measured in this run)") if traffic else None,
           "bytes_per_launch": launch_bytes}
    if gc:
        out.update({"rounds_per_wavefront": g,
                    "bytes_convention": f"ALGORITHMIC bytes of the two jobs the launch does: the expansion stores permutations x 24 rounds x 76 arrays x 512 B x groups = {gc_store} B, the evaluation "
                                        f"loads permutations x {24 // g} chunks x (101 x {g} + 25) arrays = {gc_load} B of stored values (the sum of what k_rounds_gen stores and k_rounds_check loads).  Of the "
                                        f"loads, the {gc_store} B the wavefront has just stored come back from L2, so HBM sees less than the algorithmic bytes (`traffic`); counting every resident "
                                        f"array of the launch ONCE, written or evaluated or both: {gc_once} B (`each_once`)",
                    "each_once": {"bytes": gc_once, "frac": round(gc_once / ((kchk_in_step if kchk_in_step else t_alone) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frac_alone": round(gc_once / (t_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                    "hbm_traffic_frac_alone": round(traffic / (t_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                    "as_two_launches": {"what": "the two kernels this launch replaces, alone: k_rounds_gen (8 rounds per wavefront) and k_rounds_check (pob_constraint_check's round kernel: what evaluates a "
                                                "resident vector when no launch has just written it)",
                                        "gen_ms": round(t_gen, 4), "gen_GBps": round(gen_bytes / (t_gen * 1e-3) / 1e9, 1), "check_ms": round(t_chk, 4), "check_GBps": round(chk_bytes / (t_chk * 1e-3) / 1e9, 1),
                                        "check_frac": round(chk_bytes / (t_chk * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "check_bytes": chk_bytes}})
    else:
        out.update({"rounds_per_wavefront": kr,
                    "bytes_convention": f"resident arrays covered, each once: permutations x {24 // kr} chunks x (101 x {kr} + 25) arrays x 512 B x groups; round 4's line counted 126 arrays per "
                                        f"round (every midRound state twice) = {launch_bytes_r4} B for this launch",
                    "gen_kernel": {"kernel": "k_rounds_gen, alone (8 rounds per wavefront: 8 x 76 arrays written, midRound[r0] read)", "achieved": round(gen_bytes / (t_gen * 1e-3) / 1e9, 1), "avg_ms": round(t_gen, 4)}})
    out["check_pass"] = {"what": "whole pob_constraint_check over a resident vector (all G families + Keccak rounds + chains) as a pass of its own, alone, in-order schedule (one stream: the launches follow each other)",
                         "bytes": resident, "ms": round(t_check_pass, 3), "achieved": round(resident / (t_check_pass * 1e-3) / 1e9, 1), "frac": round(resident / (t_check_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    out["lone_batch"] = {"what": "ONE batch generated and evaluated by one calculator alone on the device (latency: its launches follow each other on one stream)", "ms": round(t_lone, 3)}
    out["step"] = {"what": "generate (write the resident vector once) + evaluate (read it once) per timed step", "bytes": 2 * resident,
                   "achieved": round(2 * resident / (ms_step * 1e-3) / 1e9, 1), "frac": round(2 * resident / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def leg_emission(calc0, info):
    """.wtns emission (the step after the path): the O0 payload and the reduced (O1-style) one, through the window pipeline into pinned memory"""
    cold_s, _ = calc0.emit_throughput(0, count=1)              # first emission of this calculator: allocates the windows (3 x 256 MiB device + pinned), probe pass
    sec, nbytes = calc0.emit_throughput(1, count=2)
    emission = {"what": "canonical 32 B/wire payload expanded on the GPU in 256 MiB windows, D2H double-buffered into pinned memory; steady state = 2 witnesses back to back "
                        "after a first one that set the buffers up (first_witness_ms includes hipHostMalloc of the windows and the probe pass)",
                "GB_per_s": round(nbytes / sec / 1e9, 2), "ms_per_witness": round(sec / 2 * 1e3, 1), "bytes_per_witness": nbytes // 2,
                "first_witness_ms": round(cold_s * 1e3, 1)}
    try:
        from proof_of_burn_amd.circuit_model import keepmap
        keep, _ = keepmap.load(MAIN)
        calc0.emit_throughput(0, count=3, keep=keep, window_wires=1 << 24)            # (the map's first use: hashed, pinned, its site tables built; two more to reach the steady state)
        rsec, rbytes = calc0.emit_throughput(3, count=4, keep=keep, window_wires=1 << 24)
        emission["reduced"] = {"what": "O1-style reduced witness (circuit_model/o1.py map, stored under circuit_model/data/): only the kept wires are expanded and copied "
                                       "(pob_emit_begin_reduced), 4 witnesses back to back after three others",
                               "kept_wires": int(keep.size), "of": int(info.n_witness), "ms_per_witness": round(rsec / 4 * 1e3, 2),
                               "GB_per_s": round(rbytes / rsec / 1e9, 2), "bytes_per_witness": rbytes // 4}
    except FileNotFoundError:
        pass
    return emission


def leg_single_witness_latency(job):
    """BASELINE config 2: ONE witness of the fixture (tests/test_pob_input.json, instantiation of tests/testcases/proof_of_burn.py:53) end to end:
    input.json text -> loader -> upload -> generate -> constraint evaluation -> verdict on the host -> the whole 2.06 GB .wtns payload in pinned memory"""
    from proof_of_burn_amd import WitnessCalculator, PinnedInputs
    W = job.W
    fix_json = os.path.join(ROOT, "tests", "golden", "test_pob_input.json")
    if not os.path.exists(fix_json):
        return None
    FIX = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"
    with open(fix_json, "rb") as f:
        text = f.read()
    one = WitnessCalculator(FIX, max_batch=1, device=job.dev)
    pin1 = PinnedInputs(one, 1)

    def once():
        ta = time.perf_counter()
        one.pack_json([text], threads=1, out=pin1)
        one.upload_pinned_async(pin1)
        one.generate(); one.constraint_check(); one.fetch_records()
        rec = one.wait_records()
        tb = time.perf_counter()
        assert rec["status"][0] == 0 and rec["check_status"][0] == W.CLEAN
        nb = sum(v.size for _, v in one.witness_windows(0))
        return tb - ta, time.perf_counter() - tb, nb, int.from_bytes(rec["commitment"][0].tobytes(), "little")
    once()                                            # (first call: buffers, probe pass)
    best = min((once() for _ in range(3)), key=lambda r: r[0] + r[1])
    pin1.free(); one.close()
    return {"what": "BASELINE config 2: one witness of tests/test_pob_input.json on " + FIX + ": input.json text -> verdict on the host (load, H2D, generate, evaluate, "
                    "records), then the whole canonical .wtns payload streamed into pinned host memory; best of 3 after one warm-up",
            "ms_to_verdict": round(best[0] * 1e3, 3), "ms_emit_payload": round(best[1] * 1e3, 2), "payload_bytes": int(best[2]),
            "ms_total": round((best[0] + best[1]) * 1e3, 2), "commitment": str(best[3])}


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    from proof_of_burn_amd import PinnedInputs, TextBatch, inputs as gen
    job = Job(args)
    torch, dist = job.torch, job.dist
    rank, world, B, GB = job.rank, job.world, job.B, job.GB
    MAIN_ = MAIN if args.main == "proof_of_burn" else "Spend(31)"
    INORDER = args.schedule == "inorder"
    if args.pipeline < 0:
        args.pipeline = DEFAULT_DEPTH if INORDER else 2
    PIPE = bool(args.pipeline)
    NC = (max(2, args.pipeline) if INORDER else 2) if PIPE else 1
    NB = 1 if args.dbg_no_upload else max(1, args.distinct_batches)

    # ---- synthetic inputs (seeded): global batch b holds witnesses [b*GB, (b+1)*GB) of the global sequence; witness g depends only on (seed, g)
    t0 = time.time()
    if args.main == "spend":
        batches = [gen.synthetic_spend_batch(B, first=b * GB + job.first0) for b in range(NB)]
    else:
        batches = [gen.synthetic_batch(B, depth=args.depth, seed=0xB0B, distinct_keys=args.distinct_keys, first=b * GB + job.first0,
                                       pow_device=job.dev if args.depth > 12 else None) for b in range(NB)]
    t_synth = (time.time() - t0) / NB
    loop = ServiceLoop(job, MAIN_, NC, INORDER, args.fused)
    calc0 = loop.calcs[0]
    info = calc0.info
    # ---- the loader: input.json texts -> packed rows in pinned memory, natively on the host cores (pob_pack_json_batch8); the Python packer beside it
    texts = [TextBatch([json.dumps(inp).encode() for inp in bt.inputs]) for bt in batches]
    pinned = [PinnedInputs(calc0, B) for _ in range(NB)]
    calc0.pack_json(texts[0], out=pinned[0])            # (first call: the loader pool's threads start, first touch of the texts)
    t0 = time.time()
    for rep in range(3):
        for b in range(NB):
            calc0.pack_json(texts[b], out=pinned[b])
    t_pack_native = (time.time() - t0) / (3 * NB)
    t0 = time.time()
    ref = calc0.pack(batches[0].inputs[:min(B, 128)])
    t_pack_py = (time.time() - t0) / min(B, 128) * B
    assert all(np.array_equal(x, y[:min(B, 128)]) for x, y in zip(ref, (pinned[0].fr, pinned[0].widened(), pinned[0].forced))), "native loader differs from the Python loader"
    expect = [_expect(np, bt) for bt in batches]
    loop.set_inputs(pinned, expect)

    # ---- setup: every calculator of the loop passes one batch (a calculator's first pass touches its buffers for the first time and creates its stream's hardware queue; with W
    #      warm-up steps fewer than calculators in flight some of them would do that inside the timed region)
    loop.run(NC)
    job.fence()
    # ---- the timed region: W untimed warm-up steps, fence, K timed steps, fence
    if args.warmup:
        loop.run(args.warmup)
    job.fence()
    if not args.probe_after and loop.fetch_each:
        loop.probe(True)
    loop.reset_counters()
    t0 = time.perf_counter()
    loop.run(args.steps, k0=args.warmup)
    job.fence()
    dt_mine = time.perf_counter() - t0
    assert loop.validated == args.steps * B, "not every batch was validated inside the timed region"
    host_ms = {k: round(v / args.steps * 1e3, 3) for k, v in loop.host_s.items()}
    validated_timed, h2d_per_step = loop.validated, loop.h2d_bytes // max(args.steps, 1)
    probe_steps = 0
    if args.probe_after and loop.fetch_each:
        # the dominant kernel as it runs IN the service loop: the same loop for a few more batches with HIP events around each of its launches
        loop.probe(True)
        probe_steps = 16
        loop.run(probe_steps, k0=args.warmup + args.steps)
        job.fence()
    kchk_in_step = float(np.mean(loop.kchk_ms)) if loop.kchk_ms else None
    loop.probe(False)
    dt = job.max_over_ranks(dt_mine)
    rank_ms = [s / args.steps * 1e3 for s in job.all_ranks(dt_mine)]
    last_k = args.warmup + args.steps + probe_steps - 1
    if world > 1:
        assert int(loop.gather_bad.item()) == 0, "a gathered record of some rank is not clean"
        rec_all = loop.last_gather.cpu()
        assert int(rec_all.shape[0]) == GB
        mine = rec_all[job.first0:job.first0 + B].numpy() if job.strong else rec_all[rank * B:(rank + 1) * B].numpy()
        assert np.array_equal(mine[:, 12:], expect[last_k % NB]), "gathered records differ from this rank's commitments"
        if rank == 0 and args.dump_results:
            np.save(args.dump_results, rec_all.numpy())
    elif rank == 0 and args.dump_results:
        np.save(args.dump_results, loop.recs[last_k % NC].cpu().numpy())
    ms_step = dt / args.steps * 1e3

    # ---- legs after the timed region, same run, same box
    e2e = bare = single = tracks_pipeline = None
    if PIPE and INORDER and not args.no_single and loop.fetch_each:
        e2e = leg_e2e_from_json(job, loop, texts, max(args.steps, 40), ms_step)
    if PIPE and not args.no_single and loop.fetch_each:
        bare = leg_kernel_pipeline_only(job, loop)
    if PIPE and not args.no_single:
        single, tracks_pipeline = legs_track_schedule(job, loop)
    depths, depth16, strong_slice, separate_eval = {}, None, None, None
    lone = rank == 0 and world == 1 and not job.strong and not args.no_extra_legs and loop.fetch_each and not args.shim and args.main == "proof_of_burn"
    if lone and INORDER and PIPE:
        for d in [int(x) for x in args.other_depths.split(",") if x.strip()]:
            if d != NC and d >= 2:
                depths[str(d)] = leg_other_depth(job, MAIN_, d, pinned, expect)
        if args.fused & 2:      # the same loop with the evaluation as a pass of its own over the resident vector (--fused 1: what the timed region did in the first half of round 6)
            separate_eval = leg_other_depth(job, MAIN_, NC, pinned, expect, fused=args.fused & 1,
                                            what=f"the same service loop ({NC} in flight, 96 steps) with the evaluation as a SEPARATE PASS over the resident vector (bench.py --fused {args.fused & 1}: "
                                                 "k_rounds_gen + k_rounds_check from HBM, the input check and the seven G-family launches of pob_constraint_check) instead of riding with the generation; a process's "
                                                 "later legs run 5-10 % slower than its timed region -- as a timed region of its own: bench.py --fused 1")
    if lone:
        # depth16: BASELINE config 5's shape on one GPU (16-layer proofs, byteSecurityRelax = 1, 3-zero-byte proof of work); strong_slice: BASELINE config 4 as one GPU sees it
        deep = [gen.synthetic_batch(B, depth=16, seed=0xD16, distinct_keys=args.distinct_keys, first=b * B, pow_device=job.dev) for b in range(2)]
        depth16 = leg_other_inputs(job, loop, texts, deep, max(40, args.steps),
                                   f"BASELINE config 5's shape on one GPU: batch={B} of 16-layer (max-depth) MPT proofs per step, same service loop and pinned buffers, 2 distinct batches cycled")
        sl = [gen.synthetic_batch(B, depth=args.depth, seed=0xB0B, distinct_keys=args.distinct_keys, first=b * 8 * B) for b in range(2)]
        strong_slice = leg_other_inputs(job, loop, texts, sl, max(40, args.steps),
                                        f"BASELINE config 4 as one GPU sees it: rank 0's slice (witnesses [0, {B})) of ONE global batch of {8 * B} split over 8 GPUs, per step; "
                                        f"the other ranks' slices and the all-gather of the 44-byte records need the node")

    ranks = {"rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1), "dist_backend": job.backend,
             "launched_by": "bench.py itself" if os.environ.get("POB_SELF_LAUNCHED") else ("a launcher (torch.distributed.run / the environment)" if world > 1 else "single process"),
             "ms_per_step_min": round(min(rank_ms), 3), "ms_per_step_max": round(max(rank_ms), 3)}
    if args.shim or args.main != "proof_of_burn":
        # the plumbing run of the tests: the loop, the slices and the gather have been exercised and validated; nothing is measured
        if rank == 0:
            print(json.dumps({"metric": "proof_of_burn witnesses/sec", "value": None, "unit": "witnesses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "scaling": "strong" if job.strong else "weak", "data": "synthetic", "shim": bool(args.shim), "main": MAIN_, "ranks": ranks,
                              "config": {"workload": f"TEST RUN (not a measurement): {MAIN_}, batch {B} per rank, global {GB}", "validated_witnesses": validated_timed,
                                         "h2d_bytes_per_step": int(h2d_per_step), "calculators_in_flight": NC, "bound_cpus": (len(job.bound_cpus) if job.bound_cpus else None),
                                         "loader_threads_env": os.environ.get("POB_LOADER_THREADS"), "dist_backend": job.backend}}), flush=True)
    else:
        roofline = leg_roofline(job, loop, info, kchk_in_step, ms_step, probe_steps)
        emission = leg_emission(calc0, info) if rank == 0 and not args.no_emission else None
        latency = leg_single_witness_latency(job) if rank == 0 and world == 1 and not args.no_emission else None
        cpu = cpu_baseline(batches[0], info, args.cpu_samples) if rank == 0 and world == 1 and not args.no_cpu_baseline else None
        if rank == 0:
            line = {
                "metric": "proof_of_burn witnesses/sec", "value": round(GB * args.steps / dt, 1), "unit": "witnesses/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "strong" if job.strong else "weak",
                "vs_baseline": None, "dtype": "u64 bit-sliced lanes + BN254 Fr (8x32-bit Montgomery)", "data": "synthetic",
                "config": {"workload": (f"global batch={GB} split over {world} GPUs" if job.strong else f"batch={B}/GPU") + f" proof_of_burn witnesses per step, {MAIN}, synthetic "
                                       f"{args.depth}-layer MPT proofs ({args.distinct_keys} distinct PoW burn keys tiled, {NB} distinct input batches cycled); per step, inside the timed "
                                       f"region: H2D of the packed inputs from pinned memory, generate, per-gate constraint evaluation, records {{status, verdict, commitment}} D2H, "
                                       f"every record validated on the host" + (", one all-gather of the records" if world > 1 else ""),
                           "wires_per_witness": int(info.n_witness), "resident_bytes_per_witness": int(info.group_bytes // 64),
                           "wire_classes": {"stored_bit": int(info.n_bit), "stored_sm": int(info.n_sm), "stored_fr": int(info.n_fr), "derived": int(info.n_derived),
                                            "alias": int(info.n_alias), "constant_one": 1,
                                            "what": "stored: resident in HBM (8 B per BIT wire, 256 B per SM wire, 2 KiB per FR wire, per 64 witnesses); derived: functions of stored wires, "
                                                    "rebuilt by the emitter; alias: Keccak round-block wires that ARE another stored wire (copy / rotated / negated / constant), expanded by the emitter"},
                           "full_layout_round3": {"resident_bytes_per_witness": 27225718, "ms_per_step": 11.257, "witnesses_per_s": 90963, "what": "BENCH_r03.json: every wire of the Keccak round blocks stored"},
                           "canonical_bytes_per_witness": int(info.n_witness) * 32,
                           "parallelism": f"one slice per GPU x{world}, " + ((f"{NC} in-order calculators in flight, one stream each, on consecutive batches of {B}" if INORDER else f"two linked track-schedule calculators pipelined over consecutive batches of {B}")
                                                                                         + " (fill and drain inside the timed region)" if PIPE else f"one calculator of {B} per GPU"),
                           "schedule": args.schedule, "calculators_in_flight": NC, "fused_launches": bool(args.fused & 1), "evaluation_rides_with_generation": bool(args.fused & 2) and INORDER,
                           "evaluation": ("every stored word of the Keccak round blocks, the input rows and the G units is loaded back (from L2) and compared with its definition by the launch that "
                                          "stores it; pob_constraint_check runs the sponge chains' and the RLP family's evaluation kernels and collects the records (DESIGN.md 4.1, 4.3)")
                                         if (args.fused & 2) and INORDER else "pob_constraint_check: a separate pass over the resident vector",
                           "setup": f"before the {args.warmup} warm-up steps every one of the {NC} calculators has passed one batch (first touch of its buffers, its stream's hardware queue)", "rank_bound_to_cpus": (len(job.bound_cpus) if job.bound_cpus else None),
                           "validated_witnesses": validated_timed, "h2d_bytes_per_step": int(h2d_per_step),
                           "rccl_ranks": ranks["rccl_ranks"], "dist_backend": job.backend,
                           "input_synthesis_s_per_batch": round(t_synth, 2), "host_ms_per_step": host_ms,
                           "json_to_packed_witnesses_per_s": round(B / max(t_pack_native, 1e-9), 1),
                           "json_to_packed": {"what": "input.json texts -> packed rows (byte form) in pinned memory, pob_pack_json_batch8 on the persistent loader pool (bit-equal to the Python loader on a sample of this batch)",
                                              "host_cores": os.cpu_count(), "python_loader_witnesses_per_s_one_core": round(B / max(t_pack_py, 1e-9), 1)}},
                "ranks": ranks,
                "roofline": roofline, "cpu_baseline": cpu, "emission": emission, "single_calculator": single, "tracks_pipeline": tracks_pipeline, "kernel_pipeline_only": bare, "single_witness_latency": latency,
                "depth16": depth16, "strong_slice": strong_slice, "other_depths": depths or None, "separate_evaluation_pass": separate_eval, "e2e_from_json": e2e,
            }
            print(json.dumps(line), flush=True)
    loop.close()
    for pin in pinned:
        pin.free()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
