"""N>1 plumbing on CPU: world_size-2 gloo run of the slice-per-rank sharding and the single result all-gather
(proof_of_burn_amd/distributed.py).  The GPU path uses the same functions with backend "nccl" (= RCCL)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from proof_of_burn_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything_once():
    for total in (0, 1, 7, 64, 1000, 8192):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


WORKER = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from proof_of_burn_amd import distributed as D
    rank, local_rank, world = D.init("gloo")
    assert world == 2
    # (a) hand-made records, EVEN slices, no `total`
    n = 5
    lo, hi = D.shard_bounds(2 * n, rank, world)
    status = torch.arange(lo, hi, dtype=torch.int32) * (rank + 1)
    outs = (torch.arange(n * 32, dtype=torch.int64).reshape(n, 32) + 100 * rank).to(torch.uint8)
    st, out = D.unpack_records(D.gather_records(D.pack_records(status, outs)))
    assert st.tolist() == [0, 1, 2, 3, 4, 10, 12, 14, 16, 18], st.tolist()
    assert out.shape == (10, 32) and out[0, 1].item() == 1 and out[5, 1].item() == 101
    # (b) UNEVEN slices (7 = 4 + 3): padded to the largest slice, trimmed after the single all-gather
    lo, hi = D.shard_bounds(7, rank, world)
    status = torch.arange(lo, hi, dtype=torch.int32) + 1000
    outs = torch.full((hi - lo, 32), rank + 7, dtype=torch.uint8)
    st, out = D.unpack_records(D.gather_records(D.pack_records(status, outs), total=7))
    assert st.tolist() == [1000 + i for i in range(7)] and out[:, 0].tolist() == [7, 7, 7, 7, 8, 8, 8]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT

# Two ranks, each running the REAL calculator (the product's kernels + host scheduler on the CPU shim of tests/hostsim) on its
# slice of one global Spend(31) batch, records packed by the library's own k_collect, ONE all-gather; the gathered job must
# equal a single-rank run of the whole batch.
CALC_WORKER = textwrap.dedent("""
    import ctypes, json, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from proof_of_burn_amd import distributed as D, witness as W
    W.LIB_PATH, W._lib = %r, None
    rank, local_rank, world = D.init("gloo")
    s = next(x for x in json.load(open(os.path.join(%r, "tests", "golden", "suites.json"))) if x["name"] == "test_spend")
    base = s["cases"][0]["input"]
    inputs = []
    for g in range(7):                                  # global batch of 7 (uneven: 4 + 3); witness 2 must fail (spend.circom:41)
        d = dict(base); d["extraCommitment"] = 1000 + g; d["withdrawnBalance"] = str(5 + g)
        if g == 2: d["withdrawnBalance"] = str(int(base["balance"]) + 1)
        inputs.append(d)
    def run(slice_inputs):
        calc = W.WitnessCalculator("Spend(31)", max_batch=len(slice_inputs))
        calc.calculate(slice_inputs)
        n = len(slice_inputs)
        buf = (ctypes.c_uint8 * (D.RECORD_BYTES * n)).from_address(calc.records_device_ptr())      # "device" memory of the shim
        rec = torch.from_numpy(np.ctypeslib.as_array(buf).reshape(n, D.RECORD_BYTES).copy())
        calc.close()
        return rec
    lo, hi = D.shard_bounds(len(inputs), rank, world)
    got = D.gather_records(run(inputs[lo:hi]), total=len(inputs))
    want = run(inputs)
    assert got.shape == (7, D.RECORD_BYTES) and torch.equal(got, want)
    st, out = D.unpack_records(got)
    assert (st != 0).tolist() == [g == 2 for g in range(7)]
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


# The job as bench.py runs it at N > 1: every rank pipelines TWO linked calculators (pob_set_partner) over consecutive global batches --
# evaluate batch k-1, generate batch k, gather batch k's records -- on its slices; every gathered batch must equal a lone calculator's
# run of the whole batch, and every evaluation must come back clean for the witnesses that are valid.
PIPE_WORKER = textwrap.dedent("""
    import ctypes, json, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from proof_of_burn_amd import distributed as D, witness as W
    W.LIB_PATH, W._lib = %r, None
    rank, local_rank, world = D.init("gloo")
    s = next(x for x in json.load(open(os.path.join(%r, "tests", "golden", "suites.json"))) if x["name"] == "test_spend")
    base = s["cases"][0]["input"]
    def batch(k):                                       # global batch k: 6 witnesses, witness k of it fails (spend.circom:41)
        out = []
        for g in range(6):
            d = dict(base); d["extraCommitment"] = 100 * k + g; d["withdrawnBalance"] = str(3 + g + k)
            if g == k: d["withdrawnBalance"] = str(int(base["balance"]) + 1)
            out.append(d)
        return out
    def records(calc, n):
        buf = (ctypes.c_uint8 * (D.RECORD_BYTES * n)).from_address(calc.records_device_ptr())
        return torch.from_numpy(np.ctypeslib.as_array(buf).reshape(n, D.RECORD_BYTES).copy())
    lo, hi = D.shard_bounds(6, rank, world)
    a, b = W.WitnessCalculator("Spend(31)", max_batch=hi - lo), W.WitnessCalculator("Spend(31)", max_batch=hi - lo)
    a.set_partner(b)
    calcs, prev, gathered, checked = [a, b], None, [], []
    for k in range(3):
        cur = calcs[k %% 2]
        if prev is not None:
            prev[0].constraint_check(); checked.append((prev[1], prev[0].results(with_check=True)))
        cur.upload_packed(*cur.pack(batch(k)[lo:hi])); cur.generate()
        cur.sync()
        gathered.append(D.gather_records(records(cur, hi - lo), total=6))
        prev = (cur, k)
    prev[0].constraint_check(); checked.append((prev[1], prev[0].results(with_check=True)))
    lone = W.WitnessCalculator("Spend(31)", max_batch=6)
    for k in range(3):
        lone.calculate(batch(k))
        assert torch.equal(gathered[k], records(lone, 6)), k
        st, _ = D.unpack_records(gathered[k])
        assert (st != 0).tolist() == [g == k for g in range(6)]
    for k, res in checked:
        for g, r in zip(range(lo, hi), res):
            assert r.ok == (g != k) and (not r.ok or (r.check_status == 0 and r.bad_wire is None)), (k, g, r)
    for c in (a, b, lone): c.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


# BASELINE config 4's split with a remainder: EIGHT ranks, ONE global batch of 8 191 Spend(31) witnesses (slices of 1 024 and 1 023), every rank a REAL
# calculator on the CPU shim (generate + evaluate its slice), records gathered with one all-gather.  Every rank checks the gathered job: its own slice
# is where shard_bounds says, the first and last witness of EVERY rank's slice equal a fresh two-witness calculation, the failing witnesses are exactly
# the ones built to fail, every valid witness evaluated clean.
WORLD8_WORKER = textwrap.dedent("""
    import ctypes, json, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from proof_of_burn_amd import distributed as D, witness as W
    W.LIB_PATH, W._lib = %r, None
    rank, local_rank, world = D.init("gloo")
    assert world == 8
    s = next(x for x in json.load(open(os.path.join(%r, "tests", "golden", "suites.json"))) if x["name"] == "test_spend")
    base = s["cases"][0]["input"]
    TOTAL = 8191
    def witness(g):
        d = dict(base); d["extraCommitment"] = 7 * g + 1; d["withdrawnBalance"] = str(1 + g %% 1000)
        if g %% 1021 == 5: d["withdrawnBalance"] = str(int(base["balance"]) + 1 + g)        # fails spend.circom:41
        return d
    def records(calc, n):
        buf = (ctypes.c_uint8 * (D.RECORD_BYTES * n)).from_address(calc.records_device_ptr())
        return torch.from_numpy(np.ctypeslib.as_array(buf).reshape(n, D.RECORD_BYTES).copy())
    lo, hi = D.shard_bounds(TOTAL, rank, world)
    assert hi - lo == (1024 if rank < 7 else 1023)
    calc = W.WitnessCalculator("Spend(31)", max_batch=hi - lo)
    calc.calculate([witness(g) for g in range(lo, hi)], check=True)
    mine = records(calc, hi - lo)
    calc.close()
    got = D.gather_records(mine, total=TOTAL)
    assert got.shape == (TOTAL, D.RECORD_BYTES) and torch.equal(got[lo:hi], mine)
    st, _ = D.unpack_records(got)
    cs, bw = D.unpack_verdicts(got)
    fails = [g %% 1021 == 5 for g in range(TOTAL)]
    assert (st != 0).tolist() == fails and sum(fails) == 9
    ok = torch.tensor([not f for f in fails])
    assert bool(((cs == D.CLEAN) & (bw == D.CLEAN))[ok].all())
    edge = W.WitnessCalculator("Spend(31)", max_batch=2)
    for r in range(world):
        a, b = D.shard_bounds(TOTAL, r, world)
        edge.calculate([witness(a), witness(b - 1)], check=True)
        e = records(edge, 2)
        assert torch.equal(got[a], e[0]) and torch.equal(got[b - 1], e[1]), r
    edge.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _run_two_ranks(script, world=2, timeout=170):
    port = D.free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


@pytest.mark.timeout(180)
def test_gather_records_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    _run_two_ranks(script)


@pytest.mark.timeout(240)
def test_two_ranks_run_the_calculator_and_gather(tmp_path):
    from tests.hostsim import build as hb
    lib = hb.build()
    script = tmp_path / "calc_worker.py"
    script.write_text(CALC_WORKER % (ROOT, lib, ROOT))
    _run_two_ranks(script)


@pytest.mark.timeout(300)
def test_two_ranks_pipeline_two_calculators_each(tmp_path):
    from tests.hostsim import build as hb
    lib = hb.build()
    script = tmp_path / "pipe_worker.py"
    script.write_text(PIPE_WORKER % (ROOT, lib, ROOT))
    _run_two_ranks(script)


@pytest.mark.timeout(600)
def test_eight_ranks_uneven_global_batch(tmp_path):
    from tests.hostsim import build as hb
    lib = hb.build()
    script = tmp_path / "world8_worker.py"
    script.write_text(WORLD8_WORKER % (ROOT, lib, ROOT))
    _run_two_ranks(script, world=8, timeout=560)


def _run_bench_ranks(world, extra, timeout=280):
    """bench.py as the driver launches it for N > 1 (one process per rank, RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_* in the environment), on the CPU shim"""
    port = D.free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), POB_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--shim", "--main", "spend", "--steps", "3", "--warmup", "1", "--pipeline", "4",
           "--no-single", "--no-emission", "--no-extra-legs", "--no-cpu-baseline"] + extra
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(world)]
    outs = [p.communicate(timeout=timeout) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, e[-2000:])
    lines = [json.loads(o.strip().splitlines()[-1]) for o, _ in outs[:1]]
    assert all(not [ln for ln in o.splitlines() if ln.strip() and not ln.startswith("[Gloo]")] for o, _ in outs[1:]), "only rank 0 prints the line"
    return lines[0]


@pytest.mark.timeout(600)
def test_bench_py_eight_ranks_on_the_shim_weak_and_strong():
    """`bench.py --gpus 8` end to end under gloo, every rank the real service loop (four in-order calculators over consecutive batches, pinned inputs through the native
    loader, byte-form upload, records validated per batch, ONE all-gather of the records per batch, the other ranks' records counted on arrival) on the CPU shim with
    Spend(31): weak (64 witnesses per rank) and BASELINE config 4's strong split with a remainder (one global batch of 509 witnesses: slices of 64 and 63); every
    rank binds itself to its share of the host's CPUs and the loader's width follows"""
    from tests.hostsim import build as hb
    hb.build()
    weak = _run_bench_ranks(8, ["--batch", "64"])
    assert weak["shim"] and weak["n_gpus"] == 8 and weak["scaling"] == "weak" and weak["config"]["validated_witnesses"] == 3 * 64
    assert weak["config"]["dist_backend"] == "gloo" and weak["config"]["calculators_in_flight"] == 4
    ncpu = len(os.sched_getaffinity(0))
    assert weak["config"]["bound_cpus"] == (ncpu // 8 if ncpu >= 8 else None)
    strong = _run_bench_ranks(8, ["--total-batch", "509"])
    assert strong["scaling"] == "strong" and strong["config"]["validated_witnesses"] == 3 * 64      # rank 0's slice of 509 = 64 (ranks 5..7: 63)
    assert "global 509" in strong["config"]["workload"]


@pytest.mark.timeout(600)
def test_bench_py_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 8 ...` with NO launcher and no WORLD_SIZE / RANK / MASTER_* in the environment (the way the driver starts the N = 1 bench): bench.py starts
    its eight ranks itself (bench.launch_ranks), rank 0 prints the ONE JSON line on the command's stdout, the records of all ranks are gathered per batch"""
    from tests.hostsim import build as hb
    hb.build()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["POB_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--shim", "--main", "spend", "--steps", "3", "--warmup", "1", "--batch", "64", "--pipeline", "4",
           "--no-single", "--no-emission", "--no-extra-legs", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks"]["rccl_ranks"] == 8 and d["ranks"]["dist_backend"] == "gloo" and d["ranks"]["launched_by"] == "bench.py itself"
    assert d["config"]["validated_witnesses"] == 3 * 64 and d["ranks"]["ms_per_step_min"] <= d["ranks"]["ms_per_step_max"]
    # a rank that dies takes the job down with its exit code instead of leaving seven ranks waiting at the rendezvous
    bad = subprocess.run(cmd + ["--total-batch", "3"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)      # 3 witnesses over 8 ranks: empty slices are refused
    assert bad.returncode != 0


def test_init_refuses_a_job_without_a_port(monkeypatch):
    """the ranks of one job must agree on the rendezvous port: it comes from the launcher (torch.distributed.run / MASTER_PORT), never from a default"""
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        D.init("gloo")
    a, b = D.free_port(), D.free_port()
    assert 1024 < a < 65536 and 1024 < b < 65536
