"""GPU parity tests (run on the MI355X: `pytest -m gpu`).  Everything goes through the C ABI (libpob_hip.so);
the CPU oracle is only the checker.  Bit-exact bar: public outputs, failure sets and the FULL canonical witness
payload (every one of the W 32-byte wires) must equal the oracle's."""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import evaluator_cases as EC
from tests import oracle_ffi as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POB_FIX = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"


def _suite(name):
    with open(os.path.join(ROOT, "tests", "golden", "suites.json")) as f:
        return next(s for s in json.load(f) if s["name"] == name)


@pytest.fixture(scope="module")
def pkg():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import proof_of_burn_amd
    return proof_of_burn_amd


def _first_diff(a, b):
    n = min(len(a), len(b))
    d = np.nonzero(a[:n].reshape(-1, 32) != b[:n].reshape(-1, 32))[0]
    return None if len(d) == 0 else int(d[0])


def test_reference_suite_spend(pkg, capsys):
    """the reference's own test tuple for Spend(31) (tests/testcases/spend.py:58-72) through the run() shim"""
    from proof_of_burn_amd.harness import run
    s = _suite("test_spend")
    got = run(s["main"], [(c["input"], c["expected"]) for c in s["cases"]])
    assert got == [c["expected"] for c in s["cases"]]


def test_reference_suite_proof_of_burn_through_run_shim(pkg):
    """the reference's own test tuple for ProofOfBurn (tests/testcases/proof_of_burn.py:52-76) through the run() shim"""
    from proof_of_burn_amd.harness import run
    s = _suite("test_proof_of_burn")
    got = run(s["main"], [(c["input"], c["expected"]) for c in s["cases"]])
    assert got == [c["expected"] for c in s["cases"]]


def test_spend_wtns_bit_exact(pkg, tmp_path):
    s = _suite("test_spend")
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=4)
    res = calc.calculate([c["input"] for c in s["cases"]], check=True)
    for case_i, (c, r) in enumerate(zip(s["cases"], res)):
        assert (r.outputs if r.ok else None) == c["expected"]
        if r.ok:
            assert r.check_status == 0 and r.bad_wire is None
            ora = O.run("Spend(31)", c["input"])
            gpu = calc.witness_payload(case_i)
            assert calc.nwitness == ora.nwitness == 2_603_360
            ref = ora.witness_numpy()
            assert np.array_equal(gpu, ref), f"first differing wire: {_first_diff(gpu, ref)}"
    # the .wtns file equals the oracle's writer byte for byte
    path = str(tmp_path / "witness.wtns")
    calc.write_wtns(0, path)
    O.run("Spend(31)", s["cases"][0]["input"])
    with open(path, "rb") as f:
        data = np.frombuffer(f.read(), dtype=np.uint8)
    assert np.array_equal(data, O.run("Spend(31)", s["cases"][0]["input"]).wtns_numpy())
    calc.close()


def test_reference_suite_proof_of_burn_and_wtns(pkg):
    """test_proof_of_burn (tests/testcases/proof_of_burn.py:52-76): fixture + 4 mutations in ONE batch (a failing lane
    must not poison the others), then the 2.06 GB payload of witness 0 and of the mutated-but-valid witness 3."""
    s = _suite("test_proof_of_burn")
    assert s["main"] == POB_FIX
    calc = pkg.WitnessCalculator(s["main"], max_batch=8)
    res = calc.calculate([c["input"] for c in s["cases"]], check=True)
    got = [r.outputs if r.ok else None for r in res]
    assert got == [c["expected"] for c in s["cases"]]
    assert [r.ok for r in res] == [True, False, False, True, True]
    for r in res:
        if r.ok:
            assert r.check_status == 0 and r.bad_wire is None, r
    assert calc.nwitness == 64_355_038
    for idx in (0, 3):
        ora = O.run(s["main"], s["cases"][idx]["input"])
        assert not ora.failed
        ref = ora.witness_numpy()
        gpu = calc.witness_payload(idx)
        assert gpu.shape == ref.shape
        assert np.array_equal(gpu, ref), f"witness {idx}: first differing wire {_first_diff(gpu, ref)}"
    calc.close()


def test_constraint_evaluator_detects_corruption(pkg):
    """flip one stored bit of one witness in (a) a G-region flag, (b) a Keccak round wire: the evaluator must flag
    exactly that witness, and report a wire inside the corrupted region."""
    s = _suite("test_spend")
    inp = [s["cases"][0]["input"]] * 3
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=3)
    res = calc.calculate(inp, check=True)
    assert all(r.ok and r.bad_wire is None for r in res)
    info = calc.info
    for bit_index in (100, int(info.n_bit) // 2, int(info.n_bit) - 5000):
        calc.lib.pob_debug_xor_bits(calc.h, 0, bit_index, 1 << 1)       # witness 1 only
        calc.constraint_check()
        res = calc.results(with_check=True)
        assert res[0].bad_wire is None and res[2].bad_wire is None
        assert res[1].bad_wire is not None, f"corruption at BIT index {bit_index} not detected"
        calc.lib.pob_debug_xor_bits(calc.h, 0, bit_index, 1 << 1)       # restore
    calc.constraint_check()
    assert all(r.bad_wire is None and r.check_status == 0 for r in calc.results(with_check=True))
    calc.close()


@pytest.mark.parametrize("which", ["spend", "pob"])
def test_constraint_evaluator_corruption_sweep(pkg, which):
    """flip one stored BIT wire of witness 1 at many positions (uniform over the BIT ranks, plus a dense sweep of the first
    ranks, which are G-unit wires): every flip must flag witness 1 and only witness 1.  Covers the lane-distributed run
    evaluation (selector rows, Pad/Num2Bits ranges, AssertByteString) and its per-wire attribution replay."""
    if which == "spend":
        s = _suite("test_spend"); main = "Spend(31)"
    else:
        s = _suite("test_proof_of_burn"); main = POB_FIX
    inp = [s["cases"][0]["input"]] * 3
    calc = pkg.WitnessCalculator(main, max_batch=3)
    res = calc.calculate(inp, check=True)
    assert all(r.ok and r.bad_wire is None and r.check_status == 0 for r in res)
    nbit = int(calc.info.n_bit)
    rng = np.random.default_rng(7)
    idxs = sorted(set(rng.integers(0, nbit, 600).tolist()) | set(range(0, min(nbit, 40000), 97)) | set(range(max(0, nbit - 30000), nbit, 61)))
    missed = []
    for bit_index in idxs:
        calc.lib.pob_debug_xor_bits(calc.h, 0, bit_index, 1 << 1)
        calc.constraint_check()
        r = calc.results(with_check=True)
        calc.lib.pob_debug_xor_bits(calc.h, 0, bit_index, 1 << 1)
        flagged = [x.bad_wire is not None or x.check_status != 0 for x in r]
        if flagged != [False, True, False]:
            missed.append((bit_index, flagged, r[1].bad_wire, r[1].check_status))
    calc.constraint_check()
    assert all(r.bad_wire is None and r.check_status == 0 for r in calc.results(with_check=True))
    calc.close()
    assert not missed, f"{len(missed)} of {len(idxs)} corruptions mis-detected, first: {missed[:5]}"


@pytest.mark.parametrize("which", ["spend", "pob"])
def test_constraint_evaluator_detects_sm_sb_fr_pokes(pkg, which):
    """the evaluator on every storage class, not only bits: >= 1000 SM, >= 300 FR (and 300 more BIT) uniformly drawn
    STORED values of a 64-witness group are corrupted, 63 lanes per pass (lane 0 = control); each pass must flag exactly the poked
    lanes.  SM covers the inputs, hashes, selector / shift outputs and the Divide quotient / remainder hints, FR the Poseidon states and
    the sub-string numbers (the operand wires of IsZero / IsEqual, M[], the copies and the Keccak round blocks' alias wires are not
    stored -- policy.hpp DV, keccak_kernels.hpp -- so there is nothing of them to corrupt).  Then the named wires, one at a time, with the
    reported wire checked."""
    if which == "spend":
        s = _suite("test_spend"); main = "Spend(31)"
        named = [("poseidon", k) for k in (1, 77, 200, 413, 640, 900)] + [("pad.div.out", 0), ("pad.div.rem", 0), ("commitment", 0)]
    else:
        s = _suite("test_proof_of_burn"); main = POB_FIX
        named = ([("poseidon", k) for k in (5, 300, 800)] + [("sc.exists", k) for k in (0, 5, 513)] +
                 [("pad.div.out", k) for k in (0, 1, 3)] + [("pad.div.rem", 2), ("commitment", 0)])
    calc = EC.open_identical_batch(pkg, main, s["cases"][0]["input"])
    missed, done = EC.uniform_sweep(calc, {EC.SM: 1200, EC.FR: 400, EC.BIT: 300})
    sizes = calc.class_sizes()
    assert done["SM"] >= min(1000, sizes[EC.SM] // 3) and done["FR"] >= min(300, sizes[EC.FR] // 3)      # (uniform draws with repetition; the derived wires have no storage)
    assert not missed, f"{len(missed)} mis-detections of {done}: {missed[:8]}"
    bad = EC.named_pokes(calc, named)
    assert not bad, bad
    calc.close()


def test_failed_witness_is_not_emitted(pkg, tmp_path):
    """like the reference binary (tests/test.py:65-68): an input that fails an assert produces no witness, also through the C ABI"""
    s = _suite("test_spend")
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=4)
    res = calc.calculate([c["input"] for c in s["cases"]])
    bad = next(i for i, r in enumerate(res) if not r.ok)
    with pytest.raises(RuntimeError, match="failed an assert"):
        calc.witness_payload(bad)
    with pytest.raises(RuntimeError, match="failed an assert"):
        calc.write_wtns(bad, str(tmp_path / "x.wtns"))
    assert not os.path.exists(tmp_path / "x.wtns")
    calc.close()


def test_calc_cli_is_the_reference_binary(pkg, tmp_path):
    """`python -m proof_of_burn_amd.calc spend input.json witness.wtns` = `./main_spend input.json witness.wtns` (reference
    Makefile:4-5): exit code, stderr only on failure (tests/test.py:65-68), no file on failure, file bytes = the oracle's"""
    import subprocess
    import sys
    s = _suite("test_spend")
    ok_case = next(c for c in s["cases"] if c["expected"] is not None)
    bad_case = next(c for c in s["cases"] if c["expected"] is None)
    for name, case, want_rc in (("ok", ok_case, 0), ("bad", bad_case, 1)):
        inp, out = tmp_path / f"{name}.json", tmp_path / f"{name}.wtns"
        inp.write_text(json.dumps(case["input"]))
        r = subprocess.run([sys.executable, "-m", "proof_of_burn_amd.calc", "spend", str(inp), str(out)], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == want_rc, r.stderr
        if want_rc == 0:
            assert r.stderr.strip() == "" or "amdgpu.ids" in r.stderr      # (the image's libdrm prints a missing-file note)
            ora = O.run("Spend(31)", case["input"])
            assert np.array_equal(np.frombuffer(out.read_bytes(), dtype=np.uint8), ora.wtns_numpy())
        else:
            assert "Failed assert in template" in r.stderr and not out.exists()
    # malformed input: missing key
    d = dict(ok_case["input"]); d.pop("balance")
    inp = tmp_path / "missing.json"; inp.write_text(json.dumps(d))
    r = subprocess.run([sys.executable, "-m", "proof_of_burn_amd.calc", "spend", str(inp), str(tmp_path / "m.wtns")], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "input error" in r.stderr


def test_main_instantiation_batch(pkg):
    """production parameters ProofOfBurn(16,4,16,50,31,2,1e19,1e20) (circuits/main_proof_of_burn.circom:27) on synthetic
    10-layer proofs: all valid, commitments equal the host-side formula, constraint evaluator clean, one .wtns vs oracle."""
    from proof_of_burn_amd import inputs as gen
    main = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
    batch = gen.synthetic_batch(8, depth=10, seed=0xB0B, distinct_keys=2)
    calc = pkg.WitnessCalculator(main, max_batch=8)
    res = calc.calculate(batch.inputs, check=True)
    assert calc.nwitness == 215_907_954
    for r, exp in zip(res, batch.commitments):
        assert r.ok, r.message()
        assert r.outputs == [exp]
        assert r.check_status == 0 and r.bad_wire is None
    ora = O.run(main, batch.inputs[5])
    assert not ora.failed and ora.outputs() == [batch.commitments[5]]
    gpu = calc.witness_payload(5)
    ref = ora.witness_numpy()
    assert np.array_equal(gpu, ref), f"first differing wire {_first_diff(gpu, ref)}"
    # the reduced (O1-style) witness, expanded and cut on the device: 21.5 M of the 215.9 M wires cross PCIe (two windows of kept wires)
    from proof_of_burn_amd.circuit_model import keepmap
    keep, nw = keepmap.load(main)
    assert nw == calc.nwitness
    red = calc.witness_payload_reduced(5, keep, window_wires=1 << 24)
    assert np.array_equal(red.reshape(-1, 32), ref.reshape(-1, 32)[keep]), "reduced payload differs from the oracle's kept wires"
    assert np.array_equal(calc.witness_payload(5), ref)          # and the O0 payload again (the per-window unit lists are per map)
    calc.close()


def test_reduced_witness_on_the_device(pkg, tmp_path):
    """SURVEY 8f-3 on the device (pob_emit_begin_reduced / pob_write_wtns_reduced): Spend(31) and the fixture instantiation, reduced payload ==
    the oracle's payload at the kept wires, through one window and through many; the .wtns header carries the kept count.
    (Which representative circom's own --O1 keeps is not pinned -- no circom here; the map is data, circuit_model/o1.py.)"""
    import struct
    from proof_of_burn_amd.circuit_model import keepmap
    for main, suite, case in (("Spend(31)", "test_spend", 0), (POB_FIX, "test_proof_of_burn", 3)):
        s = _suite(suite)
        keep, nw = keepmap.load(main)
        calc = pkg.WitnessCalculator(main, max_batch=4)
        res = calc.calculate([c["input"] for c in s["cases"]][:4] if main == "Spend(31)" else [s["cases"][case]["input"]] * 2)
        idx = case if main == "Spend(31)" else 1
        assert res[idx].ok and nw == calc.nwitness
        ref = O.run(main, s["cases"][case]["input"]).witness_numpy().reshape(-1, 32)[keep]
        assert np.array_equal(calc.witness_payload_reduced(idx, keep).reshape(-1, 32), ref)
        assert np.array_equal(calc.witness_payload_reduced(idx, keep, window_wires=len(keep) // 7 + 1).reshape(-1, 32), ref)
        path = str(tmp_path / "red.wtns")
        calc.write_wtns_reduced(idx, path, keep)
        data = np.fromfile(path, dtype=np.uint8)
        assert struct.unpack("<I", data[60:64].tobytes())[0] == len(keep) and np.array_equal(data[76:].reshape(-1, 32), ref)
        with pytest.raises(RuntimeError):
            calc.witness_payload_reduced(idx, keep[1:])           # wire 0 must be kept
        calc.close()


def _mutations(base, rng):
    """(label, mutated input) pairs covering every assert / === of proof_of_burn.circom and the loader's edge cases"""
    import copy
    P = O.P
    out = [("valid", base)]

    def mut(label, fn):
        d = copy.deepcopy(base)
        fn(d)
        out.append((label, d))

    L = len(base["layers"])
    nl = base["numLayers"]
    mut("intended>actual", lambda d: d.update(intendedBalance=str(int(d["actualBalance"]) + 1)))
    mut("actual>max", lambda d: d.update(actualBalance=str(10 ** 19 + 1)))
    mut("reveal>intended", lambda d: d.update(revealAmount=str(int(d["intendedBalance"]) + 1)))
    mut("balance 2^248", lambda d: d.update(actualBalance=str(2 ** 248)))
    mut("nibbles-1", lambda d: d.update(numLeafAddressNibbles=str(int(d["numLeafAddressNibbles"]) - 1)))
    mut("nibbles+1", lambda d: d.update(numLeafAddressNibbles=str(int(d["numLeafAddressNibbles"]) + 1)))
    mut("relax=1", lambda d: d.update(byteSecurityRelax=1))
    mut("relax huge", lambda d: d.update(byteSecurityRelax=str((P + 1) // 2)))
    mut("burnKey+1", lambda d: d.update(burnKey=str(int(d["burnKey"]) + 1)))
    mut("extra+1", lambda d: d.update(burnExtraCommitment=str((int(d["burnExtraCommitment"]) + 1) % P)))
    mut("proofExtra (valid)", lambda d: d.update(_proofExtraCommitment=str(rng.randrange(P))))
    for k in (0, 40, base["layerLens"][0] - 1):
        mut(f"layer0[{k}]^1", lambda d, k=k: d["layers"][0].__setitem__(k, d["layers"][0][k] ^ 1))
    for k in (0, 5, 70):
        mut(f"leaf[{k}]^1", lambda d, k=k: d["layers"][nl - 1].__setitem__(k, d["layers"][nl - 1][k] ^ 1))
    mut("layer0 pad byte (valid: beyond layerLen)", lambda d: d["layers"][0].__setitem__(d["layerLens"][0] + 3, 7))
    mut("layerLens0-1", lambda d: d["layerLens"].__setitem__(0, d["layerLens"][0] - 1))
    mut("layerLens0+1", lambda d: d["layerLens"].__setitem__(0, d["layerLens"][0] + 1))
    mut("numLayers-1", lambda d: d.update(numLayers=nl - 1))
    mut("numLayers+1", lambda d: d.update(numLayers=nl + 1))
    mut("numLayers=0", lambda d: d.update(numLayers=0))
    mut("numLayers=L+1", lambda d: d.update(numLayers=L + 1))
    mut("numLayers=2^40", lambda d: d.update(numLayers=2 ** 40))
    mut("numLayers=p-1", lambda d: d.update(numLayers=str(P - 1)))
    mut("header stateRoot^1", lambda d: d["blockHeader"].__setitem__(100, d["blockHeader"][100] ^ 1))
    mut("header other byte (valid, new commitment)", lambda d: d["blockHeader"].__setitem__(10, d["blockHeader"][10] ^ 1))
    mut("headerLen-1 (valid, new commitment)", lambda d: d.update(blockHeaderLen=d["blockHeaderLen"] - 1))
    mut("headerLen=max", lambda d: d.update(blockHeaderLen=len(d["blockHeader"])))
    mut("byte=256", lambda d: d["layers"][0].__setitem__(3, 256))
    mut("unused layer byte=256", lambda d: d["layers"][L - 1].__setitem__(0, 256))
    mut("unused layer byte changed (valid)", lambda d: d["layers"][L - 1].__setitem__(0, 99))
    mut("unused layerLens=0", lambda d: d["layerLens"].__setitem__(L - 1, 0))
    mut("unused layerLens=30", lambda d: d["layerLens"].__setitem__(L - 1, 30))
    mut("unused layerLens=31 (valid)", lambda d: d["layerLens"].__setitem__(L - 1, 31))
    mut("unused layerLens=543 (valid)", lambda d: d["layerLens"].__setitem__(L - 1, len(d["layers"][0]) - 1))
    mut("unused layerLens=544", lambda d: d["layerLens"].__setitem__(L - 1, len(d["layers"][0])))
    mut("layerLens=p-5", lambda d: d["layerLens"].__setitem__(L - 1, str(P - 5)))
    mut("scalar as 1-element array (valid)", lambda d: d.update(numLayers=[nl]))
    mut("hex string (valid)", lambda d: d.update(revealAmount=hex(int(d["revealAmount"]))))
    return out


def _assert_payload(calc, idx, main, inp, brief, buf, label):
    """the whole canonical payload of witness idx equals the oracle's: the GPU's bytes (emitted into the caller's reused buffer) against the digest a worker process took of the
    oracle's (tests/oracle_ffi.py OraclePool); on a mismatch the oracle runs again, here, to name the first differing wire"""
    assert brief.nwitness == calc.nwitness, label
    gpu = calc.witness_payload(idx, out=buf)
    if O.payload_digest(gpu) != brief.digest:
        ref = O.run(main, inp).witness_numpy()
        assert np.array_equal(gpu, ref), f"{label}: first differing wire {_first_diff(gpu, ref)}"
        raise AssertionError(f"{label}: the digests differ but the payloads do not")


@pytest.mark.parametrize("depth", [2, 4])
def test_failure_sets_match_oracle(pkg, depth):
    """one GPU batch of ~40 mutated inputs (a failing lane must not disturb its neighbours): pass/fail and the public output
    must equal the oracle's for every one of them -- every assert / === site of the circuit is hit by some mutation"""
    import random
    from proof_of_burn_amd import inputs as gen
    params = (4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)
    base = gen.synthetic_batch(1, depth=depth, seed=99 + depth, distinct_keys=1, params=params).inputs[0]
    cases = _mutations(base, random.Random(5))
    with O.OraclePool() as pool:                  # the oracle's ~40 runs in worker processes, beside the GPU's batch
        jobs = pool.submit(POB_FIX, [c[1] for c in cases], digest=True)
        calc = pkg.WitnessCalculator(POB_FIX, max_batch=len(cases))
        res = calc.calculate([c[1] for c in cases], check=True)
        oras = pool.collect(jobs)
    n_fail = 0
    buf = np.empty(32 * calc.nwitness, dtype=np.uint8)
    for i, ((label, inp), r, ora) in enumerate(zip(cases, res, oras)):
        exp = None if ora.failed else ora.outputs()
        got = r.outputs if r.ok else None
        assert got == exp, f"{label}: GPU {got} vs oracle {exp} ({r.message()})"
        n_fail += exp is None
        if r.ok:
            assert r.check_status == 0 and r.bad_wire is None, label
            _assert_payload(calc, i, POB_FIX, inp, ora, buf, label)      # every valid mutation: the whole payload, not only the output
    assert exp is not None or n_fail > 0
    assert 15 < n_fail < len(cases) - 5          # the mutation set really exercises both outcomes
    calc.close()


def test_failure_sets_match_oracle_at_the_production_instantiation(pkg):
    """reference tests/testcases/proof_of_burn.py:40-76 at the instantiation of circuits/main_proof_of_burn.circom:27: the ~45 mutations of _mutations on
    ProofOfBurn(16,4,16,50,31,2,1e19,1e20), scattered over the three groups of ONE 130-witness batch of an IN-ORDER calculator (the schedule bench.py runs; a failing lane
    must not disturb its neighbours): the fail set and the outputs equal the oracle's, the status of a failing witness is the lowest failing site (never above the
    oracle's first one, equal to the track schedule's), every witness that passes evaluates clean, and the whole 6.9 GB payload of every VALID mutation is the oracle's"""
    import random
    import re
    from proof_of_burn_amd import inputs as gen, witness as W
    n = 130
    batch = gen.synthetic_batch(n, depth=10, seed=0x5EED, distinct_keys=2)
    cases = _mutations(batch.inputs[0], random.Random(11))
    assert 40 <= len(cases) < n
    inputs = list(batch.inputs)
    pos = [(3 + 37 * i) % n for i in range(len(cases))]          # 37 is coprime to 130: distinct positions in all three groups
    assert len(set(pos)) == len(cases)
    for (label, inp), q in zip(cases, pos):
        inputs[q] = inp
    pool = O.OraclePool()                                        # the oracle's 46 production runs (2 s, 6.9 GB each) in worker processes, beside the GPU's work
    jobs = pool.submit(PROD, [c[1] for c in cases], digest=True)
    calc = pkg.WitnessCalculator(PROD, max_batch=n)
    calc.set_inorder(7)                                          # in order, fused launch, the evaluation riding with the generation: the schedule bench.py runs
    res = calc.calculate(inputs, check=True)
    tracks = pkg.WitnessCalculator(PROD, max_batch=n)
    res_t = tracks.calculate(inputs, check=True)
    tracks.close()
    # ... whose records -- the evaluator's verdict included -- are those of the track schedule's separate evaluation pass, valid and failing witnesses alike
    assert [(r.status, r.outputs, r.check_status, r.bad_wire) for r in res] == [(r.status, r.outputs, r.check_status, r.bad_wire) for r in res_t], "the two schedules disagree about a witness"
    code_of = {name: tid for tid, name in W._TPL.items()}
    n_fail = n_valid_payloads = n_status_equal = 0
    is_case = set(pos)
    for q in range(n):                                            # the untouched neighbours
        if q not in is_case:
            assert res[q].ok and res[q].outputs == [batch.commitments[q]] and res[q].check_status == 0 and res[q].bad_wire is None, q
    oras = pool.collect(jobs); pool.close()
    buf = np.empty(32 * calc.nwitness, dtype=np.uint8)
    for ((label, inp), q), ora in zip(zip(cases, pos), oras):
        r = res[q]
        exp = None if ora.failed else ora.outputs()
        assert (r.outputs if r.ok else None) == exp, f"{label}: GPU {r.outputs if r.ok else r.message()} vs oracle {exp if exp else ora.msg}"
        if exp is None:
            n_fail += 1
            m = re.match(r"Failed assert in template (\w+) line (\d+)", ora.msg)
            if m and m.group(1) in code_of and r.status != W.FAIL_INPUT_RANGE:      # (an input that does not fit its int32 row fails up front, whatever the circuit would say)
                # the emitted calculator stops at the FIRST failing site in execution order (ora.msg), the device reports the LOWEST failing site code of the witness:
                # never above the first one -- and EQUAL to the lowest of all the sites the oracle, which keeps going, has seen fail (round 6: oracle_fail_sites)
                assert r.status <= ((code_of[m.group(1)] << 12) | int(m.group(2))), f"{label}: {r.message()} vs oracle {ora.msg}"
                if all(t in code_of for t, _ in ora.sites):
                    assert r.status == min((code_of[t] << 12) | ln for t, ln in ora.sites), f"{label}: {r.message()} vs the oracle's failing sites {ora.sites}"
                    n_status_equal += 1
        else:
            assert r.check_status == 0 and r.bad_wire is None, label
            _assert_payload(calc, q, PROD, inp, ora, buf, label)
            n_valid_payloads += 1
    assert n_fail > 15 and n_valid_payloads >= 8 and n_status_equal > 10, (n_fail, n_valid_payloads, n_status_equal)
    calc.close()


def test_max_depth_and_ragged_batches(pkg):
    """BASELINE config 5 shape: 16-layer proofs (byteSecurityRelax = 1, 3-zero-byte PoW) on the production instantiation;
    batch sizes that are not multiples of the 64-witness group; batch of one."""
    from proof_of_burn_amd import inputs as gen
    main = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
    deep = gen.synthetic_batch(3, depth=16, seed=4242, distinct_keys=1, pow_device=0)      # (the 3-zero-byte proof of work on the GPU: 2^24 hashes, the same key the host search finds -- test_pow_search_gpu_matches_host)
    shallow = gen.synthetic_batch(67, depth=8, seed=77, distinct_keys=2)
    calc = pkg.WitnessCalculator(main, max_batch=70)
    for batch in (deep.inputs + shallow.inputs, shallow.inputs[:1], shallow.inputs[:65]):
        res = calc.calculate(batch)
        exp = (deep.commitments + shallow.commitments) if len(batch) == 70 else shallow.commitments[:len(batch)]
        assert [r.outputs for r in res] == [[c] for c in exp]
    calc.close()


def _run_bench(args, env_extra=None, nproc=1, timeout=900):
    import subprocess
    import sys
    from proof_of_burn_amd import distributed as D
    env = dict(os.environ, **(env_extra or {}))
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        port = D.free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = next(ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{"))
    return json.loads(line)


def test_two_calculators_pipelined_over_consecutive_batches(pkg):
    """pob_set_partner on the GPU: the same three-batch scenario as on the CPU shim (tests/test_hostsim_cpu.py) -- there the streams are
    synchronous, here the event gating between the two handles is real"""
    from tests.test_hostsim_cpu import test_two_calculators_pipelined_over_consecutive_batches as scenario
    scenario(pkg)


def test_two_ranks_on_one_gpu_equal_a_single_rank_run(pkg, tmp_path):
    """BASELINE config 4's shape on the hardware at hand: bench.py --gpus 2 as two ranks (own process, own handle, own slice of the
    global batch) sharing GPU 0, result records through ONE all-gather (gloo here, RCCL on a node); the gathered 1024 records must
    equal a single-rank run of the same 1024 seeds (the two ranks with bench.py's default four in-order calculators in flight each, the
    single rank with one calculator).  bench.py itself asserts validity, commitments and a clean evaluator per rank."""
    a, b = str(tmp_path / "two.npy"), str(tmp_path / "one.npy")
    quick = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-emission", "--no-single", "--no-extra-legs"]
    two = _run_bench(["--gpus", "2", "--batch", "512", "--dump-results", a] + quick, {"POB_FORCE_DEVICE": "0", "POB_DIST_BACKEND": "gloo"}, nproc=2)
    one = _run_bench(["--gpus", "1", "--batch", "1024", "--pipeline", "0", "--dump-results", b] + quick)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak" and two["config"]["rccl_ranks"] == 2
    assert two["config"]["validated_witnesses"] == 3 * 512 and one["config"]["validated_witnesses"] == 3 * 1024
    ra, rb = np.load(a), np.load(b)
    assert ra.shape == rb.shape == (1024, 44) and np.array_equal(ra, rb)
    assert not ra[:, :4].any() and (ra[:, 4:12] == 0xFF).all()       # every status 0, every verdict clean
    # BASELINE config 4 as written: ONE global batch split over the ranks (strong scaling), uneven split included
    c = str(tmp_path / "strong.npy")
    # ... started WITHOUT a launcher: `python bench.py --gpus 2 ...` starts its two ranks itself (bench.launch_ranks), rank 0 prints the one line
    env_clean = {k: "" for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT") if k in os.environ}
    assert not env_clean, "the test process itself runs under a launcher"
    st = _run_bench(["--gpus", "2", "--total-batch", "1023", "--dump-results", c] + quick, {"POB_FORCE_DEVICE": "0", "POB_DIST_BACKEND": "gloo"}, nproc=1)
    assert st["n_gpus"] == 2 and st["ranks"]["launched_by"] == "bench.py itself" and st["ranks"]["rccl_ranks"] == 2
    rc = np.load(c)
    assert st["scaling"] == "strong" and rc.shape == (1023, 44) and not rc[:, :4].any() and (rc[:, 4:12] == 0xFF).all()
    assert st["config"]["validated_witnesses"] == 3 * 512 and len({bytes(x) for x in rc[:, 12:]}) == 1023       # (rank 0's slice: 512 of the 1023; all commitments differ)


def test_max_depth_config5_payload_and_bench(pkg):
    """BASELINE config 5's shape: 16-layer proofs (byteSecurityRelax = 1, 3-zero-byte proof of work found by the HIP search kernel)
    on the production instantiation: evaluator clean, one full 6.9 GB witness payload bit-exact vs the oracle, and bench.py --depth 16."""
    from proof_of_burn_amd import inputs as gen
    main = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
    deep = gen.synthetic_batch(3, depth=16, seed=4242, distinct_keys=2, pow_device=0)
    calc = pkg.WitnessCalculator(main, max_batch=3)
    res = calc.calculate(deep.inputs, check=True)
    for r, exp in zip(res, deep.commitments):
        assert r.ok and r.outputs == [exp] and r.check_status == 0 and r.bad_wire is None, r
    ora = O.run(main, deep.inputs[2])
    assert not ora.failed and ora.outputs() == [deep.commitments[2]]
    gpu = calc.witness_payload(2)
    ref = ora.witness_numpy()
    assert np.array_equal(gpu, ref), f"first differing wire {_first_diff(gpu, ref)}"
    # a 16-layer proof through the input producer (reference tests/main.py:65-178): the unpadded nodes and the header, as eth_getProof and the block RPC return them,
    # through from_account_proof -- the leaf's hex-prefix parse, the state-root position, the padding -- give the same input.json, and that one the same witness
    inp = deep.inputs[1]
    nodes = [bytes(inp["layers"][i][:inp["layerLens"][i]]) for i in range(inp["numLayers"])]
    assert len(nodes) == 16
    again = gen.from_account_proof(nodes, bytes(inp["blockHeader"][:inp["blockHeaderLen"]]), int(inp["burnKey"]), int(inp["actualBalance"]), int(inp["intendedBalance"]),
                                   int(inp["revealAmount"]), int(inp["burnExtraCommitment"]), int(inp["_proofExtraCommitment"]), inp["byteSecurityRelax"])
    assert {k: (v if isinstance(v, list) else int(v)) for k, v in again.items()} == {k: (v if isinstance(v, list) else int(v)) for k, v in inp.items()}
    r = calc.calculate([again], check=True)[0]
    assert r.ok and r.outputs == [deep.commitments[1]] and r.check_status == 0 and r.bad_wire is None
    calc.close()
    line = _run_bench(["--gpus", "1", "--depth", "16", "--batch", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-emission", "--no-single", "--no-extra-legs", "--pipeline", "4",
                       "--distinct-keys", "2", "--distinct-batches", "2"])
    assert "16-layer" in line["config"]["workload"] and line["value"] > 0


def test_gpu_witness_satisfies_the_independent_r1cs(pkg):
    """the referee (proof_of_burn_amd/circuit_model: the circuits restated as A*B = C rows over its own wire numbering) accepts the
    GPU's witness of the fixture instantiation (64.4 M rows) and of Spend(31), and rejects a witness emitted from a poked resident
    vector exactly at rows that touch the poked wire"""
    from proof_of_burn_amd.circuit_model import check as CK, circuit
    sp = _suite("test_spend")
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=1)
    assert calc.calculate(sp["cases"][0]["input"])[0].ok
    assert CK.check_witness(circuit("Spend(31)"), CK.Witness(calc.witness_payload(0))) == []
    calc.close()
    s = _suite("test_proof_of_burn")
    c = circuit(POB_FIX)
    calc = pkg.WitnessCalculator(POB_FIX, max_batch=1)
    assert calc.calculate(s["cases"][0]["input"])[0].ok
    assert CK.check_witness(c, CK.Witness(calc.witness_payload(0))) == []
    for name, k in (("poseidon", 300), ("sc.exists", 5), ("pad.div.out", 1)):
        cls, idx, wire = calc.debug_ref(name, k)
        mask = 1 if cls == 0 else 4                      # (BIT class: the bit itself)
        calc.poke(cls, idx, 0, mask)
        bad = CK.check_witness(c, CK.Witness(calc.witness_payload(0)))
        calc.poke(cls, idx, 0, mask)
        assert bad and all(wire in wires for _, wires in bad), (name, wire, bad[:3])
    calc.close()
    # (the production instantiation's 215.9 M rows: test_production_gpu_witness_satisfies_the_independent_r1cs)


PROD = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"


def test_production_batch_payloads_beyond_group_0(pkg):
    """ONE 1 024-witness production batch (BASELINE config 3's shape): the O0 payload and the reduced payload of witnesses 64, 511 and 1 023 --
    groups 1, 7 and 15: emission addresses the resident slab by idx / 64 and the lane by idx % 64 -- equal the oracle's, every record of the
    batch is clean, and the emitter's inverse paths that ran are counted (the field-element inverses of SubstringCheck's IsEqual(exists)
    operands run at production parameters; the Fermat path needs a small operand beyond +-4096, which no valid production witness has:
    tests/test_hostsim_cpu.py drives it through a Selector(6000) main)"""
    from proof_of_burn_amd import inputs as gen
    from proof_of_burn_amd.circuit_model import keepmap
    batch = gen.synthetic_batch(1024, depth=10, seed=0xB0B, distinct_keys=16)
    calc = pkg.WitnessCalculator(PROD, max_batch=1024)
    res = calc.calculate(batch.inputs, check=True)
    assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in res) and [r.outputs[0] for r in res] == batch.commitments
    keep, nw = keepmap.load(PROD)
    assert nw == calc.nwitness
    calc.emit_counters()
    for idx in (64, 511, 1023):
        ora = O.run(PROD, batch.inputs[idx])
        assert not ora.failed and ora.outputs() == [batch.commitments[idx]]
        ref = ora.witness_numpy()
        gpu = calc.witness_payload(idx)
        assert np.array_equal(gpu, ref), f"witness {idx}: first differing wire {_first_diff(gpu, ref)}"
        del gpu
        red = calc.witness_payload_reduced(idx, keep, window_wires=1 << 24)
        assert np.array_equal(red.reshape(-1, 32), ref.reshape(-1, 32)[keep]), f"witness {idx}: reduced payload differs from the oracle's kept wires"
        del red, ref, ora
    c = calc.emit_counters()
    assert c["table"] > 0 and c["field_nonzero"] >= 3 * 7000 and c["field_zero"] > 0 and c["fermat"] == 0, c
    print("emitter inverse paths over 3 O0 + 3 reduced production payloads:", c)
    calc.close()


@pytest.mark.parametrize("depth", [2, 8, 9, 10, 15, 16])
def test_production_payload_across_depths(pkg, depth):
    """production instantiation, proofs of depth 2 / 8 / 9 / 10 / 15 / 16 (the derived and alias wires are data-dependent): the whole O0 payload of a
    witness of group 1 against the oracle"""
    from proof_of_burn_amd import inputs as gen
    n = 66
    batch = gen.synthetic_batch(n, depth=depth, seed=1000 + depth, distinct_keys=2, pow_device=0 if depth > 12 else None)
    calc = pkg.WitnessCalculator(PROD, max_batch=n)
    res = calc.calculate(batch.inputs, check=True)
    assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in res) and [r.outputs[0] for r in res] == batch.commitments
    idx = n - 1
    ora = O.run(PROD, batch.inputs[idx])
    assert not ora.failed and ora.outputs() == [batch.commitments[idx]]
    gpu, ref = calc.witness_payload(idx), ora.witness_numpy()
    assert np.array_equal(gpu, ref), f"depth {depth}: first differing wire {_first_diff(gpu, ref)}"
    calc.close()


def test_production_gpu_witness_satisfies_the_independent_r1cs(pkg):
    """the referee on HEAD: all 215 962 292 rows of the production instantiation (proof_of_burn_amd/circuit_model, which shares no code with
    the generators) hold on a GPU-emitted witness of group 1 of a synthetic 10-layer batch"""
    from proof_of_burn_amd import inputs as gen
    from proof_of_burn_amd.circuit_model import check as CK, circuit
    n = 70
    batch = gen.synthetic_batch(n, depth=10, seed=0xC0FFEE, distinct_keys=2)
    calc = pkg.WitnessCalculator(PROD, max_batch=n)
    res = calc.calculate(batch.inputs, check=True)
    assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in res)
    payload = calc.witness_payload(n - 1)
    calc.close()
    c = circuit(PROD)
    assert c.n_wires == 215_907_954 and c.n_constraints == 215_962_292
    assert CK.check_witness(c, CK.Witness(payload)) == []


def test_emit_selfcheck_on_written_values(pkg):
    """pob_emit_selfcheck on the production instantiation: the derived wires' own relations (IsZero, IsEqual, SubstringCheck's M[] recurrence)
    evaluated on the values WRITTEN into the emission windows -- clean for a valid witness of group 1 (payload == the oracle's), and a witness
    emitted from a resident vector whose KeccakBytes.inLen operand was poked violates the relations of the derived IsEqual([i, inLen]) wires
    that consume it: the check names such a wire.  (tests/test_hostsim_cpu.py runs the same on the fixture instantiation.)"""
    from proof_of_burn_amd import inputs as gen
    n = 66
    batch = gen.synthetic_batch(n, depth=10, seed=0x5E1F, distinct_keys=2)
    calc = pkg.WitnessCalculator(PROD, max_batch=n)
    assert all(r.ok and r.check_status == 0 for r in calc.calculate(batch.inputs, check=True))
    ref = O.run(PROD, batch.inputs[n - 1]).witness_numpy()
    calc.emit_selfcheck(True)
    assert np.array_equal(calc.witness_payload(n - 1), ref)
    r = calc.emit_selfcheck_result()
    assert r["first_bad_wire"] is None and r["checked"] > 200_000 and r["skipped"] < 100, r
    print("self-check of one production witness:", r)
    for kb in (0, 3):
        cls, idx, wire = calc.debug_ref("kb.inLen", kb)
        calc.poke(cls, idx, 1, 1, group=1)                 # lane 1 of group 1 = witness 65 = n - 1
        got = calc.witness_payload(n - 1)
        r = calc.emit_selfcheck_result()
        calc.poke(cls, idx, 1, 1, group=1)
        assert not np.array_equal(got, ref) and r["first_bad_wire"] is not None and wire < r["first_bad_wire"] < wire + 200_000, (kb, wire, r)
    assert np.array_equal(calc.witness_payload(n - 1), ref) and calc.emit_selfcheck_result()["first_bad_wire"] is None
    # the reduced form of the same witness (the stored keep map, no class representatives given: the sites whose own wires all survive)
    from proof_of_burn_amd.circuit_model import keepmap
    keep, _ = keepmap.load(PROD)
    red = calc.witness_payload_reduced(n - 1, keep)
    r = calc.emit_selfcheck_result()
    assert np.array_equal(red.reshape(-1, 32), ref.reshape(-1, 32)[keep]) and r["first_bad_wire"] is None, r
    print("self-check of the reduced production witness without an alias map:", r)
    calc.close()


def test_emit_selfcheck_of_the_reduced_witness(pkg):
    """the self-check on what a prover built at circom's default level consumes (.github/workflows/circuitscan.yml:29,36): the fixture instantiation's reduced witness with
    the class representatives of circuit_model/o1.py -- every IsZero / IsEqual site is evaluated on representatives and constants, clean for a valid witness; a poked
    KeccakBytes.inLen is caught in the reduced payload too"""
    from proof_of_burn_amd.circuit_model.o1 import reduce_map
    from proof_of_burn_amd.circuit_model.circuits import circuit
    with open(os.path.join(ROOT, "tests", "golden", "test_pob_input.json")) as f:
        inp = json.load(f)
    m = reduce_map(circuit(POB_FIX))
    calc = pkg.WitnessCalculator(POB_FIX, max_batch=66)
    assert all(r.ok and r.check_status == 0 for r in calc.calculate([inp] * 66, check=True))
    full = O.run(POB_FIX, inp).witness_numpy().reshape(-1, 32)
    calc.emit_selfcheck(True)
    calc.emit_selfcheck_alias(m)
    red = calc.witness_payload_reduced(65, m)
    r = calc.emit_selfcheck_result()
    assert np.array_equal(red.reshape(-1, 32), full[m.keep]) and r["first_bad_wire"] is None and r["checked"] > 50_000, r
    print("self-check of the reduced fixture witness:", r)
    cls, idx, wire = calc.debug_ref("kb.inLen", 0)
    calc.poke(cls, idx, 1, 1, group=1)
    calc.witness_payload_reduced(65, m)
    r = calc.emit_selfcheck_result()
    calc.poke(cls, idx, 1, 1, group=1)
    assert r["first_bad_wire"] is not None and wire < r["first_bad_wire"] < wire + 200_000, (wire, r)
    calc.witness_payload_reduced(65, m)
    assert calc.emit_selfcheck_result()["first_bad_wire"] is None
    calc.close()


def test_inorder_schedule_equals_the_track_schedule(pkg):
    """pob_set_inorder on the GPU, production instantiation: 130 witnesses (three groups) generated and evaluated in dependency order on ONE stream, twice
    over; every record equals the track schedule's, the whole payload of a witness of group 2 equals the oracle's, and four in-order calculators driven
    round-robin on four streams (the bench's service loop in small) agree with it batch by batch"""
    import torch
    from proof_of_burn_amd import inputs as gen
    n = 130
    batches = [gen.synthetic_batch(n, depth=10, seed=0x10 + k, distinct_keys=2) for k in range(3)]
    ref = pkg.WitnessCalculator(PROD, max_batch=n)
    want = []
    for bt in batches:
        res = ref.calculate(bt.inputs, check=True)
        assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in res) and [r.outputs[0] for r in res] == bt.commitments
        want.append([(r.status, r.outputs, r.check_status, r.bad_wire) for r in res])
    ref.close()
    calcs = [pkg.WitnessCalculator(PROD, max_batch=n) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in calcs]
    for k, c in enumerate(calcs):
        c.set_inorder((1, 7, 5, 3)[k])           # (1: one launch per kernel; 3: the fused Poseidon + chain launch; | 4: the round blocks evaluated by the launch that writes them; 7 = what bench.py runs)
    for rnd in range(2):
        for k, c in enumerate(calcs):
            c.upload(batches[(k + rnd) % 3].inputs)
            c.generate(streams[k].cuda_stream)
            c.constraint_check(streams[k].cuda_stream)
        for k, c in enumerate(calcs):
            got = [(r.status, r.outputs, r.check_status, r.bad_wire) for r in c.results(with_check=True)]
            assert got == want[(k + rnd) % 3], (rnd, k)
    ora = O.run(PROD, batches[0].inputs[n - 1])
    idx0 = next(k for k in range(4) if (k + 1) % 3 == 0)         # the calculator that holds batch 0 after round 1 (k = 2: one launch per kernel, rounds evaluated with their expansion)
    gpu = calcs[idx0].witness_payload(n - 1)
    assert np.array_equal(gpu, ora.witness_numpy()), f"first differing wire {_first_diff(gpu, ora.witness_numpy())}"
    # ... and the same batch generated by the calculator that runs bench.py's schedule (fused launch, round blocks evaluated by the launch that writes them): every wire again
    calcs[1].upload(batches[0].inputs); calcs[1].generate(streams[1].cuda_stream); calcs[1].constraint_check(streams[1].cuda_stream)
    assert [(r.status, r.outputs, r.check_status, r.bad_wire) for r in calcs[1].results(with_check=True)] == want[0]
    gpu = calcs[1].witness_payload(n - 1)
    assert np.array_equal(gpu, ora.witness_numpy()), f"fused launches: first differing wire {_first_diff(gpu, ora.witness_numpy())}"
    for c in calcs:
        c.close()


def test_evaluation_riding_with_the_generation_flags_a_corrupted_store(pkg):
    """pob_set_inorder(h, 7), production instantiation, 130 witnesses: k_rounds_gc stores every gate-output array of a round block and compares what it LOADS back with the
    gate's definition; the input rows are loaded back by the launch that stores them likewise.  One store per generation reaches memory corrupted (pob_debug_store_fault: the
    wavefront goes on with the right value): exactly the witnesses of the mask are flagged -- at the round block's first wire, at the input's wire -- by the generation's own
    launches (the evaluation that follows runs neither kernel), in all three groups; the next clean generation reports nothing, the standalone evaluation of the clean vector
    (after a poke and its undoing) reports nothing either"""
    from proof_of_burn_amd import inputs as gen
    n = 130
    bt = gen.synthetic_batch(n, depth=10, seed=0x77, distinct_keys=2)
    calc = pkg.WitnessCalculator(PROD, max_batch=n)
    calc.set_inorder(7)
    res = calc.calculate(bt.inputs, check=True)
    assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in res) and [r.outputs[0] for r in res] == bt.commitments
    nbit = int(calc.info.n_bit)
    rng = np.random.default_rng(23)
    tried, wires = 0, set()
    for bit_index in rng.integers(0, nbit, 400).tolist():
        group = tried % 3
        lanes = int(rng.integers(1, 1 << 63)) if group < 2 else int(rng.integers(1, 4))       # group 2 holds witnesses 128 and 129
        want = calc.store_fault(bit_index, lanes, group=group)
        if want == calc.UNKNOWN_WIRE:
            calc.store_fault(bit_index, 0, group=group)      # (a word outside the round blocks: the G units' part below; disarm)
            continue
        calc.generate(); calc.constraint_check()
        got = [r.bad_wire for r in calc.results(with_check=True)]
        exp = [want if k // 64 == group and (lanes >> (k % 64)) & 1 else None for k in range(n)]
        assert got == exp, (bit_index, group, hex(lanes), want, [(k, g) for k, g in enumerate(got) if g != exp[k]][:6])
        tried += 1; wires.add(want)
        if tried >= 36:
            break
    assert tried >= 36 and len(wires) >= 12
    first = next(i for i in range(int(calc.info.n_sm)) if calc.store_fault(i, 0, cls=calc.CLASS_SM) != calc.UNKNOWN_WIRE)       # the first input row
    nrows = next(k for k in range(1, int(calc.info.n_sm)) if calc.store_fault(first + k, 0, cls=calc.CLASS_SM) == calc.UNKNOWN_WIRE)
    assert nrows == 1 + 16 * 544 + 16 + 1 + 16 * 136 + 1 + 1     # numLeafAddressNibbles, layers, layerLens, numLayers, blockHeader, blockHeaderLen, byteSecurityRelax (proof_of_burn.circom:43-72)
    for k in sorted(set([0, nrows - 1] + rng.integers(0, nrows, 10).tolist())):
        group = k % 3
        lanes = int(rng.integers(1, 1 << 63)) if group < 2 else int(rng.integers(1, 4))
        want = calc.store_fault(first + k, lanes, group=group, cls=calc.CLASS_SM)
        calc.generate(); calc.constraint_check()
        got = [r.bad_wire for r in calc.results(with_check=True)]
        exp = [want if j // 64 == group and (lanes >> (j % 64)) & 1 else None for j in range(n)]
        assert got == exp, (k, group, hex(lanes), want, [(j, g) for j, g in enumerate(got) if g != exp[j]][:6])
    # the G units' own stores (policy.hpp GenPT<true, true>, poseidon_wide.hpp PosWideT<true, true>): where a riding unit stores the word, exactly the witnesses of the mask are
    # flagged; a word of the sponge chains, of the RLP units (plain policy: their evaluation is a launch of pob_constraint_check) or of no unit flags nobody outside the mask
    sizes = calc.class_sizes()
    hit = {calc.CLASS_BIT: 0, calc.CLASS_SM: 0, calc.CLASS_FR: 0}
    for cls in (calc.CLASS_BIT, calc.CLASS_SM, calc.CLASS_FR):
        # (BIT: 97 % of the class are Keccak's words; the first ranks are G-unit wires -- flags, selector rows, decompositions -- between the first sponges' blocks)
        for t, idx in enumerate(rng.integers(0, min(sizes[cls], 40000) if cls == calc.CLASS_BIT else sizes[cls], 120 if cls == calc.CLASS_BIT else 30).tolist()):
            group = t % 3
            lanes = int(rng.integers(1, 1 << 63)) if group < 2 else int(rng.integers(1, 4))
            if calc.store_fault(idx, lanes, group=group, cls=cls) != calc.UNKNOWN_WIRE:
                calc.store_fault(idx, 0, group=group, cls=cls)
                continue
            calc.generate(); calc.constraint_check()
            flagged = [j for j, r in enumerate(calc.results(with_check=True)) if r.bad_wire is not None]
            exp = [j for j in range(n) if j // 64 == group and (lanes >> (j % 64)) & 1]
            assert set(flagged) <= set(exp), (cls, idx, group, hex(lanes), [j for j in flagged if j not in exp][:6])
            hit[cls] += flagged == exp
    assert hit[calc.CLASS_BIT] >= 4 and hit[calc.CLASS_SM] >= 10 and hit[calc.CLASS_FR] >= 15, hit
    calc.generate(); calc.constraint_check()
    assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in calc.results(with_check=True))
    idx = int(calc.info.n_bit) // 2
    calc.poke(calc.CLASS_BIT, idx, 5); calc.constraint_check()
    flagged = [k for k, r in enumerate(calc.results(with_check=True)) if r.bad_wire is not None or r.check_status != 0]
    assert flagged == [5]
    calc.poke(calc.CLASS_BIT, idx, 5); calc.constraint_check()
    assert all(r.bad_wire is None and r.check_status == 0 for r in calc.results(with_check=True))
    calc.close()


def test_generation_and_evaluation_on_different_streams(pkg):
    """pob_generate and pob_constraint_check called with DIFFERENT streams (both schedules): the evaluation is ordered behind the generation's end and the next
    generation behind the evaluation's end by the handle's own events -- the records of three consecutive batches equal the one-stream run's"""
    import torch
    from proof_of_burn_amd import inputs as gen
    n = 130
    batches = [gen.synthetic_batch(n, depth=10, seed=0x40 + k, distinct_keys=2) for k in range(3)]
    bad = dict(batches[1].inputs[70]); bad["numLayers"] = 9                    # one failing witness in the middle batch: the verdicts are not all alike
    batches[1].inputs[70] = bad
    ref = pkg.WitnessCalculator(PROD, max_batch=n)
    want = [[(r.status, r.outputs, r.check_status, r.bad_wire) for r in ref.calculate(bt.inputs, check=True)] for bt in batches]
    ref.close()
    assert want[1][70][0] != 0 and all(w[0] == 0 and w[2] == 0 for w in want[0])
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for inorder in (0, 1, 3, 7):
        c = pkg.WitnessCalculator(PROD, max_batch=n)
        c.set_inorder(inorder)
        for rnd in range(2):
            for k, bt in enumerate(batches):
                c.upload(bt.inputs)
                c.generate(sa.cuda_stream)
                c.constraint_check(sb.cuda_stream)
                if (k + rnd) % 2:                                              # every other batch is read; the others are overwritten while still being evaluated
                    got = [(r.status, r.outputs, r.check_status, r.bad_wire) for r in c.results(with_check=True)]
                    assert got == want[k], (inorder, rnd, k)
        c.close()


def test_device_field_inversions(pkg):
    """the device code's two field inversions (Kaliski almost-inverse: generation and the emitter's field-element IsZero.inv; Fermat ladder: the
    emitter's fall-back beyond its table of small inverses, unreachable for a valid witness -- tests/test_hostsim_cpu.py) on the GPU against
    pow(x, p - 2, p)"""
    import random
    from proof_of_burn_amd import witness as W
    rng = random.Random(5)
    xs = [0, 1, 2, 4097, 65_535, W.P - 4097, W.P - 1, 2 ** 253 % W.P] + [rng.randrange(W.P) for _ in range(2040)]
    buf = b"".join(x.to_bytes(32, "little") for x in xs)
    a, b = ctypes.create_string_buffer(len(buf)), ctypes.create_string_buffer(len(buf))
    assert pkg.load_library().pob_debug_fr_inv(0, buf, len(xs), a, b) == 0
    for k, x in enumerate(xs):
        want = pow(x, W.P - 2, W.P)
        assert int.from_bytes(a.raw[32 * k:32 * k + 32], "little") == want and int.from_bytes(b.raw[32 * k:32 * k + 32], "little") == want, (k, x)


def test_pow_search_gpu_matches_host(pkg):
    """row f1: the input producer's proof-of-work (tests/main.py:47-56) as a HIP kernel returns the same first key as the
    sequential host search, for 1..3 zero bytes, including a start key whose low 64 bits wrap."""
    import ctypes
    lib = pkg.load_library()
    from proof_of_burn_amd import inputs as gen
    cases = [(12345, 10**18, 7, 1), (2**200 + 99, 5 * 10**17, 2**100 + 3, 2), ((1 << 64) - 1000, 10**18, 0, 2), (2**255 - 5, 1, 2**250, 3)]
    for start, reveal, extra, zb in cases:
        postfix = reveal.to_bytes(32, "big") + extra.to_bytes(32, "big") + b"EIP-7503"
        o1, o2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        t1 = lib.pob_pow_search(start.to_bytes(32, "big"), postfix, len(postfix), zb, 1 << 28, o1)
        t2 = lib.pob_pow_search_gpu(0, start.to_bytes(32, "big"), postfix, len(postfix), zb, 1 << 28, o2)
        assert t1 >= 0 and t1 == t2 and o1.raw == o2.raw, (start, zb, t1, t2)
        key = int.from_bytes(o2.raw, "big")
        assert key == start + t1
        assert pkg.keccak256(o2.raw + postfix)[:zb] == bytes(zb)
        assert gen.pow_search(start, reveal, extra, zb, device=0) == key


def test_reading_a_batch_does_not_stall_the_partner(pkg):
    """the service loop's read path: pob_results_fetch / pob_results_wait of batch k go through the handle's OWN events -- when the wait
    returns, the partner calculator's generation of batch k+1 (enqueued before) is still running, i.e. nothing synchronised the device;
    and the legacy pob_results of the same batch gives the same answer afterwards"""
    import torch
    from proof_of_burn_amd import inputs as gen, witness as W
    main = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
    B = 512
    batch = gen.synthetic_batch(B, depth=10, seed=0xB0B, distinct_keys=4)
    a, b = pkg.WitnessCalculator(main, max_batch=B), pkg.WitnessCalculator(main, max_batch=B)
    a.set_partner(b)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    pin = pkg.PinnedInputs(a, B)
    a.pack_json([json.dumps(d) for d in batch.inputs], out=pin)
    still_running = 0
    for rnd in range(3):
        a.upload_pinned_async(pin); a.generate(sa.cuda_stream)
        b.upload_pinned_async(pin)
        a.constraint_check(sa.cuda_stream); a.fetch_records()
        for _ in range(4):                              # (four generations of the same inputs: several times a's evaluation, whatever the box)
            b.generate(sb.cuda_stream)
        done_b = torch.cuda.Event(); done_b.record(sb)
        rec = a.wait_records()                          # returns when a's records are on the host ...
        still_running += 0 if done_b.query() else 1     # ... while b is still generating
        assert not rec["status"].any() and (rec["check_status"] == W.CLEAN).all() and (rec["bad_wire"] == W.CLEAN).all()
        assert [int.from_bytes(c.tobytes(), "little") for c in rec["commitment"]] == batch.commitments
        b.constraint_check(sb.cuda_stream); b.fetch_records()
        recb = b.wait_records()
        assert np.array_equal(recb["commitment"], rec["commitment"]) and (recb["check_status"] == W.CLEAN).all()
    assert still_running == 3, "waiting for a batch's records synchronised the partner's generation"
    res = a.results(with_check=True)
    assert [r.outputs[0] for r in res] == batch.commitments and all(r.check_status == 0 and r.bad_wire is None for r in res)
    pin.free(); a.close(); b.close()


def test_rccl_world_of_one_gathers_the_device_records(pkg, tmp_path):
    """multi-GPU readiness without a node: the `nccl` backend (= RCCL on ROCm) initialised with world size 1 in a fresh process, the
    calculator's device-resident record tensor (a zero-copy __cuda_array_interface__ view of library memory) pushed through
    all_gather_into_tensor on a side stream: RCCL loads, accepts the view and returns the records"""
    import subprocess
    import sys
    from proof_of_burn_amd import distributed as Dd
    D_free_port = Dd.free_port()
    script = tmp_path / "rccl1.py"
    script.write_text(f"""
import json, os, sys
sys.path.insert(0, {ROOT!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str({D_free_port}), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", GPU_MAX_HW_QUEUES="16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from proof_of_burn_amd import WitnessCalculator, distributed as D, witness as W
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
s = next(x for x in json.load(open(os.path.join({ROOT!r}, "tests", "golden", "suites.json"))) if x["name"] == "test_spend")
calc = WitnessCalculator("Spend(31)", max_batch=8)
inputs = [c["input"] for c in s["cases"]]
res = calc.calculate(inputs, check=True)
rec = D.device_records(calc, len(inputs))
gs = torch.cuda.Stream()
with torch.cuda.stream(gs):
    out = torch.empty_like(rec)
    dist.all_gather_into_tensor(out, rec.contiguous())
gs.synchronize()
st, outs = D.unpack_records(out.cpu())
cs, bw = D.unpack_verdicts(out.cpu())
for i, r in enumerate(res):
    assert int(st[i]) == r.status
    if r.ok:
        assert int.from_bytes(bytes(outs[i].tolist()), "little") == r.outputs[0] and int(cs[i]) == D.CLEAN and int(bw[i]) == D.CLEAN
print("rccl ranks", dist.get_world_size(), dist.get_backend())
dist.destroy_process_group()
calc.close()
""")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ranks 1 nccl" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_c_abi_record_gather_over_rccl_world_of_one(pkg):
    """pob_gather_records: the multi-GPU path's one collective for a host that is not Python -- the library dlopens librccl and all-gathers the calculator's device records
    over the CALLER's ncclComm_t.  Here: a communicator of one rank made through ctypes (ncclGetUniqueId / ncclCommInitRank, as a C host would), a batch with one failing
    witness, the gathered bytes == the records the host reads; on a node the same call runs with nranks = 8."""
    import ctypes
    import torch
    s = _suite("test_spend")
    inputs = [c["input"] for c in s["cases"]]
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=len(inputs))
    res = calc.calculate(inputs, check=True)
    assert any(not r.ok for r in res) and any(r.ok for r in res)
    rccl = ctypes.CDLL("librccl.so.1")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    n = len(inputs)
    out = torch.zeros((n, 44), dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.Stream()
    calc.gather_records_rccl(comm.value, out.data_ptr(), n, st.cuda_stream)
    st.synchronize()
    calc.fetch_records()
    rec = calc.wait_records()
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :4].copy().view(np.uint32)[:, 0], rec["status"]) and np.array_equal(got[:, 12:], rec["commitment"])
    assert np.array_equal(got[:, 4:8].copy().view(np.uint32)[:, 0], rec["check_status"])
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    rccl.ncclCommDestroy(comm)
    calc.close()


def test_seeded_differential_fixture_instantiation(pkg):
    """64 random VALID proofs of depths 2..4 on the fixture instantiation (seeded): outputs, clean evaluator and the WHOLE canonical payload
    of every one of them against the oracle"""
    from proof_of_burn_amd import inputs as gen
    params = (4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)
    inputs, commitments = [], []
    for depth, n, seed in ((2, 21, 1001), (3, 21, 2002), (4, 22, 3003)):
        bt = gen.synthetic_batch(n, depth=depth, seed=seed, distinct_keys=3, params=params)
        inputs += bt.inputs; commitments += bt.commitments
    with O.OraclePool() as pool:                  # the oracle's 64 runs in worker processes, beside the GPU's batch
        jobs = pool.submit(POB_FIX, inputs, digest=True)
        calc = pkg.WitnessCalculator(POB_FIX, max_batch=64)
        res = calc.calculate(inputs, check=True)
        oras = pool.collect(jobs)
    buf = np.empty(32 * calc.nwitness, dtype=np.uint8)
    for i, (r, c, ora) in enumerate(zip(res, commitments, oras)):
        assert r.ok and r.outputs == [c] and r.check_status == 0 and r.bad_wire is None, (i, r)
        assert not ora.failed and ora.outputs() == [c]
        _assert_payload(calc, i, POB_FIX, inputs[i], ora, buf, f"witness {i}")
    calc.close()


# ---------------------------------------------------------------------------- gadget-level mains (reference tests/test.py:146-201)
def test_reference_list_gadget_mains_through_run(pkg, capsys):
    """every gadget-level entry of the reference's test list through the run() shim, unchanged call shape (tests/test.py:6-75)"""
    from proof_of_burn_amd.harness import run
    from tests import gadget_cases as GC
    suites = GC.gadget_suites()
    assert len(suites) == 54
    for s in suites:
        got = run(s["main"], [(c["input"], c["expected"]) for c in s["cases"]])
        assert got == [c["expected"] for c in s["cases"]], s["main"]


def test_gadget_mains_payload_and_evaluator(pkg):
    """the same 54 suites: full canonical payload == the oracle's witness, evaluator clean, no witness for a rejected input"""
    from tests import gadget_cases as GC
    bad = []
    for s in GC.gadget_suites():
        bad += GC.check_suite(pkg, s)
    assert not bad, bad[:6]


def test_gadget_mains_evaluator_catches_corruption(pkg):
    """every gadget main: stored values of every storage class corrupted (63 lanes per pass), exactly the corrupted lanes flagged"""
    from tests import gadget_cases as GC
    bad = []
    for s in GC.gadget_suites():
        missed, done = GC.sweep_suite(pkg, s, per_class=300)
        if missed:
            bad.append((s["main"], done, missed[:4]))
    assert not bad, bad[:4]


def test_gadget_mains_seeded_differential(pkg):
    """64 seeded random inputs per template (16 for the Keccak mains) against the oracle: decision, outputs, payload"""
    from tests import gadget_cases as GC
    bad, total = [], 0
    for s in GC.gadget_suites():
        b, nok = GC.differential(pkg, s, n=16 if s["main"].split("(")[0] in GC.KECCAK_MAINS else 64)
        bad += b
        total += nok
    assert not bad, bad[:4]
    assert total > 1500


def test_queued_emission_equals_one_at_a_time(pkg):
    """two witnesses in flight (pob_emit_queue, three rotating window slots): every window of every witness equals what the same witness
    gives emitted alone -- which test_reference_suite_proof_of_burn_and_wtns compares with the oracle -- for the O0 and the reduced payload"""
    import hashlib
    s = _suite("test_proof_of_burn")
    ok_cases = [c["input"] for c in s["cases"] if c["expected"] is not None]
    inputs = [ok_cases[i % len(ok_cases)] for i in range(5)]
    inputs[2] = dict(inputs[2]); inputs[2]["burnExtraCommitment"] = "12345"      # (a different witness: fails the PoW or not, either is fine)
    calc = pkg.WitnessCalculator(POB_FIX, max_batch=5)
    res = calc.calculate(inputs)
    good = [i for i, r in enumerate(res) if r.ok]
    assert len(good) >= 3
    W = calc.nwitness
    keep = np.arange(0, W, 7, dtype=np.uint32)

    def digest(idx, win, kp, queue_next=None):
        h, pos = hashlib.sha256(), 0
        for w0, view in calc.witness_windows(idx, window_wires=win, keep=kp):
            if w0 == 0 and queue_next is not None:
                calc.emit_queue(queue_next)
            assert w0 == pos
            h.update(view.tobytes()); pos += view.size // 32
        assert pos == (W if kp is None else kp.size)
        return h.hexdigest()

    for kp in (None, keep):
        for win in (4 << 20, 3_000_001):
            alone = {i: digest(i, win, kp) for i in good}
            seq = good + good[::-1]
            got = [digest(i, win, kp, queue_next=seq[k + 1] if k + 1 < len(seq) else None) for k, i in enumerate(seq)]
            assert got == [alone[i] for i in seq], (kp is not None, win)
    calc.close()


def test_gadget_mains_at_production_sizes(pkg):
    """the templates at the parameters the production circuit uses (SubstringCheck(136, 31), RlpMerklePatriciaTrieLeaf(32, 31), KeccakBytes(4) ...):
    24 seeded random inputs each against the oracle -- decision, outputs, evaluator, full payload"""
    from tests import gadget_cases as GC
    bad, total = [], 0
    for main in GC.LARGE_MAINS:
        b, nok = GC.differential(pkg, {"main": main}, n=24, seed=5)
        bad += b
        total += nok
    assert not bad, bad[:4]
    assert total > 400
