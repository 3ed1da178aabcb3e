"""The product's own kernels and host scheduler on the CPU (tests/hostsim: HIP-on-fibers shim, test infrastructure): layout
arithmetic, generation, the constraint evaluator's relations and the .wtns emitter are exercised without a GPU.  The GPU versions
of these tests live in test_gpu_parity.py; what this cannot cover is code generation and timing."""
import json
import os

import numpy as np
import pytest

from tests import evaluator_cases as EC
from tests import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POB_FIX = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"


def _suite(name):
    with open(os.path.join(ROOT, "tests", "golden", "suites.json")) as f:
        return next(s for s in json.load(f) if s["name"] == name)


@pytest.fixture(scope="module")
def pkg():
    """proof_of_burn_amd with libpob_hostsim.so in place of libpob_hip.so (restored afterwards)"""
    from tests.hostsim import build as hb
    import proof_of_burn_amd
    from proof_of_burn_amd import witness as W
    lib = hb.build()
    old = (W.LIB_PATH, W._lib)
    W.LIB_PATH, W._lib = lib, None
    yield proof_of_burn_amd
    W.LIB_PATH, W._lib = old


def test_spend_suite_wtns_and_evaluator(pkg, tmp_path):
    s = _suite("test_spend")
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=4)
    res = calc.calculate([c["input"] for c in s["cases"]], check=True)
    for i, (c, r) in enumerate(zip(s["cases"], res)):
        assert (r.outputs if r.ok else None) == c["expected"]
        if r.ok:
            assert r.check_status == 0 and r.bad_wire is None
            ora = O.run("Spend(31)", c["input"])
            assert np.array_equal(calc.witness_payload(i), ora.witness_numpy())
        else:
            with pytest.raises(RuntimeError):          # like the reference binary: no witness for a failed input
                calc.witness_payload(i)
    # the window pipeline underneath: 27 windows of 100 000 wires, then another witness through the same buffers
    ora = O.run("Spend(31)", s["cases"][0]["input"])
    ref = ora.witness_numpy().copy()
    pos = 0
    for w0, view in calc.witness_windows(0, window_wires=100_000):
        assert w0 == pos and np.array_equal(view, ref[32 * w0:32 * w0 + view.size])
        pos += view.size // 32
    assert pos == calc.nwitness
    assert np.array_equal(calc.witness_payload(3), O.run("Spend(31)", s["cases"][3]["input"]).witness_numpy())
    # two witnesses in flight (pob_emit_queue): witness 3's first window is expanded behind witness 0's last ones, three window slots rotate
    ref3 = O.run("Spend(31)", s["cases"][3]["input"]).witness_numpy().copy()
    for win in (100_000, 1_000_000, 1_700_000, 5_000_000):
        calc.emit_queue(3)                                   # announced before the emission it follows ...
        for idx, want in ((0, ref), (3, ref3), (0, ref)):
            pos = 0
            for w0, view in calc.witness_windows(idx, window_wires=win):
                if idx == 3 and w0 == 0:
                    calc.emit_queue(0)                       # ... or while it runs
                assert w0 == pos and np.array_equal(view, want[32 * w0:32 * w0 + view.size]), (win, idx, w0)
                pos += view.size // 32
            assert pos == calc.nwitness
    keep = np.arange(0, calc.nwitness, 5, dtype=np.uint32)   # ... and the reduced payload through the same three slots
    for idx, want, nxt in ((0, ref, 3), (3, ref3, 0), (0, ref, None)):
        pos = 0
        for w0, view in calc.witness_windows(idx, window_wires=90_000, keep=keep):
            if w0 == 0 and nxt is not None:
                calc.emit_queue(nxt)
            assert w0 == pos and np.array_equal(view.reshape(-1, 32), want.reshape(-1, 32)[keep[w0:w0 + view.size // 32]]), (idx, w0)
            pos += view.size // 32
        assert pos == keep.size
    calc.emit_queue(0)                                       # an announcement that is not followed: the prepared window is dropped
    assert np.array_equal(calc.witness_payload(3), ref3) and np.array_equal(calc.witness_payload(3), ref3)
    # a window pre-made from one batch must not serve the next batch: announce witness 0, drain witness 3 (witness 0's first window of batch A
    # is expanded behind it), generate batch B = the cases in reverse order, begin witness 0 with the same window size
    order = [c["input"] for c in s["cases"]]
    for win in (1_000_000, 5_000_000):
        calc.calculate(order)
        calc.emit_queue(0)
        for _ in calc.witness_windows(3, window_wires=win):
            pass
        resB = calc.calculate(order[::-1])
        assert resB[0].ok
        got0 = np.concatenate([v.copy() for _, v in calc.witness_windows(0, window_wires=win)])
        assert np.array_equal(got0, ref3), ("the first window came from the previous batch", win)
    res = calc.calculate(order, check=True)
    # a batch that was generated but not evaluated says so (it is not reported as clean)
    calc.generate()
    r0 = calc.results(with_check=True)[0]
    assert r0.evaluated is False and r0.check_status is None and r0.bad_wire is None
    calc.constraint_check()
    r0 = calc.results(with_check=True)[0]
    assert r0.evaluated is True and r0.check_status == 0 and r0.bad_wire is None
    bad_idx = next(i for i, r in enumerate(res) if not r.ok)
    with pytest.raises(RuntimeError):
        calc.emit_queue(bad_idx)                             # no witness for a failed input, announced or not
    assert np.array_equal(calc.witness_payload(0), ref)
    path = str(tmp_path / "w.wtns")
    calc.write_wtns(0, path)
    ora = O.run("Spend(31)", s["cases"][0]["input"])
    with open(path, "rb") as f:
        assert np.array_equal(np.frombuffer(f.read(), dtype=np.uint8), ora.wtns_numpy())
    calc.close()


def test_proof_of_burn_fixture_suite_and_wtns(pkg):
    s = _suite("test_proof_of_burn")
    calc = pkg.WitnessCalculator(s["main"], max_batch=8)
    res = calc.calculate([c["input"] for c in s["cases"]], check=True)
    assert [r.outputs if r.ok else None for r in res] == [c["expected"] for c in s["cases"]]
    for r in res:
        if r.ok:
            assert r.check_status == 0 and r.bad_wire is None, r
    ora = O.run(s["main"], s["cases"][3]["input"])
    assert np.array_equal(calc.witness_payload(3), ora.witness_numpy())
    calc.close()


def test_evaluator_detects_pokes_in_every_class_spend(pkg):
    s = _suite("test_spend")
    calc = EC.open_identical_batch(pkg, "Spend(31)", s["cases"][0]["input"])
    missed, done = EC.uniform_sweep(calc, {EC.SM: 126, EC.FR: 126, EC.BIT: 126})
    assert not missed, f"{len(missed)} mis-detections of {done}: {missed[:6]}"
    bad = EC.named_pokes(calc, [("poseidon", 200), ("poseidon", 640), ("pad.div.out", 0), ("pad.div.rem", 0), ("commitment", 0)])
    assert not bad, bad
    calc.close()


def test_evaluator_detects_pokes_fixture(pkg):
    s = _suite("test_proof_of_burn")
    calc = EC.open_identical_batch(pkg, POB_FIX, s["cases"][0]["input"])
    missed, done = EC.uniform_sweep(calc, {EC.SM: 126, EC.FR: 126})
    assert not missed, f"{len(missed)} mis-detections of {done}: {missed[:6]}"
    bad = EC.named_pokes(calc, [("poseidon", 300), ("sc.exists", 5), ("pad.div.out", 1)])
    assert not bad, bad
    calc.close()


def test_two_calculators_pipelined_over_consecutive_batches(pkg):
    """pob_set_partner: two handles on the shared side streams, batch k generated by one while batch k-1 is evaluated by the other
    (call order of bench.py --pipeline 1); results of every batch as from a lone calculator, and unlinking on close"""
    s = _suite("test_proof_of_burn")
    inputs = [c["input"] for c in s["cases"]]
    expected = [c["expected"] for c in s["cases"]]
    a, b = pkg.WitnessCalculator(s["main"], max_batch=8), pkg.WitnessCalculator(s["main"], max_batch=8)
    with pytest.raises(RuntimeError):
        a.set_partner(a)                                 # a handle cannot be its own partner
    a.set_partner(b); b.set_partner(a)
    batches = [inputs, inputs[::-1], inputs[2:] + inputs[:2]]
    exp = [expected, expected[::-1], expected[2:] + expected[:2]]
    calcs, prev, seen = [a, b], None, []
    for k, bt in enumerate(batches):
        cur = calcs[k % 2]
        if prev is not None:
            prev[0].constraint_check()
            seen.append((prev[1], prev[0].results(with_check=True)))
        cur.upload_packed(*cur.pack(bt))
        cur.generate()
        prev = (cur, k)
    prev[0].constraint_check()
    seen.append((prev[1], prev[0].results(with_check=True)))
    for k, res in seen:
        assert [r.outputs if r.ok else None for r in res] == exp[k]
        assert all(r.check_status == 0 and r.bad_wire is None for r in res if r.ok)
    i = next(j for j, e in enumerate(exp[2]) if e is not None)
    assert np.array_equal(a.witness_payload(i), O.run(s["main"], batches[2][i]).witness_numpy())
    a.close()                                            # unlinks b
    b.generate(); b.constraint_check()
    assert [r.outputs if r.ok else None for r in b.results(with_check=True)] == exp[1]
    b.close()


def test_service_loop_api_on_the_shim(pkg):
    """the per-batch service loop of bench.py on the CPU shim: inputs through the native JSON loader into pinned memory, asynchronous
    upload, two linked calculators over consecutive batches, records {status, verdict, commitment} fetched per batch AFTER the
    evaluation and read through the handle's own events; a generate-only batch reports NOT_EVALUATED"""
    from proof_of_burn_amd import witness as W
    s = _suite("test_proof_of_burn")
    inputs = [c["input"] for c in s["cases"]]
    expected = [c["expected"] for c in s["cases"]]
    a, b = pkg.WitnessCalculator(s["main"], max_batch=8), pkg.WitnessCalculator(s["main"], max_batch=8)
    a.set_partner(b)
    order = [list(range(len(inputs))), list(range(len(inputs)))[::-1], [2, 3, 0, 1] + list(range(4, len(inputs)))]
    pins = []
    for k, o in enumerate(order):
        pin = pkg.PinnedInputs(a, len(o), compact=(k != 1))      # batches 0 and 2 travel in the byte form (pob_upload_inputs8_async), batch 1 as int32 rows
        fr, sm, forced = a.pack_json([json.dumps(inputs[i]) for i in o], threads=2, out=pin)
        ref = a.pack([inputs[i] for i in o])
        assert fr is pin.fr and all(np.array_equal(x, y) for x, y in zip(ref, (pin.fr, pin.widened(), pin.forced)))
        assert pin.compact == (k != 1) and (pin.bytes_ok or not pin.compact)
        pins.append(pin)
    calcs, prev, got = [a, b], None, {}

    def finish(c, k):
        calcs[c].constraint_check()
        calcs[c].fetch_records()
        got[k] = calcs[c].wait_records().copy()

    for c in calcs:
        c.probe_check_kernel(True)                   # (bench.py's in-step timing of the dominant kernel: events around every evaluation from here on)

    for k, pin in enumerate(pins):
        c = k % 2
        if prev is not None:
            finish(*prev)
        nbytes = calcs[c].upload_pinned_async(pin)
        assert nbytes == pin.fr.nbytes + (pin.sm8.nbytes + pin.exc.nbytes if pin.compact else pin.sm.nbytes)
        calcs[c].generate()
        if k == 0:                                   # records of a batch whose evaluation has not run
            calcs[c].fetch_records()
            r0 = calcs[c].wait_records()
            assert (r0["check_status"] == W.NOT_EVALUATED).all() and (r0["bad_wire"] == W.NOT_EVALUATED).all()
        prev = (c, k)
    finish(*prev)
    # the probe is read one batch late, when the calculator has already been given its next batch (as the bench does)
    calcs[0].upload_pinned_async(pins[0]); calcs[0].generate()
    assert calcs[0].probe_check_kernel(True, read=True) > 0
    for c in calcs:
        c.probe_check_kernel(False)
    with pytest.raises(RuntimeError):
        calcs[0].probe_check_kernel(True, read=True)
    for k, o in enumerate(order):
        rec = got[k]
        assert rec.dtype == W.RECORD_DTYPE and rec.shape[0] == len(o)
        for j, i in enumerate(o):
            if expected[i] is None:
                assert rec["status"][j] != 0
            else:
                assert rec["status"][j] == 0 and rec["check_status"][j] == W.CLEAN and rec["bad_wire"][j] == W.CLEAN
                assert int.from_bytes(rec["commitment"][j].tobytes(), "little") == expected[i][0]
    # the blocking form reads the same records
    res = calcs[prev[0]].results(with_check=True)
    assert [r.outputs if r.ok else None for r in res] == [expected[i] for i in order[prev[1] if prev[0] != 0 else 0]]
    for pin in pins:
        pin.free()
    a.close(); b.close()


def test_declaration_order_switch_flips_all_three_statements(pkg, tmp_path):
    """the one knob for re-diffing against a real circom build: POB_DECL_ORDER=1 (+ ORACLE_DECL_ORDER=1 for the oracle) moves
    Num2Bits_strict's aliasCheck and MultiAND's and2 to declaration order in the HIP planner / kernels, in the circuit model and in the
    oracle TOGETHER: same wire count, payloads still equal, the model's R1CS still accepts the product's witness, and the payload really
    differs from the default numbering's (fresh process: the switch is read once)"""
    from tests.hostsim import build as hb
    script = tmp_path / "flip.py"
    script.write_text(f"""
import json, os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np
from proof_of_burn_amd import witness as W
from proof_of_burn_amd.circuit_model import check as CK, circuit
from tests import oracle_ffi as O
W.LIB_PATH, W._lib = {hb.build()!r}, None
suites = json.load(open(os.path.join({ROOT!r}, "tests", "golden", "suites.json")))
out = {{}}
for name, main, case in (("test_spend", "Spend(31)", 0), ("test_proof_of_burn", None, 3)):
    if name == "test_proof_of_burn" and os.environ.get("POB_DECL_ORDER") != "1":
        continue                     # (the fixture instantiation -- MultiAND(7) of the leaf detectors -- in the flipped numbering only: the default one is every other test)
    s = next(x for x in suites if x["name"] == name)
    main = main or s["main"]
    calc = W.WitnessCalculator(main, max_batch=1)
    r = calc.calculate(s["cases"][case]["input"], check=True)[0]
    assert r.ok and r.outputs == s["cases"][case]["expected"] and r.check_status == 0 and r.bad_wire is None
    ora = O.run(main, s["cases"][case]["input"])
    pay = calc.witness_payload(0)
    assert ora.nwitness == calc.nwitness and np.array_equal(pay, ora.witness_numpy())
    c = circuit(main)
    assert c.n_wires == calc.nwitness and CK.check_witness(c, CK.Witness(pay.copy())) == []
    import hashlib
    out[main] = hashlib.sha256(pay.tobytes()).hexdigest()
    calc.close()
print("DIGESTS " + json.dumps(out))
""")
    import subprocess
    import sys
    digests = []
    for env in ({"POB_DECL_ORDER": "0", "ORACLE_DECL_ORDER": "0"}, {"POB_DECL_ORDER": "1", "ORACLE_DECL_ORDER": "1"}):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        digests.append(json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("DIGESTS "))[8:]))
    assert len(digests[1]) == 2 and all(digests[0][k] != digests[1][k] for k in digests[0])


# ---------------------------------------------------------------------------- gadget-level mains (reference tests/test.py:146-201)
def test_gadget_mains_reference_suites_on_the_shim(pkg):
    """all 54 gadget-level entries of the reference's list through the product's planner, generator, evaluator and emitter: outputs ==
    the reference's expected values, full payload == the oracle's witness, evaluator clean, no witness for a rejected input"""
    from tests import gadget_cases as GC
    suites = GC.gadget_suites()
    assert len(suites) == 54
    bad = []
    for s in suites:
        bad += GC.check_suite(pkg, s)
    assert not bad, bad[:6]


def test_gadget_mains_through_the_run_shim(pkg, capsys):
    """the reference harness' own call shape, run(main, [(input, expected), ...]) (tests/test.py:6-75), for mains of every kind: one unit,
    field-valued inputs, a Keccak sponge; a wrong expectation raises like the reference's harness does"""
    from proof_of_burn_amd.harness import run
    from tests import gadget_cases as GC
    by_name = {s["main"]: s for s in GC.gadget_suites()}
    for main in ("Selector(5)", "Num2BitsSafe(254)", "RlpMerklePatriciaTrieLeaf(3, 3)", "KeccakBytes(1)", "ProofOfWorkChecker()"):
        s = by_name[main]
        got = run(main, [(c["input"], c["expected"]) for c in s["cases"]])
        assert got == [c["expected"] for c in s["cases"]]
    s = by_name["Selector(5)"]
    with pytest.raises(Exception, match="Unexpected output|Expected null"):
        run("Selector(5)", [(s["cases"][0]["input"], [12345])])
    # a byte-class signal outside int32 (the reference would reduce it mod p and run): never a wrong answer and never an escaping
    # NotImplementedError -- the case is reported as None with a warning on stderr, which the harness' own comparison turns into
    # "Expected null!" when the list expects outputs, and accepts when the list expects None
    capsys.readouterr()
    with pytest.raises(Exception, match="Expected null"):
        run("Selector(5)", [({"vals": [1, 2, 3, 4, 1 << 40], "select": 0}, [1])])
    assert "does not fit the int32 class" in capsys.readouterr().err
    assert run("Selector(5)", [({"vals": [1, 2, 3, 4, 1 << 40], "select": 0}, None), (s["cases"][0]["input"], s["cases"][0]["expected"])]) == [None, s["cases"][0]["expected"]]
    with pytest.raises((ValueError, RuntimeError)):   # an instantiation the planner does not support
        pkg.WitnessCalculator("Poseidon(9)")


def test_gadget_mains_evaluator_catches_corruption(pkg):
    """stored values of every storage class of a gadget main are corrupted (63 lanes per pass, lane 0 = control): the evaluator must
    flag exactly the corrupted lanes"""
    from tests import gadget_cases as GC
    by_name = {s["main"]: s for s in GC.gadget_suites()}
    for main in ("Selector(5)", "Pad(3, 4)", "Divide(16)", "RlpInteger(3)", "Num2BitsSafe(32)"):
        missed, done = GC.sweep_suite(pkg, by_name[main], per_class=120)
        assert not missed, (main, done, missed[:5])


def test_gadget_mains_seeded_differential(pkg):
    """seeded random inputs per template (accepting and rejecting) against the oracle: decision, outputs, payload"""
    from tests import gadget_cases as GC
    bad, total = [], 0
    for s in GC.gadget_suites():
        if s["main"].split("(")[0] in GC.KECCAK_MAINS:
            continue
        b, nok = GC.differential(pkg, s, n=12)
        bad += b
        total += nok
    assert not bad, bad[:4]
    assert total > 250


def test_gadget_mains_at_production_sizes(pkg):
    """a few templates at the parameters the production circuit uses, seeded random inputs against the oracle (the full list runs on the GPU)"""
    from tests import gadget_cases as GC
    bad, total = [], 0
    for main in ("SubstringCheck(136, 31)", "RlpMerklePatriciaTrieLeaf(32, 31)", "Concat(36, 103)", "Pad(4, 136)", "SelectorArray2D(5, 4, 16)", "Num2BitsSafe(253)", "LeafDetector(136)",
                 "Mask(300)", "Concat(300, 5)"):          # (beyond 256 entries Mask's lane-distributed filter[] runs do not reach: the per-wire loop, advisor round 4)
        b, nok = GC.differential(pkg, {"main": main}, n=5, seed=5)
        bad += b
        total += nok
    assert not bad, bad[:4]
    assert total >= 10


def test_emitter_inverse_paths_and_field_inversions(pkg):
    """the emitter's inverse paths.  IsZero.inv of a SMALL operand comes from the table of inverses of -4096..4096; the Fermat fall-back beyond
    the table (policy.hpp EmitP::emit_inv) is unreachable for a valid witness of these circuits: every IsEqual / IsZero over small operands
    compares byte positions, bytes or lengths that an assert bounds (positions < 136 * 16, lengths by AssertLessEqThan / Num2Bits(16) -- a
    LeafDetector main with layerLen = 60 001 fails AssertLessEqThan, it does not reach the emitter), and gadget mains bound their template
    parameters by 4 096.  So: (i) the counters over whole emitted payloads say which paths ran (table, field-element inverses; Fermat: 0),
    (ii) both field inversions of the device code are compared with pow(x, p - 2, p) directly (pob_debug_fr_inv), the values a Fermat
    fall-back would see (|k| > 4096, p - k) included"""
    import json
    from proof_of_burn_amd import witness as W
    with open(os.path.join(ROOT, "tests", "golden", "suites.json")) as f:
        suites = json.load(f)
    ld = next(x for x in suites if x["main"].startswith("LeafDetector"))
    calc = pkg.WitnessCalculator(ld["main"], max_batch=2)
    res = calc.calculate([ld["cases"][0]["input"], dict(ld["cases"][0]["input"], layerLen=60_001)], check=True)
    assert res[0].ok and not res[1].ok
    calc.emit_counters()
    assert np.array_equal(calc.witness_payload(0), O.run(ld["main"], ld["cases"][0]["input"]).witness_numpy())
    c = calc.emit_counters()
    assert c["table"] > 0 and c["fermat"] == 0, c
    calc.close()
    sp = next(x for x in suites if x["main"].startswith("SubstringCheck"))
    calc = pkg.WitnessCalculator(sp["main"], max_batch=1)
    ok_case = next(cs for cs in sp["cases"] if cs["expected"] is not None)
    assert calc.calculate(ok_case["input"], check=True)[0].ok
    calc.emit_counters()
    assert np.array_equal(calc.witness_payload(0), O.run(sp["main"], ok_case["input"]).witness_numpy())
    c = calc.emit_counters()
    assert c["field_nonzero"] > 0 and c["fermat"] == 0, c               # SubstringCheck's IsEqual(exists) operands: field-element inverses
    calc.close()
    import ctypes, random
    rng = random.Random(11)
    xs = [0, 1, 2, 4097, 60_001 - 16, W.P - 4097, W.P - 1, W.P - 5000, 2 ** 253 % W.P] + [rng.randrange(W.P) for _ in range(55)]
    buf = b"".join(x.to_bytes(32, "little") for x in xs)
    a, b = ctypes.create_string_buffer(len(buf)), ctypes.create_string_buffer(len(buf))
    assert pkg.load_library().pob_debug_fr_inv(0, buf, len(xs), a, b) == 0
    for k, x in enumerate(xs):
        want = pow(x, W.P - 2, W.P)
        assert int.from_bytes(a.raw[32 * k:32 * k + 32], "little") == want and int.from_bytes(b.raw[32 * k:32 * k + 32], "little") == want, (k, x)


def test_emit_selfcheck_on_written_values(pkg):
    """pob_emit_selfcheck: the derived wires' own relations (IsZero: in * inv === 1 - out, in * out === 0; IsEqual: IsZero.in === in[1] - in[0],
    out === IsZero.out; SubstringCheck: M[i+1] === M[i] + mainInput[i] * 256^i) evaluated on the values WRITTEN into the emission windows.  Clean for
    valid witnesses (one window and many); a witness emitted from a resident vector with a poked OPERAND (KeccakBytes.inLen, a SubstringCheck input byte)
    violates the relations of the derived wires that consume it, and the check names such a wire"""
    s = _suite("test_proof_of_burn")
    calc = pkg.WitnessCalculator(POB_FIX, max_batch=2)
    inp = s["cases"][0]["input"]
    assert all(r.ok for r in calc.calculate([inp, inp], check=True))
    ref = O.run(POB_FIX, inp).witness_numpy()
    calc.emit_selfcheck(True)
    got = np.concatenate([v.copy() for _, v in calc.witness_windows(1, window_wires=1_000_000)])          # 65 windows
    assert np.array_equal(got, ref)
    r = calc.emit_selfcheck_result()
    assert r["first_bad_wire"] is None and r["checked"] > 50_000 and r["skipped"] < 200, r
    calc.emit_selfcheck(False)
    calc.close()
    # a poked operand, on Spend(31) (its PublicCommitment hashes through a KeccakBytes): KeccakBytes.inLen -- the IsEqual([i, inLen]) children are derived from it, the
    # stored isEq[] bits they must equal are not
    sp = _suite("test_spend")
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=2)
    inp = sp["cases"][0]["input"]
    assert all(r.ok for r in calc.calculate([inp, inp], check=True))
    ref = O.run("Spend(31)", inp).witness_numpy()
    calc.emit_selfcheck(True)
    assert np.array_equal(calc.witness_payload(1), ref)
    r = calc.emit_selfcheck_result()
    assert r["first_bad_wire"] is None and r["checked"] > 1000, r
    cls, idx, wire = calc.debug_ref("kb.inLen", 0)
    calc.poke(cls, idx, 1, 1)
    got = calc.witness_payload(1)
    r = calc.emit_selfcheck_result()
    assert not np.array_equal(got, ref) and r["first_bad_wire"] is not None and wire < r["first_bad_wire"] < wire + 40_000, (wire, r)
    assert np.array_equal(calc.witness_payload(0), ref) and calc.emit_selfcheck_result()["first_bad_wire"] is None      # the neighbouring witness is untouched
    calc.poke(cls, idx, 1, 1)
    assert np.array_equal(calc.witness_payload(1), ref) and calc.emit_selfcheck_result()["first_bad_wire"] is None
    # the REDUCED (O1-style) witness is checked too (round 5): every site whose wires are all kept, at their ranks, through one window and through nine; the same poke
    # is caught in what a prover built at circom's default level would consume
    from proof_of_burn_amd.circuit_model.o1 import reduce_map
    from proof_of_burn_amd.circuit_model.circuits import circuit
    m = reduce_map(circuit("Spend(31)"))
    full = ref.reshape(-1, 32)
    red = calc.witness_payload_reduced(1, m)
    r0 = calc.emit_selfcheck_result()                    # without the alias map: only the sites whose own wires all survive
    assert np.array_equal(red.reshape(-1, 32), full[m.keep]) and r0["first_bad_wire"] is None
    calc.emit_selfcheck_alias(m)                         # with it: every site through its class representatives
    for window in (0, 30_000):
        red = calc.witness_payload_reduced(1, m, window_wires=window) if window else calc.witness_payload_reduced(1, m)
        assert np.array_equal(red.reshape(-1, 32), full[m.keep])
        r = calc.emit_selfcheck_result()
        # (one window: every IsZero / IsEqual site; nine windows: the sites whose representatives -- an operand's class may be represented far away -- share a window)
        assert r["first_bad_wire"] is None and r["checked"] > (200 if window else 3000) and r["checked"] > r0["checked"] and r["skipped"] > 0, (r0, r)
    calc.poke(cls, idx, 1, 1)
    calc.witness_payload_reduced(1, m)
    r = calc.emit_selfcheck_result()
    assert r["first_bad_wire"] is not None and wire < r["first_bad_wire"] < wire + 40_000, (wire, r)
    calc.poke(cls, idx, 1, 1)
    calc.witness_payload_reduced(1, m)
    assert calc.emit_selfcheck_result()["first_bad_wire"] is None
    calc.emit_selfcheck(False)
    calc.close()


def test_inorder_schedule_equals_the_track_schedule(pkg):
    """pob_set_inorder: the calculator's whole generation and evaluation in dependency order on ONE stream (the units of every track that are ready at the same depth
    of the stage graph in one launch per kernel class).  The shim executes launches one after the other in enqueue order -- which is exactly what an in-order stream
    does -- so a level that ran ahead of something it reads would show here: reference outputs, evaluator clean, payload == the oracle's, for Spend(31), the fixture
    instantiation (all tracks) and a Keccak gadget main"""
    for main, suite, mode in (("Spend(31)", "test_spend", 1), ("Spend(31)", "test_spend", 3), ("Spend(31)", "test_spend", 5), (POB_FIX, "test_proof_of_burn", 1), (POB_FIX, "test_proof_of_burn", 3),
                              (POB_FIX, "test_proof_of_burn", 7)):
        # mode 3: FUSED launches (the Poseidon blocks with the header's sponge chain, slices of the round expansion with the levels behind it, the round evaluation interleaved
        # with the wide evaluation families, the chain evaluation behind the narrow ones): the same wires from fewer, fuller launches
        # mode | 4: the round blocks evaluated by the launch that writes them (k_rounds_gc); the evaluation proper skips its round kernel -- unless a poke came in between
        s = _suite(suite)
        calc = pkg.WitnessCalculator(main, max_batch=8)
        calc.set_inorder(mode)
        for rnd in range(2):                                # (twice: the second batch reuses every buffer)
            res = calc.calculate([c["input"] for c in s["cases"]], check=True)
            assert [r.outputs if r.ok else None for r in res] == [c["expected"] for c in s["cases"]]
            assert all(r.check_status == 0 and r.bad_wire is None for r in res if r.ok)
        i = next(k for k, r in enumerate(res) if r.ok)
        assert np.array_equal(calc.witness_payload(i), O.run(main, s["cases"][i]["input"]).witness_numpy())
        # the merged evaluation kernels see a corrupted vector like the per-family ones: one stored value of each class of witness i poked, flagged, restored
        sizes = calc.class_sizes()
        for cls in (EC.BIT, EC.SM, EC.FR):
            idx = sizes[cls] // 3
            calc.poke(cls, idx, i, 1 if cls == EC.BIT else 4)
            calc.constraint_check()
            flagged = [r.bad_wire is not None or r.check_status != 0 for r in calc.results(with_check=True)]
            calc.poke(cls, idx, i, 1 if cls == EC.BIT else 4)
            assert flagged[i] and not [k for k, f in enumerate(flagged) if f and res[k].ok and k != i], (main, cls, idx, flagged)
        calc.constraint_check()
        calc.set_inorder(False)
        res2 = calc.calculate([c["input"] for c in s["cases"]], check=True)
        assert [(r.status, r.outputs, r.check_status, r.bad_wire) for r in res2] == [(r.status, r.outputs, r.check_status, r.bad_wire) for r in res]
        calc.close()
    from tests import gadget_cases as GC
    by_name = {s["main"]: s for s in GC.gadget_suites()}
    s = by_name["PublicCommitment(2)"] if "PublicCommitment(2)" in by_name else next(v for k, v in by_name.items() if k.startswith("PublicCommitment"))
    for mode in (1, 3):
        calc = pkg.WitnessCalculator(s["main"], max_batch=len(s["cases"]))
        calc.set_inorder(mode)
        res = calc.calculate([c["input"] for c in s["cases"]], check=True)
        assert [r.outputs if r.ok else None for r in res] == [c["expected"] for c in s["cases"]]
        assert all(r.check_status == 0 and r.bad_wire is None for r in res if r.ok)
        calc.close()


def test_evaluation_riding_with_the_generation_flags_a_corrupted_store(pkg):
    """pob_set_inorder(... | 4): k_rounds_gc writes a round block and evaluates it from what it LOADS back; the input rows (k_inputs MODE 2) likewise.  A store that reaches
    memory corrupted while the generating wavefront goes on with the right value (pob_debug_store_fault) must be flagged for exactly the witnesses of the mask -- at the round
    block's first wire or the input's wire -- by the generation's own launches (the evaluation that follows runs neither kernel), and the next, clean generation reports nothing"""
    s = _suite("test_spend")
    ok_case = next(c for c in s["cases"] if c["expected"] is not None)
    calc = pkg.WitnessCalculator("Spend(31)", max_batch=5)
    calc.set_inorder(5)
    nbit = int(calc.info.n_bit)
    rng = np.random.default_rng(11)
    tried, kinds = 0, set()
    for bit_index in rng.integers(0, nbit, 400).tolist():
        lanes = int(rng.integers(1, 32))                     # any subset of the 5 witnesses
        want = calc.store_fault(bit_index, lanes)
        if want == calc.UNKNOWN_WIRE:
            calc.store_fault(bit_index, 0)                   # (a word outside the round blocks: the G units' part below; disarm)
            continue
        res = calc.calculate([ok_case["input"]] * 5, check=True)
        assert [r.bad_wire for r in res] == [want if (lanes >> k) & 1 else None for k in range(5)], (bit_index, lanes, want, [r.bad_wire for r in res])
        kinds.add(want)
        tried += 1
        if tried >= 30:
            break
    assert tried >= 30 and len(kinds) >= 10                   # many different round blocks
    res = calc.calculate([ok_case["input"]] * 5, check=True)
    assert all(r.ok and r.bad_wire is None and r.check_status == 0 for r in res)
    # a generation whose verdict nobody collects does not leak into the next batch's records
    assert calc.store_fault(bit_index, 1) is not None
    calc.calculate([ok_case["input"]] * 5, check=False)
    res = calc.calculate([ok_case["input"]] * 5, check=True)
    assert all(r.ok and r.bad_wire is None and r.check_status == 0 for r in res)
    calc.close()
    # the input rows of a ProofOfBurn main: compared with the inputs by the launch that writes them (k_inputs MODE 2).  A corrupted row is READ by the generation that follows
    # (the witness built on it may fail its asserts): what is asserted is the evaluator's verdict -- the input's wire, for exactly the witnesses of the mask
    s = _suite("test_proof_of_burn")
    ok_case = next(c for c in s["cases"] if c["expected"] is not None)
    calc = pkg.WitnessCalculator(POB_FIX, max_batch=3)
    calc.set_inorder(7)
    first = next(i for i in range(int(calc.info.n_sm)) if calc.store_fault(i, 0, cls=EC.SM) != calc.UNKNOWN_WIRE)
    nrows = next(k for k in range(1, int(calc.info.n_sm)) if calc.store_fault(first + k, 0, cls=EC.SM) == calc.UNKNOWN_WIRE)
    assert nrows > 2000                                      # layers[4][544] + blockHeader[680] + the scalars
    for k in sorted(set([0, nrows - 1] + rng.integers(0, nrows, 6).tolist())):
        lanes = int(rng.integers(1, 8))
        want = calc.store_fault(first + k, lanes, cls=EC.SM)
        res = calc.calculate([ok_case["input"]] * 3, check=True)
        assert [r.bad_wire for r in res] == [want if (lanes >> j) & 1 else None for j in range(3)], (k, lanes, want, [r.bad_wire for r in res])
    res = calc.calculate([ok_case["input"]] * 3, check=True)
    assert all(r.ok and r.bad_wire is None and r.check_status == 0 for r in res)
    # the G units' own stores (policy.hpp GenPT<true, true>, poseidon_wide.hpp): words drawn from every storage class.  Where a riding unit stores the word, exactly the
    # witnesses of the mask are flagged (the wire is the corrupted word's own: a later unit that READS the corrupted word builds on it consistently); a word of the sponge
    # chains, of the RLP units (plain policy: their evaluation is a launch of pob_constraint_check, which then sees a vector whose defining unit wrote something else) or of no
    # unit at all flags whatever pob_constraint_check's remaining launches make of it -- never a witness outside the mask
    sizes = calc.class_sizes()
    hit = {EC.BIT: 0, EC.SM: 0, EC.FR: 0}
    for cls in (EC.BIT, EC.SM, EC.FR):
        for idx in rng.integers(0, min(sizes[cls], 30000) if cls == EC.BIT else sizes[cls], 40).tolist():      # (BIT: the first ranks are G-unit wires; 97 % of the class are Keccak's)
            lanes = int(rng.integers(1, 8))
            want = calc.store_fault(idx, lanes, cls=cls)
            if want != calc.UNKNOWN_WIRE:
                calc.store_fault(idx, 0, cls=cls)            # (a round block's word / an input row: covered above)
                continue
            res = calc.calculate([ok_case["input"]] * 3, check=True)
            flagged = [j for j, r in enumerate(res) if r.bad_wire is not None]
            assert all((lanes >> j) & 1 for j in flagged), (cls, idx, lanes, [(r.bad_wire, r.check_status) for r in res])
            hit[cls] += flagged == [j for j in range(3) if (lanes >> j) & 1]
    assert hit[EC.BIT] >= 8 and hit[EC.SM] >= 15 and hit[EC.FR] >= 25, hit
    res = calc.calculate([ok_case["input"]] * 3, check=True)
    assert all(r.ok and r.bad_wire is None and r.check_status == 0 for r in res)
    calc.close()


def test_riding_evaluation_gives_the_records_of_the_evaluation_pass_on_the_mutation_set(pkg):
    """the reference's mutation set on the fixture instantiation (valid and failing inputs alike): an in-order calculator whose evaluation rides with its generation
    (pob_set_inorder(h, 7): Keccak round blocks, input rows, every G unit but the RLP family) reports, witness by witness, the records of the track schedule's separate
    evaluation pass -- status, outputs, the evaluator's verdict (check_status: the same failing site) and bad_wire (none: every vector is consistent)"""
    import random
    from proof_of_burn_amd import inputs as gen
    from tests.test_gpu_parity import _mutations
    params = (4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)
    base = gen.synthetic_batch(1, depth=2, seed=101, distinct_keys=1, params=params).inputs[0]
    cases = _mutations(base, random.Random(5))
    ref = pkg.WitnessCalculator(POB_FIX, max_batch=len(cases))
    want = [(r.status, r.outputs, r.check_status, r.bad_wire) for r in ref.calculate([c[1] for c in cases], check=True)]
    ref.close()
    assert sum(1 for w in want if w[0] != 0) > 20 and sum(1 for w in want if w[0] == 0) > 5
    calc = pkg.WitnessCalculator(POB_FIX, max_batch=len(cases))
    calc.set_inorder(7)
    for _ in range(2):
        got = [(r.status, r.outputs, r.check_status, r.bad_wire) for r in calc.calculate([c[1] for c in cases], check=True)]
        assert got == want, [(cases[k][0], g, w) for k, (g, w) in enumerate(zip(got, want)) if g != w][:4]
    calc.close()


def test_failing_status_is_the_lowest_of_all_the_oracles_failing_sites(pkg):
    """the emitted calculator stops at its FIRST failing assert (tests/test.py:65-68: anything on stderr = failed); the device reports, per witness, the LOWEST failing
    site code.  The oracle keeps going after a failure and lists every failing site (oracle_fail_sites): for the reference's mutation set on the fixture instantiation the
    device's status EQUALS the lowest of them -- same verdicts, and the same site whenever the first failing site is the lowest one"""
    import random
    from proof_of_burn_amd import inputs as gen, witness as W
    from tests.test_gpu_parity import _mutations
    params = (4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)
    base = gen.synthetic_batch(1, depth=2, seed=101, distinct_keys=1, params=params).inputs[0]
    cases = _mutations(base, random.Random(5))
    calc = pkg.WitnessCalculator(POB_FIX, max_batch=len(cases))
    res = calc.calculate([c[1] for c in cases])
    code_of = {name: tid for tid, name in W._TPL.items()}
    n_eq = n_first = 0
    for (label, inp), r in zip(cases, res):
        ora = O.run(POB_FIX, inp)
        sites = O.fail_sites()
        assert ora.failed == (not r.ok), label
        if ora.failed and r.status != W.FAIL_INPUT_RANGE:
            assert sites and all(t in code_of for t, _ in sites), (label, sites)
            codes = [(code_of[t] << 12) | ln for t, ln in sites]
            assert r.status == min(codes), f"{label}: {r.message()} vs the oracle's failing sites {sites}"
            n_eq += 1
            n_first += codes[0] == min(codes)
    assert n_eq > 20 and 0 < n_first < n_eq          # both kinds occur: the first site is the lowest one / it is not
    calc.close()
