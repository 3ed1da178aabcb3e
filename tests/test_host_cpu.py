"""CPU-side tests of the product: the C-ABI library loads and exports every symbol of include/pob_hip.h, the host
planner's layout (written independently of the oracle) yields the oracle's wire counts, input packing mirrors the
emitted loader.  No compute call is made without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import proof_of_burn_amd as pkg
from proof_of_burn_amd import witness as W
from tests import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(pkg.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "pob_hip.h")).read()
    declared = set(re.findall(r"\b(pob_\w+)\s*\(", hdr))
    assert declared == set(pkg.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None


@pytest.mark.parametrize("main,w", [
    ("Spend(31)", 2_603_360),
    ("ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)", 64_355_038),
    ("ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)", 215_907_954),
])
def test_planner_wire_count_matches_model(main, w):
    info = pkg.plan_info(main)
    assert info.n_witness == w                              # SURVEY.md app. C / BASELINE.md section 2
    assert info.n_bit + info.n_sm + info.n_fr + info.n_derived + info.n_alias + 1 == w      # every wire is stored in exactly one class, derived, or an alias (+ the constant wire)


def test_planner_matches_oracle_on_other_instantiations():
    for main in ("ProofOfBurn(2, 1, 1, 20, 31, 2, 10 ** 18, 10 ** 19)", "ProofOfBurn(3, 2, 2, 20, 30, 2, 10 ** 18, 10 ** 19)", "Spend(16)"):
        name, prm = pkg.parse_main(main)
        if name == "ProofOfBurn":
            L, NB, HB = prm[:3]
            inp = {k: 0 for k in W.POB_FR_INPUTS + ["numLeafAddressNibbles", "numLayers", "blockHeaderLen", "byteSecurityRelax"]}
            inp.update(layers=[[0] * (136 * NB)] * L, layerLens=[0] * L, blockHeader=[0] * (136 * HB))
        else:
            inp = {k: 0 for k in W.SPEND_FR_INPUTS}
        assert O.run(main, inp).nwitness == pkg.plan_info(main).n_witness, main


def test_host_keccak_matches_reference_constant():
    with open(os.path.join(ROOT, "tests", "golden", "test_pob_input.json")) as f:
        pob = json.load(f)
    hdr = bytes(pob["blockHeader"][:pob["blockHeaderLen"]])
    # the one Keccak-256 value hard-coded in the reference (tests/testcases/proof_of_burn.py:22)
    assert pkg.keccak256(hdr).hex() == "e36499b50da290131c3fa32d4f60717c8c529ae1bc3a216f32d05c05fe80368d"
    assert pkg.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"


def test_parse_main_and_values():
    assert pkg.parse_main("ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)") == ("ProofOfBurn", [16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20])
    assert W.to_field("0x10") == 16 and W.to_field(str(W.P + 5)) == 5 and W.to_field(2 ** 256 - 1) == (2 ** 256 - 1) % W.P
    assert W._scalar([7]) == 7


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pkg.WitnessCalculator("Spend(31)", max_batch=1)     # hipSetDevice/hipMalloc fail: no CPU fallback exists


def test_wtns_header_equals_oracle_writer():
    r = O.run("Poseidon(2)", {"inputs": [1, 2]})
    assert bytes(r.wtns_numpy()[:76]) == pkg.wtns_header(r.nwitness)


def test_plan_rejects_unsupported_and_oversized_instantiations():
    """template parameters are validated limb-exactly and an instantiation whose wire / class counts would overflow the layout's
    32-bit offsets is refused up front instead of wrapping silently (C ABI: POB_E_ARG; here: ValueError from plan_info)"""
    from proof_of_burn_amd import plan_info
    ok = plan_info("ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)")
    assert int(ok.n_witness) == 215_907_954 and int(ok.n_bit) < (1 << 29)
    for bad in ("ProofOfBurn(32, 8, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",           # 276 Keccak-f permutations: ~690 M BIT wires
                "ProofOfBurn(64, 16, 32, 50, 31, 2, 10 ** 19, 10 ** 20)",
                f"ProofOfBurn({2 ** 64 + 16}, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",     # upper limbs must be zero
                "ProofOfBurn(16, 4, 16, 500, 31, 2, 10 ** 19, 10 ** 20)",                  # minLeafAddressNibbles > 64
                "ProofOfBurn(16, 4, 16, 50, 32, 2, 10 ** 19, 10 ** 20)",                   # amountBytes > 31
                "ProofOfBurn(16, 4, 16, 50, 31, 99, 10 ** 19, 10 ** 20)",                  # powMinimumZeroBytes > 32
                "ProofOfBurn(1, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",
                "Spend(32)", "Spend(0)"):
        with pytest.raises(ValueError):
            plan_info(bad)
    # the balance bounds are field elements: values >= p are reduced, not truncated
    assert int(plan_info(f"ProofOfBurn(4, 4, 5, 20, 31, 2, {W.P + 10 ** 18}, 10 ** 19)").n_witness) == 64_355_038


def _loader_cases():
    """inputs of the fixture instantiation that exercise the loader's acceptance (reference tests/testcases/divide.py:4 one-element arrays,
    convert.py:36 p - 1 as a string, rlp/integer.py:51-53 and assertion.py:87 out-of-range values, tests/main.py:160-178 the producer's mix of
    strings and numbers)"""
    import copy
    with open(os.path.join(ROOT, "tests", "golden", "test_pob_input.json")) as f:
        base = json.load(f)
    out = [("as produced by tests/main.py", base)]

    def mut(label, fn):
        d = copy.deepcopy(base)
        fn(d)
        out.append((label, d))
    mut("scalars as one-element arrays", lambda d: d.update(numLayers=[d["numLayers"]], burnKey=[[d["burnKey"]]], blockHeaderLen=[d["blockHeaderLen"]]))
    mut("hex and signed strings", lambda d: d.update(revealAmount=hex(int(d["revealAmount"])), numLayers="+%d" % d["numLayers"], byteSecurityRelax=" 0 "))
    mut("p - 1 and > p", lambda d: d.update(burnExtraCommitment=str(W.P - 1), _proofExtraCommitment=str(W.P + 7), burnKey=2 ** 255 + 12345))
    mut("negative scalar", lambda d: d.update(_proofExtraCommitment=-5, intendedBalance="-3"))
    mut("booleans", lambda d: d.update(byteSecurityRelax=False, _proofExtraCommitment=True))
    mut("byte = 256 / 10000 / 2^31 / 2^40 / negative / string", lambda d: (d["layers"][0].__setitem__(3, 256), d["layers"][1].__setitem__(0, 10000), d["layers"][2].__setitem__(9, 2 ** 31),
                                                                           d["layers"][3].__setitem__(5, 2 ** 40), d["blockHeader"].__setitem__(7, -1), d["blockHeader"].__setitem__(8, "17")))
    mut("small input >= 2^31", lambda d: d.update(numLayers=2 ** 31))
    mut("small input = p - 1", lambda d: d.update(numLeafAddressNibbles=str(W.P - 1)))
    mut("layerLens as strings", lambda d: d.update(layerLens=[str(x) for x in d["layerLens"]]))
    mut("flat layers", lambda d: d.update(layers=[x for row in d["layers"] for x in row]))
    return out


def test_native_json_loader_equals_the_python_loader():
    """pob_pack_json_batch (hand-written parser on host threads) against WitnessCalculator.pack on the parsed dicts: same rows, same
    forced-failure words, same refusals -- for the fixture's input.json, the loader's edge cases, Spend and a synthetic production batch"""
    from proof_of_burn_amd import inputs as gen
    fix = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"
    cases = _loader_cases()
    texts = [json.dumps(d) for _, d in cases]
    py = W.pack_inputs(fix, [d for _, d in cases])
    for threads in (1, 3):
        nat = W.pack_json(fix, texts, threads=threads)
        for x, y, what in zip(py, nat, ("fr", "sm", "forced")):
            bad = [cases[i][0] for i in range(len(cases)) if not np.array_equal(x[i], y[i])]
            assert not bad, (what, bad)
    assert py[2][0] == 0 and py[2][6] == W.FAIL_INPUT_RANGE and py[2][7] == W.FAIL_INPUT_RANGE     # out-of-range values fail up front, they do not wrap
    # the file on disk, byte for byte (whitespace and key order as written by the reference's producer)
    with open(os.path.join(ROOT, "tests", "golden", "test_pob_input.json"), "rb") as f:
        raw = f.read()
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(py, W.pack_json(fix, [raw])))
    # refusals: both loaders reject, neither truncates
    base = cases[0][1]
    for label, fn, exc in (("fraction", lambda d: d["layers"][0].__setitem__(0, 1.7), (TypeError, ValueError)),
                           ("exponent scalar", lambda d: d.update(numLayers=2e0), (TypeError, ValueError)),
                           ("null", lambda d: d.update(numLayers=None), (TypeError, ValueError)),
                           ("garbage string", lambda d: d.update(burnKey="12x"), ValueError),
                           ("missing key", lambda d: d.pop("numLayers"), KeyError),
                           ("extra key", lambda d: d.update(bogus=1), KeyError),
                           ("short array", lambda d: d["layerLens"].pop(), ValueError),
                           ("two-element scalar", lambda d: d.update(numLayers=[1, 2]), (TypeError, ValueError))):
        import copy
        d = copy.deepcopy(base)
        fn(d)
        with pytest.raises(exc):
            W.pack_inputs(fix, [d])
        with pytest.raises((KeyError, ValueError)):
            W.pack_json(fix, [json.dumps(d)])
    # texts json.loads itself refuses or resolves: a number with leading zeros is refused (the STRING "007" is 7 for both); a duplicate key
    # counts with its LAST value only -- an out-of-range first value must not leave the witness marked as failed; hostile nesting is an
    # input error, not a stack overflow of the loader thread
    good = json.dumps(base)
    assert '"numLayers": ' in good
    nl = base["numLayers"]
    with pytest.raises(ValueError):
        json.loads(good.replace(f'"numLayers": {nl}', '"numLayers": 007'))
    with pytest.raises((KeyError, ValueError)):
        W.pack_json(fix, [good.replace(f'"numLayers": {nl}', '"numLayers": 007')])
    with pytest.raises((KeyError, ValueError)):
        W.pack_json(fix, [good.replace(f'"layerLens": [', '"layerLens": [00, ', 1)])
    assert np.array_equal(W.pack_json(fix, [good.replace(f'"numLayers": {nl}', f'"numLayers": "00{nl}"')])[1], py[1][:1])
    dup = good[:-1] + f', "numLayers": {nl}' + "}"
    dup = dup.replace(f'"numLayers": {nl}', f'"numLayers": {2 ** 40}', 1)                     # first occurrence out of range, last one valid
    assert json.loads(dup)["numLayers"] == nl
    nat = W.pack_json(fix, [dup])
    assert all(np.array_equal(x[:1], y) for x, y in zip(py, nat)) and nat[2][0] == 0
    dup2 = good[:-1] + f', "numLayers": {2 ** 40}' + "}"                                        # ... and the other way round
    assert W.pack_json(fix, [dup2])[2][0] == W.FAIL_INPUT_RANGE == W.pack_inputs(fix, [json.loads(dup2)])[2][0]
    for hostile in ('{"layers": ' + "[" * 200000, '{"numLayers": ' + "[" * 200000):
        with pytest.raises((KeyError, ValueError)):
            W.pack_json(fix, [hostile])
    # Spend (no small inputs) and the production shape
    with open(os.path.join(ROOT, "tests", "golden", "test_spend_input.json")) as f:
        sp = json.load(f)
    assert all(np.array_equal(x, y) for x, y in zip(W.pack_inputs("Spend(31)", [sp, sp]), W.pack_json("Spend(31)", [json.dumps(sp)] * 2)))
    main = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
    batch = gen.synthetic_batch(6, depth=10, seed=0xB0B, distinct_keys=2)
    assert all(np.array_equal(x, y) for x, y in zip(W.pack_inputs(main, batch.inputs), W.pack_json(main, [json.dumps(d) for d in batch.inputs])))


def test_native_json_loader_array_paths_fuzz():
    """the loader's array fast paths (64-byte mask stage, zero-run skip, one-element loop, general path) against json.loads + the Python loader on texts assembled
    element by element: every mix of separators (", " | "," | " ," | newlines), zero runs of every length across the 48- / 24- / 16-byte skips, numbers of 1-12
    digits, strings, negatives, nested rows -- and the forms json.loads refuses (leading zeros, fractions, exponents, a trailing comma, two commas): refused by both"""
    import copy
    import random
    fix = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"
    with open(os.path.join(ROOT, "tests", "golden", "test_pob_input.json")) as f:
        base = json.load(f)
    rng = random.Random(0xA77A)
    nlay, nrow = len(base["layers"]), len(base["layers"][0])

    def element(kind, bad_ok):
        r = rng.random()
        if not bad_ok and r >= 0.96:
            r = 0.0
        if kind == "bytes" or r < 0.80:
            return str(rng.choice((0, 0, 0, rng.randrange(10), rng.randrange(100), rng.randrange(256))))
        if r < 0.86:
            return str(rng.choice((256, 999, 1000, 54321, 2 ** 31 - 1, 2 ** 31, 10 ** 11 + 7)))
        if r < 0.90:
            return '"%d"' % rng.randrange(300)
        if r < 0.93:
            return str(-rng.randrange(1, 300))
        if r < 0.96:
            return rng.choice(("true", "false"))
        if rng.random() < 0.98:
            return "0"
        return rng.choice(("00", "01", "1.5", "2e1", "1E2", "-0", " 7", "0x10", ""))           # most of these are not JSON (or not integers)

    def text_of(kind, seps, zero_run):
        rows = []
        bad_ok = kind != "bytes" and rng.random() < 0.5
        for _ in range(nlay):
            el = [element(kind, bad_ok) for _ in range(nrow)]
            if zero_run:
                a = rng.randrange(nrow - zero_run); el[a:a + zero_run] = ["0"] * zero_run
            row = ""
            for i, x in enumerate(el):
                row += x + (rng.choice(seps) if i + 1 < nrow else "")
            rows.append("[" + row + rng.choice(("", " ", "", "", ",", "")) + "]" if kind == "hostile" else "[" + row + "]")
        flat_form = kind != "bytes" and rng.random() < 0.2
        body = ("[" + ", ".join(r[1:-1] for r in rows) + "]") if flat_form else "[" + rng.choice((", ", ",", ",\n ")).join(rows) + "]"
        d = copy.deepcopy(base); d["layers"] = "@@"
        return json.dumps(d).replace('"@@"', body)
    n_ok = n_refused = 0
    for it in range(400):
        kind = ("bytes", "mixed", "hostile")[it % 3]
        seps = [(", ",), (",",), (", ", ","), (", ", " ,", ",  ", ",\n", ", \t", " , ")][it % 4]
        zero_run = rng.choice((0, 0, 1, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100))
        text = text_of(kind, seps, zero_run)
        try:
            want = W.pack_inputs(fix, [json.loads(text)])
        except (ValueError, TypeError, KeyError):
            want = None
        try:
            got = W.pack_json(fix, [text])
        except (ValueError, KeyError):
            got = None
        if want is None:
            assert got is None, f"text {it}: json.loads / the Python loader refuse it, the native loader accepted it"
            n_refused += 1
            continue
        assert got is not None, f"text {it} ({kind}, {seps}): refused by the native loader only"
        assert all(np.array_equal(x, y) for x, y in zip(want, got)), f"text {it} ({kind}, {seps}, zero run {zero_run})"
        n_ok += 1
        if kind == "bytes":
            g8 = W.pack_json8(fix, [text])
            assert g8 is not None and np.array_equal(W.widen_inputs(g8[1], g8[2]), want[1])
    assert n_ok > 150 and n_refused > 40, (n_ok, n_refused)


def test_byte_form_of_the_small_inputs():
    """pob_narrow_inputs / pob_pack_json_batch8 (round 5: bytes as bytes on the wire): narrowing then widening is the identity, a witness with more than
    EXC_CAP values outside 0..255 is refused (never truncated), the loader's byte form stands for the rows of its int32 form on the loader's edge cases and on
    production batches, and the persistent loader pool gives the same rows for any width, repeatedly and from two caller threads at once"""
    import threading
    from proof_of_burn_amd import inputs as gen
    rng = np.random.default_rng(5)
    sm = rng.integers(0, 256, size=(37, 2864), dtype=np.int32)
    for w in range(37):
        k = rng.choice(2864, size=w % (W.EXC_CAP + 1), replace=False)           # 0 .. EXC_CAP exceptions per witness, the last rows exactly EXC_CAP
        sm[w, k] = rng.choice(np.array([256, 544, 2175, -1, -200, 0x7FFFFFFF, 1 << 20], dtype=np.int32), size=k.size)
    nar = W.narrow_inputs(sm)
    assert nar is not None and np.array_equal(W.widen_inputs(*nar), sm)
    assert int((nar[1]["k"] != W.EXC_NONE).sum()) == int(((sm < 0) | (sm > 255)).sum())
    over = sm.copy(); over[5, :W.EXC_CAP + 1] = 300
    assert W.narrow_inputs(over) is None
    fix = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"
    cases = _loader_cases()
    texts = [json.dumps(d) for _, d in cases]
    ref = W.pack_json(fix, texts, threads=1)
    for threads in (1, 2, 7, 0):
        got = W.pack_json8(fix, texts, threads=threads)
        assert got is not None
        assert np.array_equal(got[0], ref[0]) and np.array_equal(W.widen_inputs(got[1], got[2]), ref[1]) and np.array_equal(got[3], ref[2]), threads
    # a text with too many out-of-range bytes: the byte form refuses, the int32 form carries it
    bad = json.loads(texts[0]); bad["layers"][0][:40] = [256] * 40
    assert W.pack_json8(fix, [texts[0], json.dumps(bad)]) is None and W.pack_json(fix, [json.dumps(bad)])[1][0, 1] == 256
    with pytest.raises(KeyError):
        W.pack_json8(fix, [texts[0].replace('"numLayers"', '"numLayerz"')])
    main = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
    batch = gen.synthetic_batch(64, depth=10, seed=0xB0B, distinct_keys=2)
    ptexts = [json.dumps(d) for d in batch.inputs]
    pref = W.pack_json(main, ptexts, threads=1)
    out = {}

    def worker(tag):
        for _ in range(3):
            g = W.pack_json8(main, ptexts, threads=0)
            out[tag] = np.array_equal(g[0], pref[0]) and np.array_equal(W.widen_inputs(g[1], g[2]), pref[1]) and np.array_equal(g[3], pref[2]) and out.get(tag, True)
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert out == {0: True, 1: True}
    g = W.pack_json8(main, ptexts)
    assert g[1].nbytes + g[2].nbytes + g[0].nbytes < 12 * 1024 * 64, "the byte form of a production witness is 11.4 KB"


def test_stored_keep_maps():
    """the compressed O1-style keep maps under circuit_model/data/ (what the device-side reduced emission of the bench and the GPU tests
    reads): the stored Spend(31) map equals the one derived from the model now; the production map has the recorded size"""
    from proof_of_burn_amd.circuit_model import circuit, keepmap
    from proof_of_burn_amd.circuit_model.o1 import reduce_map
    keep, nw = keepmap.load("Spend(31)")
    c = circuit("Spend(31)")
    assert nw == c.n_wires and np.array_equal(keep, reduce_map(c).keep.astype(np.uint32))
    keep, nw = keepmap.load("ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)")
    assert nw == 215_907_954 and keep.size == 21_454_032 and keep[0] == 0 and (np.diff(keep.astype(np.int64)) > 0).all()
    keep, nw = keepmap.load("ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)")
    assert nw == 64_355_038 and keep[0] == 0 and keep.size < nw // 8
