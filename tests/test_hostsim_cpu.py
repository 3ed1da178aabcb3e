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
    missed, done = EC.uniform_sweep(calc, {EC.SM: 126, EC.SB: 126, EC.FR: 126, EC.BIT: 126})
    assert not missed, f"{len(missed)} mis-detections of {done}: {missed[:6]}"
    bad = EC.named_pokes(calc, [("poseidon", 200), ("poseidon", 640), ("pad.div.out", 0), ("pad.div.rem", 0), ("pad.iseq.inv", 0), ("commitment", 0)])
    assert not bad, bad
    calc.close()


def test_evaluator_detects_pokes_fixture(pkg):
    s = _suite("test_proof_of_burn")
    calc = EC.open_identical_batch(pkg, POB_FIX, s["cases"][0]["input"])
    missed, done = EC.uniform_sweep(calc, {EC.SM: 126, EC.SB: 63, EC.FR: 126})
    assert not missed, f"{len(missed)} mis-detections of {done}: {missed[:6]}"
    bad = EC.named_pokes(calc, [("poseidon", 300), ("sc.M", 17), ("sc.exists", 5), ("sc.isz.inv", 40), ("pad.div.out", 1), ("pad.iseq.inv", 2)])
    assert not bad, bad
    calc.close()
