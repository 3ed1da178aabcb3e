"""GPU parity tests (run on the MI355X: `pytest -m gpu`).  Everything goes through the C ABI (libpob_hip.so);
the CPU oracle is only the checker.  Bit-exact bar: public outputs, failure sets and the FULL canonical witness
payload (every one of the W 32-byte wires) must equal the oracle's."""
import json
import os

import numpy as np
import pytest

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
    calc.close()
