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


@pytest.mark.parametrize("env", [{"POB_CHECK_EARLY_K": "1"}, {"POB_CHECK_PLAN": "2,K;1,7,5;3,6,0;4"}, {"POB_LONG_SPONGE": "4", "POB_KCHK_NT": "0"},
                                 {"POB_EMIT_PROBE": "0", "POB_EMIT_FILL": "0"}])
def test_scheduling_switches_keep_the_results(pkg, tmp_path, env):
    """the optional schedules (early Keccak evaluation, other stream plans, long sponges on their own stream, emitter without the
    probe pass) are the same work in a different order: fixture instantiation, outputs + clean evaluator + a detected poke + the
    first 200 000 wires of the payload, in a fresh process with the switch set (the library reads them once)"""
    from tests.hostsim import build as hb
    script = tmp_path / "w.py"
    script.write_text(f"""
import json, os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np
from proof_of_burn_amd import witness as W
from tests import oracle_ffi as O
W.LIB_PATH, W._lib = {hb.build()!r}, None
s = next(x for x in json.load(open(os.path.join({ROOT!r}, "tests", "golden", "suites.json"))) if x["name"] == "test_proof_of_burn")
calc = W.WitnessCalculator(s["main"], max_batch=8)
res = calc.calculate([c["input"] for c in s["cases"]], check=True)
assert [r.outputs if r.ok else None for r in res] == [c["expected"] for c in s["cases"]]
assert all(r.check_status == 0 and r.bad_wire is None for r in res if r.ok)
cls, idx, wire = calc.debug_ref("sc.M", 9)
calc.poke(cls, idx, 3, 2); calc.constraint_check(); r2 = calc.results(with_check=True); calc.poke(cls, idx, 3, 2)
assert r2[3].bad_wire is not None and r2[0].bad_wire is None
ref = O.run(s["main"], s["cases"][0]["input"]).witness_numpy()
for w0, v in calc.witness_windows(0, 200_000):
    assert np.array_equal(v, ref[32 * w0:32 * w0 + v.size]); break
print("ok")
""")
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
