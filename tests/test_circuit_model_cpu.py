"""The symbolic circuit model (proof_of_burn_amd/circuit_model): a third, independent statement of the O0 wire numbering plus every
`<==` / `===` as a rank-1 constraint.  It referees the witness generators: the CPU oracle's witness and the product's witness (the
HIP kernels, here on the CPU shim) must satisfy every row; a corrupted wire must fail exactly at rows that touch it."""
import hashlib
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from proof_of_burn_amd import plan_info
from proof_of_burn_amd.circuit_model import P, check as CK, circuit
from tests import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POB_FIX = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"


def _suite(name):
    with open(os.path.join(ROOT, "tests", "golden", "suites.json")) as f:
        return next(s for s in json.load(f) if s["name"] == name)


def test_model_wire_counts_equal_the_planner():
    """three statements of the layout (HIP planner, oracle, model) agree on nWitness"""
    for main, w in (("Spend(31)", 2_603_360), (POB_FIX, 64_355_038)):
        c = circuit(main)
        assert c.n_wires == w == int(plan_info(main).n_witness)


def test_oracle_witness_satisfies_every_constraint_and_pokes_fail_locally():
    s = _suite("test_spend")
    c = circuit("Spend(31)")
    assert c.n_constraints == 2_605_281 and c.n_outputs == 1 and c.n_inputs == 4
    ora = O.run("Spend(31)", s["cases"][0]["input"])
    pay = ora.witness_numpy().copy()
    assert CK.check_witness(c, CK.Witness(pay)) == []
    rng = np.random.default_rng(3)
    # one wire of every kind: main output, a Poseidon state wire, a Keccak round gate, a byte, a selector flag
    for w in [1, 7, 5000, 123_456, 1_500_000, 2_603_000] + rng.integers(1, c.n_wires, 6).tolist():
        p2 = pay.copy()
        p2[32 * w] ^= 1
        bad = CK.check_witness(c, CK.Witness(p2))
        if not bad:      # an unconstrained wire (TruncatedAddressHash.temp is the only kind, not in Spend) or IsZero.inv of a zero operand
            v = int.from_bytes(pay[32 * w:32 * w + 32].tobytes(), "little")
            assert v == 0, f"poke of wire {w} (value {v}) went unnoticed"
            continue
        assert all(w in wires for _, wires in bad), (w, bad[:3])       # only rows that touch the poked wire fail


def test_fixture_proof_of_burn_oracle_witness_satisfies_every_constraint():
    s = _suite("test_proof_of_burn")
    c = circuit(POB_FIX)
    ora = O.run(POB_FIX, s["cases"][0]["input"])
    assert CK.check_witness(c, CK.Witness(ora.witness_numpy().copy())) == []
    # a mutated input that FAILS the circuit's asserts leaves a witness that violates constraints
    bad_case = next(cs for cs in s["cases"] if cs["expected"] is None)
    ora = O.run(POB_FIX, bad_case["input"])
    assert ora.failed and CK.check_witness(c, CK.Witness(ora.witness_numpy().copy())) != []


def test_product_witness_on_the_cpu_shim_satisfies_every_constraint(tmp_path):
    """the HIP generator's own .wtns (kernels + host scheduler run on tests/hostsim) through the `check` CLI"""
    from tests.hostsim import build as hb
    from proof_of_burn_amd import witness as W
    lib = hb.build()
    old = (W.LIB_PATH, W._lib)
    W.LIB_PATH, W._lib = lib, None
    try:
        s = _suite("test_spend")
        calc = W.WitnessCalculator("Spend(31)", max_batch=1)
        assert calc.calculate(s["cases"][3]["input"])[0].ok
        path = str(tmp_path / "spend.wtns")
        calc.write_wtns(0, path)
        from proof_of_burn_amd.circuit_model.o1 import reduce_map
        m = reduce_map(circuit("Spend(31)"))
        red = str(tmp_path / "spend_o1.wtns")
        calc.write_wtns_reduced(0, red, m)                         # expanded and cut on the device (pob_write_wtns_reduced)
        red_windows = calc.witness_payload_reduced(0, m, window_wires=30_000)       # the same through 9 windows of kept wires
        full_again = calc.witness_payload(0)                       # and the O0 payload after a reduced emission (the probe tables are per map)
        calc.close()
    finally:
        W.LIB_PATH, W._lib = old
    r = subprocess.run([sys.executable, "-m", "proof_of_burn_amd.circuit_model", "check", "Spend(31)", path], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr
    # O1-style reduced witness: header says len(keep) wires, values = the kept wires of the full witness, aliases really are equal
    full = np.fromfile(path, dtype=np.uint8)[76:].reshape(-1, 32)
    data = np.fromfile(red, dtype=np.uint8)
    assert struct.unpack("<I", data[60:64].tobytes())[0] == len(m.keep) and len(m.keep) < full.shape[0] // 8
    assert np.array_equal(data[76:].reshape(-1, 32), full[m.keep])
    assert np.array_equal(red_windows.reshape(-1, 32), full[m.keep]) and np.array_equal(full_again.reshape(-1, 32), full)
    dropped = np.nonzero(m.alias >= 0)[0]
    assert np.array_equal(full[dropped], full[m.alias[dropped]])
    for w, v in list(zip(m.const_wires, m.const_values))[:2000]:
        assert int.from_bytes(full[w].tobytes(), "little") == v


def test_sym_and_r1cs_exports(tmp_path):
    c = circuit("Spend(31)")
    path = str(tmp_path / "spend.sym")
    digest = c.write_sym(path)
    with open(path) as f:
        lines = f.read().splitlines()
    assert len(lines) == c.n_wires - 1 and hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest() == digest
    assert lines[0] == "1,1,0,main.commitment" and lines[1] == "2,2,0,main.burnKey" and lines[4] == "5,5,0,main.extraCommitment"
    assert lines[5] == "6,6,0,main.coin" and lines[7].endswith("main.coinBytes[0]")
    first_child = next(ln for ln in lines if ln.split(",")[2] == "1")
    assert first_child.endswith("main.AssertGreaterEqThan_41.a")
    with open(os.path.join(ROOT, "tests", "golden", "spend31_sym.sha256")) as f:
        assert f.read().split()[0] == digest, "the .sym of Spend(31) changed: regenerate tests/golden/spend31_sym.sha256 if that is intended"
    # .r1cs: header + constraint count of a small template, parsed back
    from proof_of_burn_amd.circuit_model.core import Circuit
    from proof_of_burn_amd.circuit_model.lib import Poseidon
    small = Circuit(Poseidon.get(2))
    rp = str(tmp_path / "poseidon2.r1cs")
    small.write_r1cs(rp)
    data = open(rp, "rb").read()
    assert data[:4] == b"r1cs" and struct.unpack("<II", data[4:12]) == (1, 3)
    sec, ln = struct.unpack("<IQ", data[12:24])
    assert sec == 1 and int.from_bytes(data[28:60], "little") == P
    n_wires, n_out, n_pub, n_prv, n_lab, n_con = struct.unpack("<IIIIQI", data[60:60 + 28])
    assert (n_wires, n_out, n_prv, n_con) == (small.n_wires, 1, 2, small.n_constraints)
