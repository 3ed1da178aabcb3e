"""Shared bodies of the gadget-level main tests (reference tests/test.py:146-201: `component main = T(params);` around one template).
They run on the GPU (`tests/test_gpu_parity.py`, `-m gpu`) and on the CPU through the HIP-on-fibers shim (`tests/test_hostsim_cpu.py`):
same product code either way.  The oracle (oracle/pob_oracle.c through tests/oracle_ffi.py) is the checker."""
from __future__ import annotations

import json
import os
import random

import numpy as np

from tests import evaluator_cases as EC
from tests import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CIRCUITS = ("ProofOfBurn", "Spend")
KECCAK_MAINS = ("KeccakBytes", "PublicCommitment", "BurnAddressHash", "ProofOfWorkChecker")


def gadget_suites():
    """the 54 gadget-level entries of the reference's list, as regenerated into tests/golden/suites.json"""
    with open(os.path.join(ROOT, "tests", "golden", "suites.json")) as f:
        return [s for s in json.load(f) if s["main"].split("(")[0] not in CIRCUITS]


def check_suite(pkg, s) -> list:
    """one suite through the calculator: outputs == the reference's expected values == the oracle's, the FULL canonical payload
    == the oracle's witness, the constraint evaluator clean, no witness for a failed input.  Returns the list of complaints."""
    main, cases = s["main"], s["cases"]
    bad = []
    calc = pkg.WitnessCalculator(main, max_batch=max(1, len(cases)))
    try:
        res = calc.calculate([c["input"] for c in cases], check=True)
        for i, (c, r) in enumerate(zip(cases, res)):
            got = r.outputs if r.ok else None
            ora = O.run(main, c["input"])
            oexp = None if ora.failed else ora.outputs()
            if oexp != c["expected"]:
                bad.append((main, i, "oracle disagrees with the reference's expected value", oexp, c["expected"]))
            if got != c["expected"]:
                bad.append((main, i, "outputs", got, c["expected"], hex(r.status)))
            elif r.ok:
                if r.check_status != 0 or r.bad_wire is not None:
                    bad.append((main, i, "evaluator", r.check_status, r.bad_wire))
                pay, ow = calc.witness_payload(i), ora.witness_numpy()
                if pay.size != ow.size:
                    bad.append((main, i, "nWitness", pay.size // 32, ow.size // 32))
                elif not np.array_equal(pay, ow):
                    d = np.nonzero((pay.reshape(-1, 32) != ow.reshape(-1, 32)).any(axis=1))[0]
                    bad.append((main, i, "payload differs from the oracle's at wires", d[:8].tolist()))
            else:
                try:
                    calc.witness_payload(i)
                    bad.append((main, i, "a failed input produced a witness"))
                except RuntimeError:
                    pass
    finally:
        calc.close()
    return bad


def sweep_case(s):
    """the valid case the corruption sweep runs on: one whose IsZero gadgets over field elements see non-zero operands -- IsZero.inv of a
    zero operand is a free wire (comparators.circom:30-33: out = 1 whatever inv is), so a corruption there is rightly not a violation"""
    name = s["main"].split("(")[0]
    ok = [c for c in s["cases"] if c["expected"] is not None]
    if name == "SubstringCheck":
        return next(c for c in ok if c["expected"] == [0])
    if name in ("RlpInteger", "RlpEmptyAccount"):
        key = "in" if name == "RlpInteger" else "balance"
        return next(c for c in ok if int(str(c["input"][key]), 0) > 300)
    return ok[0]


def sweep_suite(pkg, s, per_class: int = 400):
    """64 identical valid witnesses; stored values of every class are corrupted, 63 lanes per pass, and the evaluator must flag exactly the
    corrupted lanes (tests/evaluator_cases.py).  Up to per_class uniformly drawn ranks per class (all of them for the small mains)."""
    calc = EC.open_identical_batch(pkg, s["main"], sweep_case(s)["input"])
    try:
        sizes = calc.class_sizes()
        missed, done = EC.uniform_sweep(calc, {k: min(4 * v, per_class) for k, v in sizes.items()})
    finally:
        calc.close()
    return missed, done


# ---- seeded random inputs per template (beyond the reference's cases): small values around the interesting boundaries, so that
# both accepting and rejecting witnesses occur; every one is compared with the oracle (status, outputs, payload)
def _rand_inputs(name, params, rng: random.Random):
    from proof_of_burn_amd.witness import GADGET_INPUTS, P
    spec = GADGET_INPUTS[name](params)
    d = {}
    for nm, kind, cnt in spec:
        if kind == "f":
            pick = rng.random()
            vals = [rng.randrange(P) if pick < 0.3 else rng.randrange(1 << rng.choice((1, 8, 16, 24, 32, 64, 200))) for _ in range(cnt)]
        else:
            hi = rng.choice((2, 16, 16, 256, 256, 300))
            vals = [rng.randrange(hi) for _ in range(cnt)]
        d[nm] = vals if cnt != 1 else vals[0]            # (both loaders flatten: a 1-element array and a scalar are the same input)
    # lengths / selectors near their arrays' sizes so that the accepting paths are exercised too
    lens = {"mainLen": "mainInput", "count": "in", "aLen": "a", "bLen": "b", "select": None, "inLen": "in", "layerLen": "layer",
            "addressHashNibblesLen": "addressHashNibbles"}
    for ln, arr in lens.items():
        if ln in d and rng.random() < 0.8:
            n = len(d[arr]) if arr and isinstance(d.get(arr), list) else (params[0] if params else 4)
            d[ln] = rng.randrange(0, n + 2)
    return d


# the templates at the sizes the production circuit instantiates them with (and the largest the gadget-main path takes), beyond the reference's small cases
LARGE_MAINS = ["SubstringCheck(136, 31)", "RlpMerklePatriciaTrieLeaf(32, 31)", "Selector(16)", "SelectorArray1D(16, 136)", "SelectorArray2D(5, 4, 16)",
               "Concat(36, 103)", "ShiftLeft(64)", "ShiftRight(103, 36)", "Pad(4, 136)", "TruncatedAddressHash(32)", "Num2BigEndianBytes(31)", "Num2LittleEndianBytes(32)",
               "Poseidon(4)", "KeccakBytes(4)", "PublicCommitment(4)", "Divide(30)", "IsInRange(30)", "Mask(139)", "CountBytes(31)", "Fit(32, 31)", "Fit(104, 136)",
               "AssertBits(253)", "AssertByteString(136)", "Num2BitsSafe(255)", "Num2BitsSafe(253)", "LittleEndianBytes2Num(31)", "BigEndianBytes2Num(31)",
               "Bytes2Nibbles(32)", "Nibbles2Bytes(33)", "RlpInteger(31)", "RlpEmptyAccount(31)", "LeafDetector(136)", "Filter(64)", "Reverse(31)", "Flatten(6, 32)",
               "AssertLessEqThan(30)", "AssertLessThan(16)", "AssertGreaterEqThan(16)", "ConcatFixed4(32, 32, 32, 8)",
               "Mask(300)", "Concat(300, 5)"]        # (beyond 256 entries: Mask's filter[] runs carry 4 x 64 bits, larger ones take the per-wire loop -- advisor, round 4)


def differential(pkg, s, n: int = 48, seed: int = 11) -> list:
    """n seeded random inputs of the suite's main against the oracle: same accept / reject decision, same outputs, same payload"""
    from proof_of_burn_amd.witness import parse_main
    main = s["main"]
    name, params = parse_main(main)
    rng = random.Random(f"{main}/{seed}")
    inputs = [_rand_inputs(name, params, rng) for _ in range(n)]
    bad = []
    calc = pkg.WitnessCalculator(main, max_batch=n)
    try:
        res = calc.calculate(inputs, check=True)
        nok = 0
        for i, (inp, r) in enumerate(zip(inputs, res)):
            ora = O.run(main, inp)
            if r.ok != (not ora.failed):
                bad.append((main, i, "accept/reject", r.ok, not ora.failed, hex(r.status), ora.msg, inp))
                continue
            if not r.ok:
                continue
            nok += 1
            if r.outputs != ora.outputs():
                bad.append((main, i, "outputs", r.outputs, ora.outputs()))
            if r.check_status != 0 or r.bad_wire is not None:
                bad.append((main, i, "evaluator", r.check_status, r.bad_wire))
            if not np.array_equal(calc.witness_payload(i), ora.witness_numpy()):
                bad.append((main, i, "payload"))
    finally:
        calc.close()
    return bad, nok
