#!/usr/bin/env python3
"""Regenerate tests/golden/suites.json from the reference's own known-answer tests.

Run in the BUILD CONTAINER ONLY (needs /root/reference; the GPU box has no such path):

    python3 tests/golden/make_golden.py

It imports /root/reference/tests/testcases/*.py *unmodified* (with tests/refshim.py standing in
for the absent web3 / rlp / eth_abi packages) and dumps, in the order of the 56 `run(*test_...)`
calls of /root/reference/tests/test.py:146-201, every

    (template-instantiation string, [(input dict, expected outputs | None)])

tuple.  `None` = "the calculator must fail" (tests/test.py:65-68).  The two JSON fixtures
(tests/test_pob_input.json, tests/test_spend_input.json) are embedded as the inputs of the
ProofOfBurn / Spend suites.  Nothing here is hand-written expectation: every expected value is
what the reference's test code computes.
"""
import importlib
import importlib.util
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main() -> None:
    spec = importlib.util.spec_from_file_location("refshim", os.path.join(HERE, "..", "refshim.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    shim.install_shims()

    os.chdir(REF)  # proof_of_burn.py / spend.py open "tests/..." relative to cwd
    sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
    for k in [k for k in sys.modules if k == "tests" or k.startswith("tests.")]:
        del sys.modules[k]

    src = open(os.path.join(REF, "tests", "test.py")).read()
    name_to_mod = {}
    for m in re.finditer(r"from \.testcases\.([\w\.]+) import\s*(\(([^)]*)\)|([^\n]+))", src):
        names = m.group(3) if m.group(3) is not None else m.group(4)
        for n in re.findall(r"\w+", names):
            name_to_mod[n] = m.group(1)
    order = re.findall(r"^run\(\*(\w+)\)", src, flags=re.M)
    assert len(order) == 56, len(order)

    suites = []
    for n in order:
        mod = importlib.import_module("tests.testcases." + name_to_mod[n])
        main_str, cases = getattr(mod, n)
        suites.append({
            "name": n,
            "main": main_str,
            "cases": [{"input": inp, "expected": exp} for inp, exp in cases],
        })

    # sanity: the one Keccak-256 value the reference hard-codes (proof_of_burn.py:22)
    pob = json.load(open(os.path.join(REF, "tests", "test_pob_input.json")))
    hdr = bytes(pob["blockHeader"][:pob["blockHeaderLen"]])
    assert shim.keccak256(hdr).hex() == "e36499b50da290131c3fa32d4f60717c8c529ae1bc3a216f32d05c05fe80368d"

    out = os.path.join(HERE, "suites.json")
    with open(out, "w") as f:
        json.dump(suites, f, separators=(",", ":"))
    ncase = sum(len(s["cases"]) for s in suites)
    nnone = sum(1 for s in suites for c in s["cases"] if c["expected"] is None)
    print(f"{len(suites)} suites, {ncase} cases ({nnone} must-fail) -> {out} ({os.path.getsize(out)} bytes)")

    for fn in ("test_pob_input.json", "test_spend_input.json"):
        with open(os.path.join(REF, "tests", fn)) as f, open(os.path.join(HERE, fn), "w") as g:
            g.write(f.read())


if __name__ == "__main__":
    main()
