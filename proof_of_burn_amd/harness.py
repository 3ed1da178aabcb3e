"""`run(main, test_cases)` with the signature and semantics of the reference harness (tests/test.py:6-75),
executed on the HIP calculator instead of `circom -c` + the emitted binary.  Every entry of the reference's list
(tests/test.py:146-201) runs: the two circuits with a `component main` (ProofOfBurn(...), Spend(...)) and the 54
gadget-level mains the harness wraps around one template of circuits/utils (csrc/gadget_mains.hpp).
"""
import sys

from .witness import WitnessCalculator


def run(main, test_cases):
    print()
    print(f"Testing {main}")
    print("=" * 20)
    calc = WitnessCalculator(main, max_batch=max(1, len(test_cases)))
    try:
        # malformed inputs (missing keys / wrong shapes) abort the reference binary => None
        good, results = [], [None] * len(test_cases)
        for i, (case, _) in enumerate(test_cases):
            try:
                calc.pack([case])
                good.append(i)
            except (KeyError, ValueError, TypeError):
                pass
            except NotImplementedError as e:
                # a byte / length / selector signal given a value outside int32: the reference reduces it mod p and runs (tests/test.py:65-73: None
                # or outputs).  The gadget path keeps such signals as int32 and cannot represent it: reported as None -- LOUDLY, and if the
                # reference's list expects outputs for this case the comparison below raises
                print(f"warning: {main} case {i}: {e}; reported as None", file=sys.stderr)
        if good:
            res = calc.calculate([test_cases[i][0] for i in good])
            for i, r in zip(good, res):
                results[i] = r.outputs if r.ok else None
        for (case, expected), got in zip(test_cases, results):
            if got is None:
                if expected is not None:
                    raise Exception("Expected null!")
            elif got != expected:
                raise Exception(f"Unexpected output! {got} != {expected}")
        return results
    finally:
        calc.close()
