"""Drop-in for the emitted calculator binaries (reference Makefile:4-5):

    python -m proof_of_burn_amd.calc proof_of_burn input.json witness.wtns
    python -m proof_of_burn_amd.calc spend         input.json witness.wtns
    python -m proof_of_burn_amd.calc "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)" input.json witness.wtns

Like the reference binary: on a failed assert a message goes to stderr (the harness' failure signal,
tests/test.py:65-68) and no witness is written.  Exit code 0/1.
"""
import sys

from .witness import calculate_witness

MAINS = {
    "proof_of_burn": "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",   # circuits/main_proof_of_burn.circom:27
    "main_proof_of_burn": "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",
    "spend": "Spend(31)",                                                        # circuits/main_spend.circom:6
    "main_spend": "Spend(31)",
}


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3:
        print("usage: python -m proof_of_burn_amd.calc <proof_of_burn|spend|'Template(args)'> <input.json> <witness.wtns>", file=sys.stderr)
        return 2
    main_str = MAINS.get(argv[0], argv[0])
    try:
        r = calculate_witness(main_str, argv[1], argv[2])
    except (KeyError, ValueError) as e:          # the emitted loader: "Not all inputs have been set"
        print(f"input error: {e}", file=sys.stderr)
        return 1
    if not r.ok:
        print(r.message(), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
