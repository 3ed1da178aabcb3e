#!/usr/bin/env python3
"""Derive the Poseidon (x^5, BN254 Fr, R_F = 8) parameters from the public specification and
emit them as C headers for the CPU oracle and the HIP product.

Nothing here is copied from the reference: the round constants and the MDS matrix are
regenerated with the Poseidon paper's Grain-LFSR procedure (Grassi et al., "Poseidon",
USENIX Sec'21, app. F / the authors' generate_parameters_grain script):

  * 80-bit state = field(2b)=1 | sbox(4b)=0 | n(12b)=254 | t(12b) | R_F(10b) | R_P(10b) | 30 ones,
    taps 62,51,38,23,13,0; 160 warm-up clocks; bits are consumed in pairs (b1,b2): emit b2 iff b1=1;
  * round constants: 254-bit samples, rejected when >= p;
  * MDS: the next 2t samples (reduced mod p, no rejection) are xs|ys, M[i][j] = 1/(xs[i]+ys[j]).

`--check-reference` (only usable in the build container, where /root/reference exists) verifies
that this reproduces tests/poseidon.py:POSEIDON_C / POSEIDON_M of the reference exactly.

circomlib's `poseidon.circom` (UNVENDORED dependency of the reference) does
not evaluate the plain round function: it uses the "optimised" schedule
    ark(C[0:t]) | 3x [sbox_full, +C, Mix(M)] | sbox_full, +C, Mix(P) |
    R_P x [sbox(s0), s0 += C, MixS(S_r)] | 3x [sbox_full, +C, Mix(M)] | sbox_full, MixLast
with a folded constant vector C (len 8t+R_P), a pre-matrix P and sparse matrices
S_r = [[M00, v_r], [w_r, I]].  Requiring (a) exactly that structure and (b) equality with the plain
permutation for every input determines C, P, S uniquely (derivation in `optimized()` below), so the
intermediate wires are pinned by the structure even though the circomlib constant file is absent.
`self_check()` proves (b) on random inputs.
"""
from __future__ import annotations

import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the specification part (Grain LFSR constants, MDS, plain permutation) lives in the package: the input producer needs it at run time
from proof_of_burn_amd.poseidon_host import P, R_F, R_P_TABLE, inv, _grain_stream, plain_constants, mat_vec, pow5  # noqa: E402,F401
from proof_of_burn_amd.poseidon_host import poseidon as poseidon_plain  # noqa: E402


from proof_of_burn_amd.poseidon_host import mat_mul, mat_inv, identity, optimized  # noqa: E402,F401


# ------------------------------------------------------------------ reference evaluations

def poseidon_opt_trace(inputs, params=None):
    """Evaluate circomlib's schedule; returns (hash, list of per-stage states) for wire checks."""
    t = len(inputs) + 1
    rp = R_P_TABLE[t]
    C, S, A, Pm = params or optimized(t)
    trace = []
    s = [(x + C[i]) % P for i, x in enumerate([0] + [v % P for v in inputs])]
    trace.append(("ark0", list(s)))
    for r in range(3):
        s = [pow5(x) for x in s]
        s = [(s[i] + C[(r + 1) * t + i]) % P for i in range(t)]
        s = mat_vec(A, s)
        trace.append((f"mix{r}", list(s)))
    s = [pow5(x) for x in s]
    s = [(s[i] + C[4 * t + i]) % P for i in range(t)]
    s = mat_vec(Pm, s)
    trace.append(("mixP", list(s)))
    for r in range(rp):
        s0 = (pow5(s[0]) + C[5 * t + r]) % P
        base = (2 * t - 1) * r
        n0 = (S[base] * s0 + sum(S[base + i] * s[i] for i in range(1, t))) % P
        s = [n0] + [(s[i] + s0 * S[base + t + i - 1]) % P for i in range(1, t)]
    trace.append(("partial_end", list(s)))
    for r in range(3):
        s = [pow5(x) for x in s]
        s = [(s[i] + C[5 * t + rp + r * t + i]) % P for i in range(t)]
        s = mat_vec(A, s)
    s = [pow5(x) for x in s]
    out = sum(A[0][j] * s[j] for j in range(t)) % P
    return out, trace


def self_check(seed: int = 7503) -> None:
    rng = random.Random(seed)
    for t in (2, 3, 4, 5):
        prm = optimized(t)
        for _ in range(4):
            inp = [rng.randrange(P) for _ in range(t - 1)]
            assert poseidon_plain(inp) == poseidon_opt_trace(inp, prm)[0], t
    # published circomlib/circomlibjs test vectors for the plain function
    assert poseidon_plain([1, 2]) == 7853200120776062878684798364095072458815029376092732009249414926327459813530


def check_reference(path: str = "/root/reference") -> None:
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_poseidon", os.path.join(path, "tests", "poseidon.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for n_in in (1, 2, 3, 4):
        c, m = plain_constants(n_in + 1)
        assert c == [x.val for x in ref.POSEIDON_C[n_in - 1]], n_in
        assert m == [[x.val for x in row] for row in ref.POSEIDON_M[n_in - 1]], n_in
        assert ref.ROUNDS_P[n_in - 1] == R_P_TABLE[n_in + 1]
    rng = random.Random(1)
    for n_in in (2, 3, 4):
        inp = [rng.randrange(P) for _ in range(n_in)]
        assert ref.poseidon([ref.Field(x) for x in inp]).val == poseidon_plain(inp)
    print("reference tables reproduced exactly (C, M for t=2..5; hashes for 2/3/4 inputs)")


# ------------------------------------------------------------------ emitters

def _limbs64(x: int):
    return [(x >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]


def _limbs32(x: int):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def emit_oracle_header(path: str) -> None:
    """Canonical (non-Montgomery) 4x64 little-endian limbs for the C oracle."""
    out = ["/* GENERATED by tools/gen_poseidon.py -- do not edit.  Poseidon x^5 / BN254 Fr / R_F=8",
           " * parameters in circomlib's optimised schedule (C, S, M, P), canonical 4x64 LE limbs. */",
           "#ifndef ORACLE_POSEIDON_CONSTS_H", "#define ORACLE_POSEIDON_CONSTS_H", "#include <stdint.h>", ""]
    for t in (2, 3, 4, 5):
        C, S, A, Pm = optimized(t)
        rp = R_P_TABLE[t]

        def arr(name, vals):
            out.append(f"static const uint64_t {name}[{len(vals)}][4] = {{")
            for v in vals:
                out.append("  {" + ",".join(f"0x{w:016x}ULL" for w in _limbs64(v)) + "},")
            out.append("};")
        out.append(f"#define POS_RP_{t} {rp}")
        arr(f"POS_C_{t}", C)
        arr(f"POS_S_{t}", S)
        arr(f"POS_M_{t}", [A[i][j] for i in range(t) for j in range(t)])   # row-major A[i][j], new[i]=sum_j A[i][j] old[j]
        arr(f"POS_P_{t}", [Pm[i][j] for i in range(t) for j in range(t)])
        out.append("")
    out.append("#endif")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def emit_device_header(path: str) -> None:
    """Montgomery form (x * 2^256 mod p), 8x32 LE limbs, one flat table + offsets, for the HIP side."""
    R = (1 << 256) % P
    out = ["// GENERATED by tools/gen_poseidon.py -- do not edit.",
           "// Poseidon x^5 / BN254 Fr / R_F = 8, circomlib's optimised schedule (C | S | M | P per t),",
           "// Montgomery form (x*2^256 mod p) as 8x32-bit little-endian limbs.",
           "#pragma once", "#include <stdint.h>", ""]
    flat = []
    offs = {}
    for t in (2, 3, 4, 5):
        C, S, A, Pm = optimized(t)
        offs[t] = {}
        for name, vals in (("C", C), ("S", S), ("M", [A[i][j] for i in range(t) for j in range(t)]),
                           ("P", [Pm[i][j] for i in range(t) for j in range(t)])):
            offs[t][name] = len(flat)
            flat.extend(vals)
    out.append(f"#define POS_TABLE_LEN {len(flat)}")
    for t in (2, 3, 4, 5):
        out.append(f"#define POS_RP_{t} {R_P_TABLE[t]}")
        for name in "CSMP":
            out.append(f"#define POS_OFF_{name}_{t} {offs[t][name]}")
    out.append("static const uint32_t POS_TABLE_MONT[POS_TABLE_LEN][8] = {")
    for v in flat:
        out.append("  {" + ",".join(f"0x{w:08x}u" for w in _limbs32(v * R % P)) + "},")
    out.append("};")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-reference", action="store_true")
    ap.add_argument("--emit", action="store_true", help="write oracle/ and csrc/ headers")
    args = ap.parse_args()
    self_check()
    print("optimised schedule == plain permutation on random inputs (t=2..5)")
    if args.check_reference:
        check_reference()
    if args.emit:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        emit_oracle_header(os.path.join(root, "oracle", "poseidon_consts.h"))
        emit_device_header(os.path.join(root, "proof_of_burn_amd", "csrc", "poseidon_consts.h"))
        print("headers written")


if __name__ == "__main__":
    main()
