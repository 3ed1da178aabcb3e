"""Host-side Poseidon (x^5, BN254 Fr, R_F = 8, circomlib's parameter set) for the INPUT PRODUCER: burn address, nullifier and
remaining coin of synthetic / real proofs (reference tests/main.py:29-45 calls tests/poseidon.py for the same values).

The round constants and the MDS matrix are regenerated from the public specification with the Poseidon paper's Grain-LFSR
procedure (Grassi et al., USENIX Sec'21, app. F):
  * 80-bit state = field(2b)=1 | sbox(4b)=0 | n(12b)=254 | t(12b) | R_F(10b) | R_P(10b) | 30 ones, taps 62,51,38,23,13,0;
    160 warm-up clocks; bits are consumed in pairs (b1,b2): emit b2 iff b1=1;
  * round constants: 254-bit samples, rejected when >= p;  MDS: the next 2t samples (mod p) are xs|ys, M[i][j] = 1/(xs[i]+ys[j]).
tools/gen_poseidon.py (which derives circomlib's optimised schedule for the device tables) imports these definitions and
checks them against the reference's tests/poseidon.py tables in the build container."""
from __future__ import annotations

import functools

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R_F = 8
R_P_TABLE = {2: 56, 3: 57, 4: 56, 5: 60}  # indexed by t = nInputs + 1


def inv(x: int) -> int:
    return pow(x % P, -1, P)


def _grain_stream(t: int, rf: int, rp: int, n: int = 254):
    bits = [int(b) for b in (
        format(1, "02b") + format(0, "04b") + format(n, "012b") + format(t, "012b")
        + format(rf, "010b") + format(rp, "010b"))] + [1] * 30
    assert len(bits) == 80

    def clock() -> int:
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb

    for _ in range(160):
        clock()

    def next_bit() -> int:
        b = clock()
        while b == 0:
            clock()
            b = clock()
        return clock()

    while True:
        x = 0
        for _ in range(n):
            x = (x << 1) | next_bit()
        yield x


@functools.lru_cache(maxsize=None)
def plain_constants(t: int):
    rp = R_P_TABLE[t]
    g = _grain_stream(t, R_F, rp)
    c = []
    while len(c) < (R_F + rp) * t:
        x = next(g)
        if x < P:
            c.append(x)
    rl = [next(g) % P for _ in range(2 * t)]
    xs, ys = rl[:t], rl[t:]
    m = [[inv(xs[i] + ys[j]) for j in range(t)] for i in range(t)]
    return c, m


def mat_vec(a, v):
    return [sum(a[i][j] * v[j] for j in range(len(v))) % P for i in range(len(a))]


def pow5(x):
    x2 = x * x % P
    return x2 * x2 % P * x % P


def poseidon(inputs) -> int:
    """the plain permutation: capacity element 0 first, out = state[0]"""
    t = len(inputs) + 1
    rp = R_P_TABLE[t]
    c, A = plain_constants(t)
    s = [0] + [x % P for x in inputs]
    for r in range(R_F + rp):
        s = [(s[i] + c[r * t + i]) % P for i in range(t)]
        if r < R_F // 2 or r >= R_F // 2 + rp:
            s = [pow5(x) for x in s]
        else:
            s[0] = pow5(s[0])
        s = mat_vec(A, s)
    return s[0]
