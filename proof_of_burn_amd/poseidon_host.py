"""Host-side Poseidon (x^5, BN254 Fr, R_F = 8, circomlib's parameter set) for the INPUT PRODUCER: burn address, nullifier and
remaining coin of synthetic / real proofs (reference tests/main.py:29-45 calls tests/poseidon.py for the same values).

The round constants and the MDS matrix are regenerated from the public specification with the Poseidon paper's Grain-LFSR
procedure (Grassi et al., USENIX Sec'21, app. F):
  * 80-bit state = field(2b)=1 | sbox(4b)=0 | n(12b)=254 | t(12b) | R_F(10b) | R_P(10b) | 30 ones, taps 62,51,38,23,13,0;
    160 warm-up clocks; bits are consumed in pairs (b1,b2): emit b2 iff b1=1;
  * round constants: 254-bit samples, rejected when >= p;  MDS: the next 2t samples (mod p) are xs|ys, M[i][j] = 1/(xs[i]+ys[j]).
`optimized(t)` derives the constants of circomlib's OPTIMISED schedule (poseidon.circom: folded constants C, sparse matrices S, pre-matrix
P) from the plain ones -- the device tables (tools/gen_poseidon.py emits them) and the circuit model's Poseidon constraints use them.
tools/gen_poseidon.py checks all of this against the reference's tests/poseidon.py tables in the build container."""
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


def mat_mul(a, b):
    n, k, m = len(a), len(b), len(b[0])
    return [[sum(a[i][x] * b[x][j] for x in range(k)) % P for j in range(m)] for i in range(n)]


def mat_inv(a):
    n = len(a)
    aug = [list(r) + [1 if i == j else 0 for j in range(n)] for i, r in enumerate(a)]
    for col in range(n):
        piv = next(r for r in range(col, n) if aug[r][col] % P)
        aug[col], aug[piv] = aug[piv], aug[col]
        iv = inv(aug[col][col])
        aug[col] = [x * iv % P for x in aug[col]]
        for r in range(n):
            if r != col and aug[r][col]:
                f = aug[r][col]
                aug[r] = [(x - f * y) % P for x, y in zip(aug[r], aug[col])]
    return [r[n:] for r in aug]


def identity(n):
    return [[1 if i == j else 0 for j in range(n)] for i in range(n)]


# ------------------------------------------------------------------ optimised schedule

@functools.lru_cache(maxsize=None)
def optimized(t: int):
    """Return (C_opt, S, A, Pm) for circomlib's schedule.  Column convention: new = A @ old.

    Let u be the state after the 4th full S-box layer.  Plain: w0 = A u + c4; per partial round r:
    w <- A*sbox0(w) + c_{5+r}.  Optimised: y0 = Pm (u + k); y <- S_r (sbox0(y) + kappa_r e0).
    Ansatz w_r = T_r y_r + d_r with T_r = diag(1, B_r), d_r[0] = 0, T_RP = I, d_RP = 0.  Matching
    terms gives  S_r = T_{r+1}^-1 A T_r  (bottom-right block must be I  =>  B_r = Ahat^-1 B_{r+1}),
    kappa_r*A e0 + d_{r+1} = A d_r + c_{5+r},  Pm = T_0^-1 A,  k = A^-1 (c4 - d_0).
    """
    rp = R_P_TABLE[t]
    c, A = plain_constants(t)
    cr = [c[i * t:(i + 1) * t] for i in range(R_F + rp)]
    Ainv = mat_inv(A)
    Ahat = [row[1:] for row in A[1:]]
    Ahat_inv = mat_inv(Ahat)
    a_row = A[0][1:]
    a_col = [A[i][0] for i in range(1, t)]

    # B_r for r = RP .. 0
    B = [None] * (rp + 1)
    B[rp] = identity(t - 1)
    for r in range(rp - 1, -1, -1):
        B[r] = mat_mul(Ahat_inv, B[r + 1])

    S = []
    for r in range(rp):
        v = [sum(a_row[x] * B[r][x][j] for x in range(t - 1)) % P for j in range(t - 1)]
        w_hat = mat_vec(mat_inv(B[r + 1]), a_col)
        S.extend([A[0][0]] + v + w_hat)  # (2t-1) entries per round, circomlib's S layout

    # constants, backwards
    kappa = [0] * rp
    d = [None] * (rp + 1)
    d[rp] = [0] * t
    for r in range(rp - 1, -1, -1):
        # A (d_r - kappa_r e0) = d_{r+1} - c_{5+r}
        rhs = [(d[r + 1][i] - cr[5 + r][i]) % P for i in range(t)]
        x = mat_vec(Ainv, rhs)
        kappa[r] = (-x[0]) % P
        d[r] = [0] + x[1:]
    T0_inv = [[1] + [0] * (t - 1)] + [[0] + row for row in mat_inv(B[0])]
    Pm = mat_mul(T0_inv, A)
    k = mat_vec(Ainv, [(cr[4][i] - d[0][i]) % P for i in range(t)])

    C = list(cr[0])
    for r in (1, 2, 3):
        C.extend(mat_vec(Ainv, cr[r]))
    C.extend(k)
    C.extend(kappa)
    for r in range(3):
        C.extend(mat_vec(Ainv, cr[4 + rp + 1 + r]))
    assert len(C) == R_F * t + rp and len(S) == rp * (2 * t - 1)
    return C, S, A, Pm


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
