"""`snarkjs wtns check` equivalent: evaluate every rank-1 constraint A*B = C of the circuit model on a canonical witness
(a .wtns file or its payload) -- independently of the code that produced the witness.

Almost every row touches only bits / bytes / lengths with tiny coefficients: those are evaluated exactly in int64 with numpy
(|A|,|B| < 2^31 is verified per row, then A*B - C over the integers is below p in magnitude, so "!= 0" is exact); rows with a
field-sized wire or coefficient (Poseidon, SubstringCheck, the 254-bit decompositions, IsZero inverses) go through Python integers.
"""
from __future__ import annotations

import numpy as np

from .core import COEF, Circuit, Flat, P

SMALL = 1 << 31


class Witness:
    def __init__(self, payload: np.ndarray):
        """payload: uint8[32 * W] canonical little-endian values"""
        self.limbs = np.ascontiguousarray(payload).view("<u8").reshape(-1, 4)
        self.n = self.limbs.shape[0]
        hi = (self.limbs[:, 1] | self.limbs[:, 2] | self.limbs[:, 3]) != 0
        lo = self.limbs[:, 0]
        self.small = np.where(hi | (lo >= SMALL), -1, lo.astype(np.int64)).astype(np.int64)      # -1: not a small value

    @staticmethod
    def from_wtns(path: str) -> "Witness":
        data = np.fromfile(path, dtype=np.uint8)
        assert bytes(data[:4]) == b"wtns", "not a .wtns file"
        return Witness(data[76:])

    def value(self, w: int) -> int:
        a = self.limbs[w]
        return int(a[0]) | (int(a[1]) << 64) | (int(a[2]) << 128) | (int(a[3]) << 192)


_coef_small_cache = [np.zeros(0, dtype=np.int64), np.zeros(0, dtype=bool)]


def _coef_tables():
    """int64 value of every interned coefficient as a signed small number (c or c - p), and which are not small"""
    n = len(COEF)
    if len(_coef_small_cache[0]) != n:
        vals = np.zeros(n, dtype=np.int64); big = np.zeros(n, dtype=bool)
        for i, c in enumerate(COEF):
            s = c if c < P // 2 else c - P
            if -SMALL < s < SMALL:
                vals[i] = s
            else:
                big[i] = True
        _coef_small_cache[0], _coef_small_cache[1] = vals, big
    return _coef_small_cache


def _row_sums(ptr, vals):
    """per-row sums of a CSR segment list (rows may be empty)"""
    out = np.zeros(len(ptr) - 1, dtype=np.int64)
    nz = ptr[1:] > ptr[:-1]
    if vals.size:
        out[nz] = np.add.reduceat(vals, ptr[:-1][nz])
    return out


def eval_piece(fl: Flat, base: int, wit: Witness):
    """-> indices (within the piece) of the violated rows"""
    cvals, cbig = _coef_tables()
    slow = np.zeros(fl.n, dtype=bool)
    sums = []
    for s in range(3):
        idx = fl.idx[s]
        absw = np.where(idx < 0, 0, idx + base)
        v = wit.small[absw]
        bad_term = (v < 0) | cbig[fl.cid[s]]
        if bad_term.any():
            rows = np.searchsorted(fl.ptr[s], np.nonzero(bad_term)[0], side="right") - 1
            slow[rows] = True
        term = np.where(bad_term, 0, v * cvals[fl.cid[s]])
        sums.append(_row_sums(fl.ptr[s], term))
    a, b, c = sums
    slow |= (np.abs(a) >= SMALL) | (np.abs(b) >= SMALL) | (np.abs(c) >= (1 << 62))
    bad = (~slow) & (a * b != c)
    out = list(np.nonzero(bad)[0])
    for r in np.nonzero(slow)[0]:
        acc = []
        for s in range(3):
            t = 0
            for j in range(int(fl.ptr[s][r]), int(fl.ptr[s][r + 1])):
                i = int(fl.idx[s][j])
                t += COEF[int(fl.cid[s][j])] * (1 if i < 0 else wit.value(i + base))
            acc.append(t % P)
        if (acc[0] * acc[1] - acc[2]) % P:
            out.append(int(r))
    return sorted(out)


def row_wires(fl: Flat, base: int, r: int):
    """absolute wire indices a row touches (for reports)"""
    ws = set()
    for s in range(3):
        for j in range(int(fl.ptr[s][r]), int(fl.ptr[s][r + 1])):
            i = int(fl.idx[s][j])
            ws.add(0 if i < 0 else i + base)
    return sorted(ws)


def check_witness(circ: Circuit, wit: Witness, max_report: int = 16):
    """-> list of (global row number, wires of the row) of violated constraints (empty = the witness satisfies the circuit)"""
    assert wit.n == circ.n_wires, f"witness has {wit.n} wires, circuit {circ.n_wires}"
    assert wit.value(0) == 1, "wire 0 must be the constant 1"
    bad, row0 = [], 0
    for fl, base in circ.pieces():
        for r in eval_piece(fl, base, wit):
            if len(bad) < max_report:
                bad.append((row0 + r, row_wires(fl, base, r)))
        row0 += fl.n
    assert row0 == circ.n_constraints
    return bad
