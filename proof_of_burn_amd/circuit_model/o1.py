"""O1-style reduced witness: circom's --O1 removes the signals that a LINEAR constraint pins to another signal
(`a === b`, as every `<==` copy through component boundaries is at --O0) or to a constant, and keeps one representative.  The map is
derived from the model's own constraints; which representative circom keeps, and its renumbering, are NOT pinned (no circom here):
this map keeps the lowest wire of each class (so main's outputs and inputs survive) and numbers the survivors in wire order.

    m = reduce_map(circuit)            # m.keep: surviving O0 wire indices; m.alias[w]: representative of a dropped wire (or -1: constant)
    reduced_payload = payload.reshape(-1, 32)[m.keep]
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from .core import COEF, Circuit, P


@dataclass
class O1Map:
    keep: np.ndarray          # int64, sorted: O0 wire indices of the reduced witness (wire 0 first)
    alias: np.ndarray         # int64[W]: for every O0 wire its representative (itself if kept), -1 for a wire pinned to a constant
    const_wires: np.ndarray   # the wires pinned to constants
    const_values: list        # their values
    n_linear_rows: int
    n_rows: int

    def reduce(self, payload: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(payload.reshape(-1, 32)[self.keep]).reshape(-1)


def reduce_map(circ: Circuit) -> O1Map:
    W = circ.n_wires
    xs, ys, cw, cv = [], [], [], []
    n_lin = 0
    for fl, base in circ.pieces():
        na = fl.ptr[0][1:] - fl.ptr[0][:-1]; nb = fl.ptr[1][1:] - fl.ptr[1][:-1]; nc = fl.ptr[2][1:] - fl.ptr[2][:-1]
        lin = (na == 0) | (nb == 0)                          # A = 0 or B = 0: the row says C = 0
        n_lin += int(lin.sum())
        start = fl.ptr[2][:-1]
        idx, cid = fl.idx[2], fl.cid[2]
        # signal === signal: two wire terms with opposite coefficients, no constant
        r2 = np.nonzero(lin & (nc == 2))[0]
        if r2.size:
            i0, i1 = idx[start[r2]], idx[start[r2] + 1]
            c0 = np.array([COEF[int(c)] for c in np.unique(cid)], dtype=object)
            lut = {int(c): COEF[int(c)] for c in np.unique(cid)}
            opp = np.fromiter(((lut[int(a)] + lut[int(b)]) % P == 0 for a, b in zip(cid[start[r2]], cid[start[r2] + 1])), dtype=bool, count=r2.size)
            ok = opp & (i0 >= 0) & (i1 >= 0)
            xs.append(i0[ok] + base); ys.append(i1[ok] + base)
            # signal === constant written as  c*w + k*ONE = 0
            isc = (~ok) & ((i0 < 0) ^ (i1 < 0))
            for r in r2[isc]:
                a, b = int(start[r]), int(start[r]) + 1
                if idx[a] < 0:
                    a, b = b, a
                cw.append(int(idx[a]) + base); cv.append((-COEF[int(cid[b])] * pow(COEF[int(cid[a])], -1, P)) % P)
            del c0
        # signal === 0
        r1 = np.nonzero(lin & (nc == 1))[0]
        for r in r1:
            a = int(start[r])
            if idx[a] >= 0:
                cw.append(int(idx[a]) + base); cv.append(0)
    x = np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)
    y = np.concatenate(ys) if ys else np.zeros(0, dtype=np.int64)
    g = coo_matrix((np.ones(x.size, dtype=np.int8), (x, y)), shape=(W, W))
    _, lab = connected_components(g, directed=False)
    _, first = np.unique(lab, return_index=True)                # lowest wire of every class (labels are scanned in wire order)
    alias = first[lab].astype(np.int64)
    # a class that contains a constant-pinned wire is constant as a whole
    const_wires = np.array(sorted(set(cw)), dtype=np.int64)
    val = dict(zip(cw, cv))
    const_class = np.zeros(W, dtype=bool)
    const_class[alias[const_wires]] = True
    is_const = const_class[alias]
    is_const[0] = False
    keep = np.nonzero((alias == np.arange(W)) & ~is_const)[0]
    cwires = np.nonzero(is_const)[0]
    class_val = {int(alias[w]): v for w, v in val.items()}
    alias_out = alias.copy()
    alias_out[is_const] = -1
    return O1Map(keep.astype(np.int64), alias_out, cwires, [class_val[int(alias[w])] for w in cwires], n_lin, circ.n_constraints)
