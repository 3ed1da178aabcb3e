"""A symbolic model of the circuits: what the circom compiler derives from the .circom sources at --O0 -- the wire numbering, the
signal names (.sym) and the rank-1 constraint system (.r1cs) -- restated as a small Python DSL.

It is the INDEPENDENT REFEREE of the witness generators: the HIP planner (csrc/circuits.hpp) and the CPU oracle state the wire order
as executable layout code; this model states it a third time, together with every `<==` / `===` as an A*B = C row, so a witness
produced by either can be checked row by row without trusting the code that produced it (check.py), and a corrupted wire fails
at the row that defines it.

Rules restated (SURVEY.md app. B/D hypothesis, the same one the planner and the oracle follow; "parity unpinned" against a real
circom build -- the reference holds no .sym/.r1cs):
  * wire 0 is the constant 1; `main`'s block starts at wire 1;
  * a component's block = its own signals: outputs | inputs | intermediates, each in declaration order, arrays row-major --
    followed by the blocks of its sub-components in the order in which they are initialised;
  * every `<==` is a constraint (also the linear ones: A = B = 0), every `===` is a constraint, `<--` is not.

A Template subclass declares signals (`output/input/signal`), instantiates children (`comp`) in initialisation order and states
constraints over linear combinations (LC).  Instances are memoised per (class, parameters) and RELOCATABLE: all wire indices are
relative to the instance's base, so a Keccakf block is built once and used 84 times.
"""
from __future__ import annotations

import hashlib
import struct
from typing import Iterable

import numpy as np

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ONE = -1                       # relative wire index of the constant-1 wire (absolute wire 0)

# interned coefficients: constraint arrays carry int32 ids into COEF
COEF: list[int] = []
_COEF_ID: dict[int, int] = {}


def coef_id(c: int) -> int:
    c %= P
    i = _COEF_ID.get(c)
    if i is None:
        i = len(COEF)
        COEF.append(c)
        _COEF_ID[c] = i
    return i


class LC:
    """linear combination: {term: coefficient} + constant; a term is (owner, signal name, flat index) resolved when the template seals"""
    __slots__ = ("t", "k")

    def __init__(self, t=None, k=0):
        self.t = t or {}
        self.k = k % P

    @staticmethod
    def of(x) -> "LC":
        return x if isinstance(x, LC) else LC(None, int(x))

    def __add__(self, o):
        o = LC.of(o)
        t = dict(self.t)
        for key, c in o.t.items():
            v = (t.get(key, 0) + c) % P
            if v:
                t[key] = v
            else:
                t.pop(key, None)
        return LC(t, self.k + o.k)

    __radd__ = __add__

    def __neg__(self):
        return LC({key: (-c) % P for key, c in self.t.items()}, -self.k)

    def __sub__(self, o):
        return self + (-LC.of(o))

    def __rsub__(self, o):
        return LC.of(o) + (-self)

    def __mul__(self, c):
        if isinstance(c, LC):
            if not c.t:
                c = c.k
            elif not self.t:
                return c * self.k
            else:
                raise TypeError("LC * LC is quadratic: state it with Template.mul / constrain")
        c = int(c) % P
        if c == 0:
            return LC()
        return LC({key: v * c % P for key, v in self.t.items()}, self.k * c)

    __rmul__ = __mul__


class Sig:
    """a declared signal (array): indexing yields the LC of one wire"""

    def __init__(self, owner, name: str, shape: tuple, kind: str):
        self.owner, self.name, self.shape, self.kind = owner, name, tuple(shape), kind
        self.size = int(np.prod(shape)) if shape else 1
        self.offset = None           # relative to the owner's block; set when the owner seals

    def flat(self, idx) -> int:
        if not self.shape:
            assert idx in ((), None, 0)
            return 0
        if not isinstance(idx, tuple):
            idx = (idx,)
        assert len(idx) == len(self.shape), (self.name, idx, self.shape)
        f = 0
        for i, n in zip(idx, self.shape):
            assert 0 <= i < n, (self.name, idx, self.shape)
            f = f * n + i
        return f

    def __getitem__(self, idx) -> LC:
        return LC({(self.owner, self.name, self.flat(idx)): 1})

    def w(self, f: int) -> LC:
        """wire by flat (row-major) index"""
        assert 0 <= f < self.size
        return LC({(self.owner, self.name, f): 1})

    def all(self):
        return [self.w(f) for f in range(self.size)]

    @property
    def lc(self) -> LC:
        assert not self.shape
        return LC({(self.owner, self.name, 0): 1})


class Child:
    """a sub-component instance inside a template: access to its signals as LCs of the parent's block"""

    def __init__(self, parent, name: str, tpl: "Template"):
        self.parent, self.name, self.tpl = parent, name, tpl
        self.offset = None

    def __getitem__(self, signame: str) -> "ChildSig":
        return ChildSig(self, self.tpl.sigs[signame])


class ChildSig:
    def __init__(self, child: Child, sig: Sig):
        self.child, self.sig = child, sig
        self.shape, self.size = sig.shape, sig.size

    def __getitem__(self, idx) -> LC:
        return LC({(self.child, self.sig.name, self.sig.flat(idx)): 1})

    def w(self, f: int) -> LC:
        return LC({(self.child, self.sig.name, f): 1})

    def all(self):
        return [self.w(f) for f in range(self.size)]

    @property
    def lc(self) -> LC:
        return self.w(0)


class Flat:
    """constraints as numpy CSR triplets over RELATIVE wire indices (ONE = -1)"""

    def __init__(self, ptr, idx, cid):
        self.ptr, self.idx, self.cid = ptr, idx, cid       # each: [A, B, C]
        self.n = len(ptr[0]) - 1

    @staticmethod
    def empty():
        z = lambda: np.zeros(1, dtype=np.int64)  # noqa: E731
        e = lambda d: np.zeros(0, dtype=d)       # noqa: E731
        return Flat([z(), z(), z()], [e(np.int64)] * 3, [e(np.int32)] * 3)

    @staticmethod
    def concat(parts: Iterable[tuple["Flat", int]]) -> "Flat":
        parts = [(f, off) for f, off in parts if f.n]
        if not parts:
            return Flat.empty()
        ptr, idx, cid = [], [], []
        for s in range(3):
            base, ps, ix, cs = 0, [np.zeros(1, dtype=np.int64)], [], []
            for f, off in parts:
                ps.append(f.ptr[s][1:] + base)
                base += int(f.ptr[s][-1])
                i = f.idx[s]
                ix.append(np.where(i < 0, i, i + off))
                cs.append(f.cid[s])
            ptr.append(np.concatenate(ps)); idx.append(np.concatenate(ix)); cid.append(np.concatenate(cs))
        return Flat(ptr, idx, cid)


class Template:
    """base class of a circuit template; subclasses implement build()"""
    _cache: dict = {}
    FLAT_LIMIT = 4_000_000        # templates with at most this many constraints (incl. children) are evaluated / exported in one piece

    @classmethod
    def get(cls, *params) -> "Template":
        key = (cls, params)
        t = Template._cache.get(key)
        if t is None:
            t = cls.__new__(cls)
            Template._cache[key] = t
            t._init(params)
        return t

    def _init(self, params):
        self.params = params
        self.sigs: dict[str, Sig] = {}
        self._order: list[Sig] = []
        self.children: list[Child] = []
        self._rows: list[tuple[LC, LC, LC]] = []
        self._flat_all = None
        self.build(*params)
        self._seal()

    # ---- declaration
    def _decl(self, name, shape, kind) -> Sig:
        assert name not in self.sigs, name
        s = Sig(self, name, shape, kind)
        self.sigs[name] = s
        self._order.append(s)
        return s

    def output(self, name, *shape) -> Sig:
        return self._decl(name, shape, "out")

    def input(self, name, *shape) -> Sig:
        return self._decl(name, shape, "in")

    def signal(self, name, *shape) -> Sig:
        return self._decl(name, shape, "mid")

    def comp(self, name: str, tpl: "Template") -> Child:
        c = Child(self, name, tpl)
        self.children.append(c)
        return c

    # ---- constraints
    def constrain(self, a, b, c):
        """a * b === c"""
        self._rows.append((LC.of(a), LC.of(b), LC.of(c)))

    def eq(self, lhs, rhs):
        """lhs === rhs (linear)"""
        self._rows.append((LC(), LC(), LC.of(lhs) - LC.of(rhs)))

    def assign(self, dst, expr):
        """dst <== linear expression"""
        self.eq(expr, dst)

    def mul(self, dst, a, b, lin=0):
        """dst <== a * b + lin"""
        self._rows.append((LC.of(a), LC.of(b), LC.of(dst) - LC.of(lin)))

    def copy(self, dst, src):
        """dst[i] <== src[i] over two equally sized signal (views)"""
        d, s = dst.all() if hasattr(dst, "all") else dst, src.all() if hasattr(src, "all") else src
        assert len(d) == len(s), (len(d), len(s))
        for x, y in zip(d, s):
            self.assign(x, y)

    # ---- sealing: offsets, numpy form
    def _seal(self):
        off = 0
        for kind in ("out", "in", "mid"):
            for s in self._order:
                if s.kind == kind:
                    s.offset = off
                    off += s.size
        self.n_own = off
        for c in self.children:
            c.offset = off
            off += c.tpl.n_wires
        self.n_wires = off
        self.n_rows_total = len(self._rows) + sum(c.tpl.n_rows_total for c in self.children)

        def rel(term) -> int:
            owner, name, f = term
            if owner is self:
                return self.sigs[name].offset + f
            assert isinstance(owner, Child) and owner.parent is self, "a template may only touch its own and its children's signals"
            return owner.offset + owner.tpl.sigs[name].offset + f

        ptr, idx, cid = [], [], []
        for s in range(3):
            p, ix, cs = [0], [], []
            for row in self._rows:
                lc = row[s]
                for term, c in lc.t.items():
                    ix.append(rel(term)); cs.append(coef_id(c))
                if lc.k:
                    ix.append(ONE); cs.append(coef_id(lc.k))
                p.append(len(ix))
            ptr.append(np.array(p, dtype=np.int64)); idx.append(np.array(ix, dtype=np.int64)); cid.append(np.array(cs, dtype=np.int32))
        self.own = Flat(ptr, idx, cid)
        self._rows = None

    def flat(self) -> Flat:
        """own + all descendants' constraints, own first then children in order (memoised; only for templates <= FLAT_LIMIT rows)"""
        if self._flat_all is None:
            assert self.n_rows_total <= self.FLAT_LIMIT, "too large to flatten: walk it"
            self._flat_all = Flat.concat([(self.own, 0)] + [(c.tpl.flat(), c.offset) for c in self.children])
        return self._flat_all

    def pieces(self, base: int = 0):
        """(Flat, absolute base) pieces covering every constraint of the instance at `base`, each piece <= FLAT_LIMIT rows"""
        if self.n_rows_total <= self.FLAT_LIMIT:
            if self.n_rows_total:
                yield self.flat(), base
            return
        if self.own.n:
            yield self.own, base
        for c in self.children:
            yield from c.tpl.pieces(base + c.offset)

    # ---- names (.sym)
    def names(self, prefix: str, base: int, comp_counter: list):
        """yields (wire, component index, qualified name) in wire order"""
        me = comp_counter[0]
        comp_counter[0] += 1
        for kind in ("out", "in", "mid"):
            for s in self._order:
                if s.kind != kind:
                    continue
                if not s.shape:
                    yield base + s.offset, me, f"{prefix}.{s.name}"
                else:
                    for f in range(s.size):
                        ix, r = [], f
                        for n in reversed(s.shape):
                            ix.append(r % n); r //= n
                        yield base + s.offset + f, me, f"{prefix}.{s.name}" + "".join(f"[{i}]" for i in reversed(ix))
        for c in self.children:
            yield from c.tpl.names(f"{prefix}.{c.name}", base + c.offset, comp_counter)

    def build(self, *params):
        raise NotImplementedError


class Circuit:
    """`component main = T(params)`: main's block at wire 1"""

    def __init__(self, main: Template):
        self.main = main
        self.n_wires = 1 + main.n_wires
        self.n_constraints = main.n_rows_total
        self.n_outputs = sum(s.size for s in main._order if s.kind == "out")
        self.n_inputs = sum(s.size for s in main._order if s.kind == "in")

    def pieces(self):
        yield from self.main.pieces(1)

    # ---- .sym  (circom's columns: signal number, witness index, component index, name; at --O0 signal == witness index)
    def sym_lines(self):
        for w, c, name in self.main.names("main", 1, [0]):
            yield f"{w},{w},{c},{name}\n"

    def write_sym(self, path: str) -> str:
        h = hashlib.sha256()
        with open(path, "w") as f:
            buf = []
            for line in self.sym_lines():
                buf.append(line)
                if len(buf) >= 1 << 16:
                    s = "".join(buf); f.write(s); h.update(s.encode()); buf = []
            s = "".join(buf); f.write(s); h.update(s.encode())
        return h.hexdigest()

    # ---- .r1cs (iden3 binary format v1: header, constraints, wire2label)
    def write_r1cs(self, path: str):
        coef_bytes = {}

        def cb(i):
            b = coef_bytes.get(i)
            if b is None:
                b = COEF[i].to_bytes(32, "little")
                coef_bytes[i] = b
            return b

        with open(path, "wb") as f:
            f.write(b"r1cs" + struct.pack("<II", 1, 3))
            hdr = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIIIQI", self.n_wires, self.n_outputs, 0, self.n_inputs, self.n_wires, self.n_constraints)
            f.write(struct.pack("<IQ", 1, len(hdr)) + hdr)
            pos_len = f.tell() + 4
            f.write(struct.pack("<IQ", 2, 0))
            start = f.tell()
            for fl, base in self.pieces():
                for r in range(fl.n):
                    for s in range(3):
                        a, b = int(fl.ptr[s][r]), int(fl.ptr[s][r + 1])
                        f.write(struct.pack("<I", b - a))
                        for j in range(a, b):
                            i = int(fl.idx[s][j])
                            f.write(struct.pack("<I", 0 if i < 0 else i + base) + cb(int(fl.cid[s][j])))
            end = f.tell()
            f.write(struct.pack("<IQ", 3, 8 * self.n_wires))
            f.write(np.arange(self.n_wires, dtype="<u8").tobytes())
            f.seek(pos_len)
            f.write(struct.pack("<Q", end - start))
