"""circuits/utils/{assert,array,divide,selector,convert,public_commitment}.circom of the reference as circuit-model templates
(citations: file:line of the reference).  Anonymous components are named `<Template>_<line>`."""
from __future__ import annotations

from .core import P, Template
from .lib import Bits2Num, GreaterEqThan, IsEqual, LessEqThan, LessThan, Num2Bits, Num2Bits_strict


# ------------------------------------------------------------------------------------------------ assert.circom
class AssertBits(Template):            # :13-17
    def build(self, nbits):
        assert nbits < 254
        inp = self.input("in"); bits = self.signal("bits", nbits)
        n2b = self.comp("Num2Bits_16", Num2Bits.get(nbits))
        self.assign(n2b["in"].lc, inp.lc); self.copy(bits, n2b["out"])


class AssertByteString(Template):      # :26-31
    def build(self, n):
        inp = self.input("in", n)
        for i in range(n):
            a = self.comp(f"AssertBits_29[{i}]", AssertBits.get(8))
            self.assign(a["in"].lc, inp[i])


def _assert_cmp(cmp_tpl, line):
    class AssertCmp(Template):         # :40-47 / :56-63 / :72-79   out === 1
        def build(self, nbits):
            a = self.input("a"); b = self.input("b"); out = self.signal("out")
            x = self.comp(f"AssertBits_{line - 2}", AssertBits.get(nbits)); self.assign(x["in"].lc, a.lc)
            y = self.comp(f"AssertBits_{line - 1}", AssertBits.get(nbits)); self.assign(y["in"].lc, b.lc)
            c = self.comp(f"{cmp_tpl.__name__}_{line}", cmp_tpl.get(nbits))
            self.assign(c["in"][0], a.lc); self.assign(c["in"][1], b.lc); self.assign(out.lc, c["out"].lc)
            self.eq(out.lc, 1)
    AssertCmp.__name__ = "Assert" + cmp_tpl.__name__
    return AssertCmp


AssertLessThan = _assert_cmp(LessThan, 45)
AssertLessEqThan = _assert_cmp(LessEqThan, 61)
AssertGreaterEqThan = _assert_cmp(GreaterEqThan, 77)


# ------------------------------------------------------------------------------------------------ array.circom
class Filter(Template):                # :26-40
    def build(self, n):
        inp = self.input("in"); out = self.output("out", n); is_eq = self.signal("isEq", n)
        for i in range(n):
            e = self.comp(f"IsEqual_32[{i}]", IsEqual.get())
            self.assign(e["in"][0], i); self.assign(e["in"][1], inp.lc); self.assign(is_eq[i], e["out"].lc)
            if i > 0:
                self.mul(out[i], out[i - 1], 1 - is_eq[i])
            else:
                self.assign(out[i], 1 - is_eq[i])


class Fit(Template):                   # :47-57
    def build(self, m, n):
        inp = self.input("in", m); out = self.output("out", n)
        for i in range(n):
            self.assign(out[i], inp[i] if i < m else 0)


class Flatten(Template):               # :64-72
    def build(self, m, n):
        inp = self.input("in", m, n); out = self.output("out", m * n)
        for k in range(m * n):
            self.assign(out[k], inp.w(k))


class Reshape(Template):               # :79-87
    def build(self, m, n):
        inp = self.input("in", m * n); out = self.output("out", m, n)
        for k in range(m * n):
            self.assign(out.w(k), inp[k])


class Reverse(Template):               # :94-100
    def build(self, n):
        inp = self.input("in", n); out = self.output("out", n)
        for i in range(n):
            self.assign(out[i], inp[n - 1 - i])


# ------------------------------------------------------------------------------------------------ divide.circom
class Divide(Template):                # :17-33  out <-- a \ b; rem <-- a % b; out*b + rem === a
    def build(self, n):
        a = self.input("a"); b = self.input("b"); out = self.output("out"); rem = self.output("rem")
        lt = self.comp("AssertLessThan_27", AssertLessThan.get(n)); self.assign(lt["a"].lc, rem.lc); self.assign(lt["b"].lc, b.lc)
        le = self.comp("AssertLessEqThan_30", AssertLessEqThan.get(n)); self.assign(le["a"].lc, out.lc); self.assign(le["b"].lc, a.lc)
        self.constrain(out.lc, b.lc, a.lc - rem.lc)


# ------------------------------------------------------------------------------------------------ selector.circom
class Selector(Template):              # :21-46
    def build(self, n):
        vals = self.input("vals", n); select = self.input("select"); out = self.output("out")
        is_eq = self.signal("isEq", n); s = self.signal("sum", n + 1)
        self.assign(s[0], 0)
        total = 0
        for i in range(n):
            e = self.comp(f"IsEqual_35[{i}]", IsEqual.get())
            self.assign(e["in"][0], select.lc); self.assign(e["in"][1], i); self.assign(is_eq[i], e["out"].lc)
            total = is_eq[i] + total
            self.mul(s[i + 1], is_eq[i], vals[i], s[i])
        self.eq(total, 1)
        self.assign(out.lc, s[n])


class SelectorArray1D(Template):       # :62-77
    def build(self, n, p):
        arrays = self.input("arrays", n, p); select = self.input("select"); out = self.output("out", p); tr = self.signal("arraysT", p, n)
        for i in range(n):
            for j in range(p):
                self.assign(tr[j, i], arrays[i, j])
        for i in range(p):
            s = self.comp(f"Selector_75[{i}]", Selector.get(n))
            for k in range(n):
                self.assign(s["vals"][k], tr[i, k])
            self.assign(s["select"].lc, select.lc); self.assign(out[i], s["out"].lc)


class SelectorArray2D(Template):       # :91-111
    def build(self, n, p, q):
        arrays = self.input("arrays", n, p, q); select = self.input("select"); out = self.output("out", p, q); tr = self.signal("arraysT", p, q, n)
        for i in range(n):
            for j in range(p):
                for k in range(q):
                    self.assign(tr[j, k, i], arrays[i, j, k])
        for i in range(p):
            for j in range(q):
                s = self.comp(f"Selector_107[{i}][{j}]", Selector.get(n))
                for k in range(n):
                    self.assign(s["vals"][k], tr[i, j, k])
                self.assign(s["select"].lc, select.lc); self.assign(out[i, j], s["out"].lc)


# ------------------------------------------------------------------------------------------------ convert.circom
class LittleEndianBytes2Num(Template):  # :12-26
    def build(self, n):
        assert n <= 31
        inp = self.input("in", n); out = self.output("out")
        a = self.comp("AssertByteString_18", AssertByteString.get(n)); self.copy(a["in"], inp)
        lc = 0
        for i in range(n):
            lc = inp[i] * pow(256, i, P) + lc
        self.assign(out.lc, lc)


class BigEndianBytes2Num(Template):    # :33-39
    def build(self, n):
        inp = self.input("in", n); out = self.output("out"); rev = self.signal("inReversed", n)
        r = self.comp("Reverse_37", Reverse.get(n)); self.copy(r["in"], inp); self.copy(rev, r["out"])
        le = self.comp("LittleEndianBytes2Num_38", LittleEndianBytes2Num.get(n)); self.copy(le["in"], rev); self.assign(out.lc, le["out"].lc)


class Num2BitsSafe(Template):          # :46-56
    def build(self, n):
        inp = self.input("in"); out = self.output("out", n)
        if n >= 254:
            bs = self.signal("bitsStrict", 254)
            s = self.comp("Num2Bits_strict_51", Num2Bits_strict.get()); self.assign(s["in"].lc, inp.lc); self.copy(bs, s["out"])
            f = self.comp("Fit_52", Fit.get(254, n)); self.copy(f["in"], bs); self.copy(out, f["out"])
        else:
            b = self.comp("Num2Bits_54", Num2Bits.get(n)); self.assign(b["in"].lc, inp.lc); self.copy(out, b["out"])


class Num2LittleEndianBytes(Template):  # :69-83
    def build(self, n):
        assert n <= 32
        inp = self.input("in"); out = self.output("out", n); bits = self.signal("bits", 8 * n); ba = self.signal("byteArrays", n, 8)
        s = self.comp("Num2BitsSafe_76", Num2BitsSafe.get(8 * n)); self.assign(s["in"].lc, inp.lc); self.copy(bits, s["out"])
        r = self.comp("Reshape_77", Reshape.get(n, 8)); self.copy(r["in"], bits); self.copy(ba, r["out"])
        for i in range(n):
            b = self.comp(f"Bits2Num_81[{i}]", Bits2Num.get(8))
            for k in range(8):
                self.assign(b["in"][k], ba[i, k])
            self.assign(out[i], b["out"].lc)


class Num2BigEndianBytes(Template):    # :90-96
    def build(self, n):
        inp = self.input("in"); out = self.output("out", n); le = self.signal("littleEndian", n)
        c = self.comp("Num2LittleEndianBytes_94", Num2LittleEndianBytes.get(n)); self.assign(c["in"].lc, inp.lc); self.copy(le, c["out"])
        r = self.comp("Reverse_95", Reverse.get(n)); self.copy(r["in"], le); self.copy(out, r["out"])


class Bytes2Nibbles(Template):         # :103-121
    def build(self, n):
        inp = self.input("in", n); out = self.output("out", 2 * n); dec = self.signal("inDecomposed", n, 8)
        for i in range(n):
            b = self.comp(f"Num2Bits_110[{i}]", Num2Bits.get(8)); self.assign(b["in"].lc, inp[i])
            lower = higher = 0
            for j in range(8):
                self.assign(dec[i, j], b["out"][j])
            for j in range(4):
                lower = dec[i, j] * (1 << j) + lower
                higher = dec[i, j + 4] * (1 << j) + higher
            self.assign(out[2 * i], higher); self.assign(out[2 * i + 1], lower)


class Nibbles2Bytes(Template):         # :132-142
    def build(self, n):
        nib = self.input("nibbles", 2 * n); by = self.output("bytes", n)
        for i in range(n):
            a = self.comp(f"AssertBits_137[{i}]", AssertBits.get(4)); self.assign(a["in"].lc, nib[2 * i])
            b = self.comp(f"AssertBits_138[{i}]", AssertBits.get(4)); self.assign(b["in"].lc, nib[2 * i + 1])
            self.assign(by[i], nib[2 * i] * 16 + nib[2 * i + 1])
