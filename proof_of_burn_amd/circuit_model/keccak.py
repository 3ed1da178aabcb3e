"""circuits/utils/keccak.circom of the reference as circuit-model templates (line numbers refer to that file).
Anonymous components `T(p)(args)` are named `<Template>_<line>` (+ `[i]...` inside loops): circom's own naming of anonymous
components in the .sym is not reproduced (unpinned)."""
from __future__ import annotations

from .core import Template
from .lib import AND, OR, XOR, Bits2Num, IsEqual, Num2Bits
from .utils import AssertLessEqThan, AssertLessThan, Divide, Flatten, Reshape, SelectorArray2D

RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
      0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
ROT = [1, 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1]


class ShR(Template):           # :19-31  out[i] = i + r >= n ? 0 : in[i + r]
    def build(self, n, r):
        inp = self.input("in", n); out = self.output("out", n)
        for i in range(n):
            self.assign(out[i], 0 if i + r >= n else inp[i + r])


class ShL(Template):           # :38-51  out[i] = i < r ? 0 : in[i - r]
    def build(self, n, r):
        inp = self.input("in", n); out = self.output("out", n)
        for i in range(n):
            self.assign(out[i], 0 if i < r else inp[i - r])


def _gate_array(gate, gname, line):
    class GateArray(Template):  # :77-111  out[i] <== GATE()(a[i], b[i])
        def build(self, n):
            a = self.input("a", n); b = self.input("b", n); out = self.output("out", n)
            for i in range(n):
                g = self.comp(f"{gname}_{line}[{i}]", gate.get())
                self.assign(g["a"].lc, a[i]); self.assign(g["b"].lc, b[i]); self.assign(out[i], g["out"].lc)
    GateArray.__name__ = gname + "Array"
    return GateArray


XorArray = _gate_array(XOR, "XOR", 82)
OrArray = _gate_array(OR, "OR", 105)
AndArray = _gate_array(AND, "AND", 117)


class NotArray(Template):      # :89-96  out[i] <== 1 - a[i]
    def build(self, n):
        a = self.input("a", n); out = self.output("out", n)
        for i in range(n):
            self.assign(out[i], 1 - a[i])


class Xor5(Template):          # :58-70
    def build(self, n):
        ins = [self.input(x, n) for x in "abcde"]; out = self.output("out", n)
        mids = [self.signal(x, n) for x in ("xor_ab", "xor_abc", "xor_abcd")]
        prev = ins[0]
        for k, (line, dst) in enumerate(zip((66, 67, 68, 69), mids + [out])):
            x = self.comp(f"XorArray_{line}", XorArray.get(n))
            self.copy(x["a"], prev); self.copy(x["b"], ins[k + 1]); self.copy(dst, x["out"])
            prev = dst


class D(Template):             # :135-144
    def build(self):
        a = self.input("a", 64); b = self.input("b", 64); out = self.output("out", 64)
        aux0 = self.signal("aux0", 64); aux1 = self.signal("aux1", 64); aux2 = self.signal("aux2", 64)
        shl = self.comp("ShL_140", ShL.get(64, 1)); self.copy(shl["in"], a); self.copy(aux0, shl["out"])
        shr = self.comp("ShR_141", ShR.get(64, 63)); self.copy(shr["in"], a); self.copy(aux1, shr["out"])
        o = self.comp("OrArray_142", OrArray.get(64)); self.copy(o["a"], aux0); self.copy(o["b"], aux1); self.copy(aux2, o["out"])
        x = self.comp("XorArray_143", XorArray.get(64)); self.copy(x["a"], b); self.copy(x["b"], aux2); self.copy(out, x["out"])


def _row(sig, i):
    """the 64 wires of row i of a [25][64] (or [n][64]) signal view"""
    return [sig.w(64 * i + k) for k in range(64)]


class Theta(Template):         # :151-170
    def build(self):
        inp = self.input("in", 25, 64); out = self.output("out", 25, 64); c = self.signal("c", 5, 64); d = self.signal("d", 5, 64)
        for i in range(5):
            x = self.comp(f"Xor5_157[{i}]", Xor5.get(64))
            for k, name in enumerate("abcde"):
                self.copy(x[name], _row(inp, 5 * k + i))
            self.copy(_row(c, i), x["out"])
        for i in range(5):
            dd = self.comp(f"D_162[{i}]", D.get())
            self.copy(dd["a"], _row(c, (i + 1) % 5)); self.copy(dd["b"], _row(c, (i + 4) % 5)); self.copy(_row(d, i), dd["out"])
        for i in range(5):
            for j in range(5):
                x = self.comp(f"XorArray_167[{i}][{j}]", XorArray.get(64))
                self.copy(x["a"], _row(inp, i + 5 * j)); self.copy(x["b"], _row(d, i)); self.copy(_row(out, i + 5 * j), x["out"])


class stepRhoPi(Template):     # :177-184
    def build(self, shl, shr):
        a = self.input("a", 64); out = self.output("out", 64); aux0 = self.signal("aux0", 64); aux1 = self.signal("aux1", 64)
        r = self.comp("ShR_181", ShR.get(64, shr)); self.copy(r["in"], a); self.copy(aux0, r["out"])
        l_ = self.comp("ShL_182", ShL.get(64, shl)); self.copy(l_["in"], a); self.copy(aux1, l_["out"])
        o = self.comp("OrArray_183", OrArray.get(64)); self.copy(o["a"], aux0); self.copy(o["b"], aux1); self.copy(out, o["out"])


class RhoPi(Template):         # :191-204
    def build(self):
        inp = self.input("in", 25, 64); out = self.output("out", 25, 64)
        self.copy(_row(out, 0), _row(inp, 0))
        for i in range(24):
            shl = ((i + 1) * (i + 2) // 2) % 64
            s = self.comp(f"stepRhoPi_202[{i}]", stepRhoPi.get(shl, 64 - shl))
            self.copy(s["a"], _row(inp, ROT[i])); self.copy(_row(out, ROT[i + 1]), s["out"])


class stepChi(Template):       # :212-221
    def build(self):
        a = self.input("a", 64); b = self.input("b", 64); c = self.input("c", 64); out = self.output("out", 64)
        bx = self.signal("bXor", 64); bc = self.signal("bc", 64)
        n = self.comp("NotArray_218", NotArray.get(64)); self.copy(n["a"], b); self.copy(bx, n["out"])
        an = self.comp("AndArray_219", AndArray.get(64)); self.copy(an["a"], bx); self.copy(an["b"], c); self.copy(bc, an["out"])
        x = self.comp("XorArray_220", XorArray.get(64)); self.copy(x["a"], a); self.copy(x["b"], bc); self.copy(out, x["out"])


class Chi(Template):           # :228-241
    def build(self):
        inp = self.input("in", 25, 64); out = self.output("out", 25, 64)
        for i in range(25):
            y = i // 5 * 5
            s = self.comp(f"stepChi_{234 if i % 5 == 3 else 236 if i % 5 == 4 else 238}[{i}]", stepChi.get())
            self.copy(s["a"], _row(inp, i)); self.copy(s["b"], _row(inp, y + (i + 1) % 5)); self.copy(s["c"], _row(inp, y + (i + 2) % 5))
            self.copy(_row(out, i), s["out"])


class RoundConstants(Template):  # :248-266
    def build(self, r):
        out = self.output("out", 64)
        for i in range(64):
            self.assign(out[i], (RC[r] >> i) & 1)


class Iota(Template):          # :273-283
    def build(self, r):
        inp = self.input("in", 25, 64); out = self.output("out", 25, 64); rc = self.signal("roundConstants", 64)
        c = self.comp("RoundConstants_277", RoundConstants.get(r)); self.copy(rc, c["out"])
        x = self.comp("XorArray_278", XorArray.get(64)); self.copy(x["a"], _row(inp, 0)); self.copy(x["b"], rc); self.copy(_row(out, 0), x["out"])
        for i in range(1, 25):
            self.copy(_row(out, i), _row(inp, i))


class KeccakfRound(Template):  # :290-297
    def build(self, r):
        inp = self.input("in", 25, 64); out = self.output("out", 25, 64)
        th = self.signal("theta", 25, 64); rp = self.signal("rhopi", 25, 64); ch = self.signal("chi", 25, 64)
        t = self.comp("Theta_293", Theta.get()); self.copy(t["in"], inp); self.copy(th, t["out"])
        p = self.comp("RhoPi_294", RhoPi.get()); self.copy(p["in"], th); self.copy(rp, p["out"])
        c = self.comp("Chi_295", Chi.get()); self.copy(c["in"], rp); self.copy(ch, c["out"])
        i = self.comp("Iota_296", Iota.get(r)); self.copy(i["in"], ch); self.copy(out, i["out"])


class Keccakf(Template):       # :356-367
    def build(self):
        inp = self.input("in", 25, 64); out = self.output("out", 25, 64); mid = self.signal("midRound", 25, 25, 64)
        for k in range(1600):
            self.assign(mid.w(k), inp.w(k))
        for i in range(24):
            rnd = self.comp(f"KeccakfRound_364[{i}]", KeccakfRound.get(i))
            for k in range(1600):
                self.assign(rnd["in"].w(k), mid.w(1600 * i + k))
                self.assign(mid.w(1600 * (i + 1) + k), rnd["out"].w(k))
        for k in range(1600):
            self.assign(out.w(k), mid.w(1600 * 24 + k))


class Absorb(Template):        # :304-323
    def build(self):
        s = self.input("s", 25, 64); block = self.input("block", 17, 64); out = self.output("out", 25, 64); aux = self.signal("aux", 25, 64)
        for i in range(25):
            if i < 17:
                x = self.comp(f"XorArray_317[{i}]", XorArray.get(64))
                self.copy(x["a"], _row(s, i)); self.copy(x["b"], _row(block, i)); self.copy(_row(aux, i), x["out"])
            else:
                self.copy(_row(aux, i), _row(s, i))
        f = self.comp("Keccakf_322", Keccakf.get()); self.copy(f["in"], aux); self.copy(out, f["out"])


class Final(Template):         # :330-349
    def build(self, n):
        inp = self.input("in", n, 17, 64); blocks = self.input("blocks"); out = self.output("out", 25, 64); s = self.signal("s", n + 1, 25, 64)
        for k in range(1600):
            self.assign(s.w(k), 0)
        for b in range(n):
            a = self.comp(f"Absorb_344[{b}]", Absorb.get())
            for k in range(1600):
                self.assign(a["s"].w(k), s.w(1600 * b + k))
            for k in range(1088):
                self.assign(a["block"].w(k), inp.w(1088 * b + k))
            for k in range(1600):
                self.assign(s.w(1600 * (b + 1) + k), a["out"].w(k))
        sel = self.comp("SelectorArray2D_348", SelectorArray2D.get(n + 1, 25, 64))
        self.copy(sel["arrays"], s); self.assign(sel["select"].lc, blocks.lc); self.copy(out, sel["out"])


class Keccak(Template):        # :374-385
    def build(self, n):
        inp = self.input("in", n, 17, 64); blocks = self.input("blocks"); out = self.output("out", 256); fs = self.signal("finalState", 25, 64)
        f = self.comp("Final_380", Final.get(n)); self.copy(f["in"], inp); self.assign(f["blocks"].lc, blocks.lc); self.copy(fs, f["out"])
        for i in range(256):
            self.assign(out[i], fs.w(i))


class Pad(Template):           # :412-446
    def build(self, max_blocks, block_size):
        m = max_blocks * block_size
        inp = self.input("in", m); in_len = self.input("inLen"); out = self.output("out", m); nb = self.output("numBlocks")
        div = self.signal("div"); rem = self.signal("rem"); flt = self.signal("filter", m + 1); is_eq = self.signal("isEq", m); is_last = self.signal("isLast", m)
        dv = self.comp("Divide_420", Divide.get(16))
        self.assign(dv["a"].lc, in_len.lc); self.assign(dv["b"].lc, block_size)
        self.assign(div.lc, dv["out"].lc); self.assign(rem.lc, dv["rem"].lc)
        self.assign(nb.lc, div.lc + 1)
        le = self.comp("AssertLessEqThan_423", AssertLessEqThan.get(16)); self.assign(le["a"].lc, nb.lc); self.assign(le["b"].lc, max_blocks)
        self.assign(flt[0], 1)
        for i in range(m):
            e = self.comp(f"IsEqual_429[{i}]", IsEqual.get())
            self.assign(e["in"][0], i); self.assign(e["in"][1], in_len.lc); self.assign(is_eq[i], e["out"].lc)
            self.mul(flt[i + 1], flt[i], 1 - is_eq[i])
        for i in range(m):
            e = self.comp(f"IsEqual_443[{i}]", IsEqual.get())
            self.assign(e["in"][0], i); self.assign(e["in"][1], nb.lc * block_size - 1); self.assign(is_last[i], e["out"].lc)
            self.mul(out[i], inp[i], flt[i + 1], is_eq[i] + is_last[i] * 0x80)


class KeccakBytes(Template):   # :454-489
    def build(self, max_blocks):
        m = max_blocks * 136
        inp = self.input("in", m); in_len = self.input("inLen"); out = self.output("out", 32)
        padded = self.signal("padded", m); nb = self.signal("numBlocks"); bits_arr = self.signal("inBitsArray", m, 8); bits = self.signal("inBits", 8 * m)
        blocks = self.signal("inBlocks", max_blocks, 17, 64); out_bits = self.signal("outBits", 256); out_bytes = self.signal("outBytes", 32, 8)
        lt = self.comp("AssertLessThan_460", AssertLessThan.get(16)); self.assign(lt["a"].lc, in_len.lc); self.assign(lt["b"].lc, m)
        pad = self.comp("Pad_463", Pad.get(max_blocks, 136))
        self.copy(pad["in"], inp); self.assign(pad["inLen"].lc, in_len.lc); self.copy(padded, pad["out"]); self.assign(nb.lc, pad["numBlocks"].lc)
        for i in range(m):
            n2b = self.comp(f"Num2Bits_470[{i}]", Num2Bits.get(8))
            self.assign(n2b["in"].lc, padded[i])
            for k in range(8):
                self.assign(bits_arr[i, k], n2b["out"][k])
        fl = self.comp("Flatten_473", Flatten.get(m, 8)); self.copy(fl["in"], bits_arr); self.copy(bits, fl["out"])
        for k in range(8 * m):
            self.assign(blocks.w(k), bits.w(k))
        kc = self.comp("Keccak_484", Keccak.get(max_blocks))
        self.copy(kc["in"], blocks); self.assign(kc["blocks"].lc, nb.lc); self.copy(out_bits, kc["out"])
        rs = self.comp("Reshape_485", Reshape.get(32, 8)); self.copy(rs["in"], out_bits); self.copy(out_bytes, rs["out"])
        for i in range(32):
            b2n = self.comp(f"Bits2Num_487[{i}]", Bits2Num.get(8))
            for k in range(8):
                self.assign(b2n["in"][k], out_bytes[i, k])
            self.assign(out[i], b2n["out"].lc)
