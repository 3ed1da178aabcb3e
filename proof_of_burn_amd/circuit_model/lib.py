"""circomlib templates the circuits include (UNVENDORED submodule of the reference; restated from the published circomlib v2.0.5
circuits: gates, bitify, comparators, mux1, compconstant, aliascheck, poseidon) as circuit-model templates.
Signal order inside a template = circom's: outputs | inputs | intermediates; children in initialisation order (for the explicitly
declared components of circomlib: the statement that assigns their last input -- Num2Bits_strict: n2b then aliasCheck;
MultiAND: ands[0], ands[1], and2)."""
from __future__ import annotations

import os

from .. import poseidon_host as PH
from .core import P, Template

# The one numbering switch of the three statements of the O0 layout (csrc/policy.hpp POB_DECL_ORDER; the oracle's ORACLE_DECL_ORDER):
# 1 = the two circomlib templates whose children are DECLARED in another order than they are initialised follow the declaration order.
DECL_ORDER = os.environ.get("POB_DECL_ORDER", "0")[:1] == "1"


class XOR(Template):           # out <== a + b - 2*a*b
    def build(self):
        out = self.output("out"); a = self.input("a"); b = self.input("b")
        self.mul(out.lc, a.lc * -2, b.lc, a.lc + b.lc)


class AND(Template):           # out <== a*b
    def build(self):
        out = self.output("out"); a = self.input("a"); b = self.input("b")
        self.mul(out.lc, a.lc, b.lc)


class OR(Template):            # out <== a + b - a*b
    def build(self):
        out = self.output("out"); a = self.input("a"); b = self.input("b")
        self.mul(out.lc, a.lc * -1, b.lc, a.lc + b.lc)


class Num2Bits(Template):      # out[i] <-- (in >> i) & 1; out[i]*(out[i]-1) === 0; sum out[i] 2^i === in
    def build(self, n):
        out = self.output("out", n); inp = self.input("in")
        lc = 0
        for i in range(n):
            self.constrain(out[i], out[i] - 1, 0)
            lc = out[i] * pow(2, i, P) + lc
        self.eq(lc, inp.lc)


class Bits2Num(Template):      # out <== sum in[i] 2^i
    def build(self, n):
        out = self.output("out"); inp = self.input("in", n)
        lc = 0
        for i in range(n):
            lc = inp[i] * pow(2, i, P) + lc
        self.assign(out.lc, lc)


class IsZero(Template):        # inv <-- in != 0 ? 1/in : 0;  out <== -in*inv + 1;  in*out === 0
    def build(self):
        out = self.output("out"); inp = self.input("in"); inv = self.signal("inv")
        self.mul(out.lc, inp.lc * -1, inv.lc, 1)
        self.constrain(inp.lc, out.lc, 0)


class IsEqual(Template):       # isz.in <== in[1] - in[0]; out <== isz.out
    def build(self):
        out = self.output("out"); inp = self.input("in", 2)
        isz = self.comp("isz", IsZero.get())
        self.assign(isz["in"].lc, inp[1] - inp[0])
        self.assign(out.lc, isz["out"].lc)


class LessThan(Template):      # n2b.in <== in[0] + (1 << n) - in[1]; out <== 1 - n2b.out[n]
    def build(self, n):
        assert n <= 252
        out = self.output("out"); inp = self.input("in", 2)
        n2b = self.comp("n2b", Num2Bits.get(n + 1))
        self.assign(n2b["in"].lc, inp[0] + (1 << n) - inp[1])
        self.assign(out.lc, 1 - n2b["out"][n])


class LessEqThan(Template):    # lt.in <== [in[0], in[1] + 1]
    def build(self, n):
        out = self.output("out"); inp = self.input("in", 2)
        lt = self.comp("lt", LessThan.get(n))
        self.assign(lt["in"][0], inp[0]); self.assign(lt["in"][1], inp[1] + 1)
        self.assign(out.lc, lt["out"].lc)


class GreaterEqThan(Template):  # lt.in <== [in[1], in[0] + 1]
    def build(self, n):
        out = self.output("out"); inp = self.input("in", 2)
        lt = self.comp("lt", LessThan.get(n))
        self.assign(lt["in"][0], inp[1]); self.assign(lt["in"][1], inp[0] + 1)
        self.assign(out.lc, lt["out"].lc)


class MultiAND(Template):
    def build(self, n):
        out = self.output("out"); inp = self.input("in", n)
        if n == 1:
            self.assign(out.lc, inp[0])
        elif n == 2:
            a = self.comp("and1", AND.get())
            self.assign(a["a"].lc, inp[0]); self.assign(a["b"].lc, inp[1]); self.assign(out.lc, a["out"].lc)
        else:
            n1, n2 = n // 2, n - n // 2
            if DECL_ORDER:                   # and2 is DECLARED first (circomlib gates.circom); default: initialisation order, see DECL_ORDER
                a2 = self.comp("and2", AND.get()); m0 = self.comp("ands[0]", MultiAND.get(n1)); m1 = self.comp("ands[1]", MultiAND.get(n2))
            else:
                m0 = self.comp("ands[0]", MultiAND.get(n1)); m1 = self.comp("ands[1]", MultiAND.get(n2)); a2 = self.comp("and2", AND.get())
            for i in range(n1):
                self.assign(m0["in"][i], inp[i])
            for i in range(n2):
                self.assign(m1["in"][i], inp[n1 + i])
            self.assign(a2["a"].lc, m0["out"].lc); self.assign(a2["b"].lc, m1["out"].lc); self.assign(out.lc, a2["out"].lc)


class MultiMux1(Template):     # out[i] <== (c[i][1] - c[i][0])*s + c[i][0]
    def build(self, n):
        out = self.output("out", n); c = self.input("c", n, 2); s = self.input("s")
        for i in range(n):
            self.mul(out[i], c[i, 1] - c[i, 0], s.lc, c[i, 0])


class Mux1(Template):
    def build(self):
        out = self.output("out"); c = self.input("c", 2); s = self.input("s")
        mux = self.comp("mux", MultiMux1.get(1))
        for i in range(2):
            self.assign(mux["c"][0, i], c[i])
        self.assign(mux["s"].lc, s.lc)
        self.assign(out.lc, mux["out"][0])


class CompConstant(Template):  # out = (in > ct) over 254-bit little-endian bits (compconstant.circom)
    def build(self, ct):
        out = self.output("out"); inp = self.input("in", 254); parts = self.signal("parts", 127); sout = self.signal("sout")
        b = (1 << 128) - 1; a = 1; e = 1
        total = 0
        for i in range(127):
            clsb, cmsb = (ct >> (2 * i)) & 1, (ct >> (2 * i + 1)) & 1
            slsb, smsb = inp[2 * i], inp[2 * i + 1]
            if not cmsb and not clsb:      # parts <== -b*smsb*slsb + b*smsb + b*slsb
                self.mul(parts[i], smsb * (-b), slsb, smsb * b + slsb * b)
            elif not cmsb and clsb:        # a*smsb*slsb - a*slsb + b*smsb - a*smsb + a
                self.mul(parts[i], smsb * a, slsb, slsb * (-a) + smsb * b - smsb * a + a)
            elif cmsb and not clsb:        # b*smsb*slsb - a*smsb + a
                self.mul(parts[i], smsb * b, slsb, smsb * (-a) + a)
            else:                          # -a*smsb*slsb + a
                self.mul(parts[i], smsb * (-a), slsb, a)
            total = parts[i] + total
            b -= e; a += e; e *= 2
        self.assign(sout.lc, total)
        n2b = self.comp("num2bits", Num2Bits.get(135))
        self.assign(n2b["in"].lc, sout.lc)
        self.assign(out.lc, n2b["out"][127])


class AliasCheck(Template):    # compConstant(p - 1).out === 0
    def build(self):
        inp = self.input("in", 254)
        cc = self.comp("compConstant", CompConstant.get(P - 1))
        for i in range(254):
            self.assign(cc["in"][i], inp[i])
        self.eq(cc["out"].lc, 0)


class Num2Bits_strict(Template):
    def build(self):
        out = self.output("out", 254); inp = self.input("in")
        if DECL_ORDER:                       # aliasCheck is DECLARED first (circomlib bitify.circom)
            ac = self.comp("aliasCheck", AliasCheck.get()); n2b = self.comp("n2b", Num2Bits.get(254))
        else:
            n2b = self.comp("n2b", Num2Bits.get(254)); ac = self.comp("aliasCheck", AliasCheck.get())
        self.assign(n2b["in"].lc, inp.lc)
        for i in range(254):
            self.assign(out[i], n2b["out"][i])
            self.assign(ac["in"][i], n2b["out"][i])


# ---------------------------------------------------------------------------------------------------- poseidon.circom (optimised)
class Sigma(Template):         # in2 <== in*in; in4 <== in2*in2; out <== in4*in
    def build(self):
        out = self.output("out"); inp = self.input("in"); in2 = self.signal("in2"); in4 = self.signal("in4")
        self.mul(in2.lc, inp.lc, inp.lc); self.mul(in4.lc, in2.lc, in2.lc); self.mul(out.lc, in4.lc, inp.lc)


class Ark(Template):           # out[i] <== in[i] + C[i + r]
    def build(self, t, r):
        C = PH.optimized(t)[0]
        out = self.output("out", t); inp = self.input("in", t)
        for i in range(t):
            self.assign(out[i], inp[i] + C[i + r])


class Mix(Template):           # out = A @ in   (which = "M": the MDS matrix, "P": the pre-matrix of the partial rounds)
    def build(self, t, which):
        A = PH.optimized(t)[2 if which == "M" else 3]
        out = self.output("out", t); inp = self.input("in", t)
        for i in range(t):
            lc = 0
            for j in range(t):
                lc = inp[j] * A[i][j] + lc
            self.assign(out[i], lc)


class MixS(Template):          # out[0] = sum S[base+i] in[i];  out[i] = in[i] + in[0]*S[base+t+i-1]
    def build(self, t, r):
        S = PH.optimized(t)[1]
        base = (2 * t - 1) * r
        out = self.output("out", t); inp = self.input("in", t)
        lc = 0
        for i in range(t):
            lc = inp[i] * S[base + i] + lc
        self.assign(out[0], lc)
        for i in range(1, t):
            self.assign(out[i], inp[i] + inp[0] * S[base + t + i - 1])


class MixLast(Template):       # out = row s of the MDS matrix times in
    def build(self, t, s):
        A = PH.optimized(t)[2]
        out = self.output("out"); inp = self.input("in", t)
        lc = 0
        for j in range(t):
            lc = inp[j] * A[s][j] + lc
        self.assign(out.lc, lc)


class PoseidonEx(Template):
    def build(self, n_inputs, n_outs):
        assert n_outs == 1
        t = n_inputs + 1
        rp = PH.R_P_TABLE[t]
        C = PH.optimized(t)[0]
        out = self.output("out", n_outs); inputs = self.input("inputs", n_inputs); init = self.input("initialState")
        ark0 = self.comp("ark[0]", Ark.get(t, 0))
        for j in range(t):
            self.assign(ark0["in"][j], inputs[j - 1] if j > 0 else init.lc)
        prev = ark0
        for r in range(3):                                         # first half of the full rounds but the last
            sig = [self.comp(f"sigmaF[{r}][{j}]", Sigma.get()) for j in range(t)]
            for j in range(t):
                self.assign(sig[j]["in"].lc, prev["out"][j])
            ark = self.comp(f"ark[{r + 1}]", Ark.get(t, (r + 1) * t))
            for j in range(t):
                self.assign(ark["in"][j], sig[j]["out"].lc)
            mix = self.comp(f"mix[{r}]", Mix.get(t, "M"))
            for j in range(t):
                self.assign(mix["in"][j], ark["out"][j])
            prev = mix
        sig = [self.comp(f"sigmaF[3][{j}]", Sigma.get()) for j in range(t)]
        for j in range(t):
            self.assign(sig[j]["in"].lc, prev["out"][j])
        ark = self.comp("ark[4]", Ark.get(t, 4 * t))
        for j in range(t):
            self.assign(ark["in"][j], sig[j]["out"].lc)
        mix = self.comp("mix[3]", Mix.get(t, "P"))
        for j in range(t):
            self.assign(mix["in"][j], ark["out"][j])
        prev = mix
        for r in range(rp):                                        # partial rounds
            sp = self.comp(f"sigmaP[{r}]", Sigma.get())
            self.assign(sp["in"].lc, prev["out"][0])
            ms = self.comp(f"mixS[{r}]", MixS.get(t, r))
            self.assign(ms["in"][0], sp["out"].lc + C[5 * t + r])
            for j in range(1, t):
                self.assign(ms["in"][j], prev["out"][j])
            prev = ms
        for r in range(3):                                         # second half of the full rounds but the last
            sig = [self.comp(f"sigmaF[{4 + r}][{j}]", Sigma.get()) for j in range(t)]
            for j in range(t):
                self.assign(sig[j]["in"].lc, prev["out"][j])
            ark = self.comp(f"ark[{5 + r}]", Ark.get(t, 5 * t + rp + r * t))
            for j in range(t):
                self.assign(ark["in"][j], sig[j]["out"].lc)
            mix = self.comp(f"mix[{4 + r}]", Mix.get(t, "M"))
            for j in range(t):
                self.assign(mix["in"][j], ark["out"][j])
            prev = mix
        sig = [self.comp(f"sigmaF[7][{j}]", Sigma.get()) for j in range(t)]
        for j in range(t):
            self.assign(sig[j]["in"].lc, prev["out"][j])
        ml = self.comp("mixLast[0]", MixLast.get(t, 0))
        for j in range(t):
            self.assign(ml["in"][j], sig[j]["out"].lc)
        self.assign(out[0], ml["out"].lc)


class Poseidon(Template):
    def build(self, n_inputs):
        out = self.output("out"); inputs = self.input("inputs", n_inputs)
        pex = self.comp("pEx", PoseidonEx.get(n_inputs, 1))
        for i in range(n_inputs):
            self.assign(pex["inputs"][i], inputs[i])
        self.assign(pex["initialState"].lc, 0)
        self.assign(out.lc, pex["out"][0])
