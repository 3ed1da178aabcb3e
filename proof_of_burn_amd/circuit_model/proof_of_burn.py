"""circuits/proof_of_burn.circom and the gadgets only it uses (shift, concat, substring_check, burn_address, proof_of_work,
rlp/*) of the reference as circuit-model templates (citations: file:line of the reference)."""
from __future__ import annotations

from .circuits import POSEIDON_PREFIX, PublicCommitment
from .core import P, Template
from .keccak import KeccakBytes
from .lib import IsEqual, IsZero, LessEqThan, LessThan, MultiAND, Mux1, Poseidon
from .utils import (AssertBits, AssertByteString, AssertGreaterEqThan, AssertLessEqThan, AssertLessThan, Bytes2Nibbles, Divide, Filter, Fit,
                    LittleEndianBytes2Num, Nibbles2Bytes, Num2BigEndianBytes, Selector, SelectorArray1D)


# ------------------------------------------------------------------------------------------------ shift.circom / concat.circom
class ShiftLeft(Template):             # shift.circom:17-36
    def build(self, n):
        inp = self.input("in", n); count = self.input("count"); out = self.output("out", n)
        is_eq = self.signal("isEq", n, n); temp = self.signal("temp", n, n)
        a = self.comp("AssertLessEqThan_22", AssertLessEqThan.get(16)); self.assign(a["a"].lc, count.lc); self.assign(a["b"].lc, n)
        for i in range(n):
            acc = 0
            for j in range(n):
                e = self.comp(f"IsEqual_30[{i}][{j}]", IsEqual.get())
                self.assign(e["in"][0], i); self.assign(e["in"][1], j - count.lc); self.assign(is_eq[i, j], e["out"].lc)
                self.mul(temp[i, j], is_eq[i, j], inp[j])
                acc = temp[i, j] + acc
            self.assign(out[i], acc)


class ShiftRight(Template):            # shift.circom:51-76
    def build(self, n, max_shift):
        inp = self.input("in", n); count = self.input("count"); out = self.output("out", n + max_shift)
        is_eq = self.signal("isEq", max_shift + 1); temps = self.signal("temps", max_shift + 1, n)
        a = self.comp("AssertLessEqThan_56", AssertLessEqThan.get(16)); self.assign(a["a"].lc, count.lc); self.assign(a["b"].lc, max_shift)
        acc = [0] * (n + max_shift)
        for i in range(max_shift + 1):
            e = self.comp(f"IsEqual_66[{i}]", IsEqual.get())
            self.assign(e["in"][0], i); self.assign(e["in"][1], count.lc); self.assign(is_eq[i], e["out"].lc)
            for j in range(n):
                self.mul(temps[i, j], is_eq[i], inp[j])
                acc[i + j] = temps[i, j] + acc[i + j]
        for i in range(n + max_shift):
            self.assign(out[i], acc[i])


class Mask(Template):                  # concat.circom:18-30
    def build(self, n):
        inp = self.input("in", n); count = self.input("count"); out = self.output("out", n); flt = self.signal("filter", n)
        f = self.comp("Filter_24", Filter.get(n)); self.assign(f["in"].lc, count.lc); self.copy(flt, f["out"])
        for i in range(n):
            self.mul(out[i], inp[i], flt[i])


class Concat(Template):                # concat.circom:47-84
    def build(self, la, lb):
        a = self.input("a", la); a_len = self.input("aLen"); b = self.input("b", lb); b_len = self.input("bLen")
        out = self.output("out", la + lb); out_len = self.output("outLen")
        ma = self.signal("maskedA", la); mb = self.signal("maskedB", lb); sb = self.signal("shiftedB", la + lb)
        x = self.comp("AssertLessEqThan_57", AssertLessEqThan.get(16)); self.assign(x["a"].lc, a_len.lc); self.assign(x["b"].lc, la)
        y = self.comp("AssertLessEqThan_58", AssertLessEqThan.get(16)); self.assign(y["a"].lc, b_len.lc); self.assign(y["b"].lc, lb)
        m1 = self.comp("Mask_65", Mask.get(la)); self.copy(m1["in"], a); self.assign(m1["count"].lc, a_len.lc); self.copy(ma, m1["out"])
        m2 = self.comp("Mask_68", Mask.get(lb)); self.copy(m2["in"], b); self.assign(m2["count"].lc, b_len.lc); self.copy(mb, m2["out"])
        sr = self.comp("ShiftRight_71", ShiftRight.get(lb, la)); self.copy(sr["in"], mb); self.assign(sr["count"].lc, a_len.lc); self.copy(sb, sr["out"])
        for i in range(la + lb):
            self.assign(out[i], ma[i] + sb[i] if i < la else sb[i])
        self.assign(out_len.lc, a_len.lc + b_len.lc)


# ------------------------------------------------------------------------------------------------ substring_check.circom
class SubstringCheck(Template):        # :24-100
    def build(self, mm, sl):
        assert sl <= 31
        k = mm - sl + 1
        main = self.input("mainInput", mm); main_len = self.input("mainLen"); sub = self.input("subInput", sl); out = self.output("out")
        num = self.signal("subInputNum"); M = self.signal("M", mm + 1); exists = self.signal("exists", k); last = self.signal("isLastIndex", k)
        allowed = self.signal("allowed", k + 1); sums = self.signal("sums", k + 1); dne = self.signal("doesNotExist")
        a = self.comp("AssertByteString_33", AssertByteString.get(sl)); self.copy(a["in"], sub)
        b = self.comp("AssertByteString_34", AssertByteString.get(mm)); self.copy(b["in"], main)
        c = self.comp("AssertLessEqThan_36", AssertLessEqThan.get(16)); self.assign(c["a"].lc, main_len.lc); self.assign(c["b"].lc, mm)
        d = self.comp("AssertLessEqThan_37", AssertLessEqThan.get(16)); self.assign(d["a"].lc, sl); self.assign(d["b"].lc, main_len.lc)
        le = self.comp("LittleEndianBytes2Num_40", LittleEndianBytes2Num.get(sl)); self.copy(le["in"], sub); self.assign(num.lc, le["out"].lc)
        self.assign(M[0], 0)
        for i in range(mm):
            self.assign(M[i + 1], main[i] * pow(256, i, P) + M[i])
        self.assign(allowed[0], 1); self.assign(sums[0], 0)
        for i in range(k):
            e1 = self.comp(f"IsEqual_87[{i}]", IsEqual.get())
            self.assign(e1["in"][0], i); self.assign(e1["in"][1], main_len.lc - sl + 1); self.assign(last[i], e1["out"].lc)
            self.mul(allowed[i + 1], allowed[i], 1 - last[i])
            e2 = self.comp(f"IsEqual_91[{i}]", IsEqual.get())
            self.assign(e2["in"][0], num.lc * pow(256, i, P)); self.assign(e2["in"][1], M[i + sl] - M[i]); self.assign(exists[i], e2["out"].lc)
            self.mul(sums[i + 1], allowed[i + 1], exists[i], sums[i])
        z = self.comp("IsZero_98", IsZero.get()); self.assign(z["in"].lc, sums[k]); self.assign(dne.lc, z["out"].lc)
        self.assign(out.lc, 1 - dne.lc)


# ------------------------------------------------------------------------------------------------ rlp/integer.circom
class CountBytes(Template):            # :16-49
    def build(self, n):
        by = self.input("bytes", n); ln = self.output("len"); iz = self.signal("isZero", n); sz = self.signal("stillZero", n)
        for i in range(n):
            z = self.comp(f"IsZero_27[{i}]", IsZero.get()); self.assign(z["in"].lc, by[i]); self.assign(iz[i], z["out"].lc)
        lead = 0
        for i in range(n):
            if i == 0:
                self.assign(sz[i], iz[i])
            else:
                self.mul(sz[i], iz[i], sz[i - 1])
            lead = sz[i] + lead
        self.assign(ln.lc, n - lead)


class RlpInteger(Template):            # :67-110
    def build(self, n):
        assert n <= 31
        inp = self.input("in"); out = self.output("out", n + 1); out_len = self.output("outLen")
        by = self.signal("bytes", n); length = self.signal("length"); be = self.signal("bigEndian", n)
        single = self.signal("isSingleByte"); is_zero = self.signal("isZero"); first = self.signal("firstRlpByte")
        c = self.comp("Num2BigEndianBytes_83", Num2BigEndianBytes.get(n)); self.assign(c["in"].lc, inp.lc); self.copy(by, c["out"])
        cb = self.comp("CountBytes_84", CountBytes.get(n)); self.copy(cb["bytes"], by); self.assign(length.lc, cb["len"].lc)
        sl = self.comp("ShiftLeft_85", ShiftLeft.get(n)); self.copy(sl["in"], by); self.assign(sl["count"].lc, n - length.lc); self.copy(be, sl["out"])
        lt = self.comp("LessThan_91", LessThan.get(n * 8)); self.assign(lt["in"][0], inp.lc); self.assign(lt["in"][1], 128); self.assign(single.lc, lt["out"].lc)
        z = self.comp("IsZero_92", IsZero.get()); self.assign(z["in"].lc, inp.lc); self.assign(is_zero.lc, z["out"].lc)
        m = self.comp("Mux1_93", Mux1.get()); self.assign(m["c"][0], 0x80 + length.lc); self.assign(m["c"][1], inp.lc); self.assign(m["s"].lc, single.lc)
        self.assign(first.lc, m["out"].lc)
        self.assign(out[0], first.lc + is_zero.lc * 0x80)
        for i in range(1, n + 1):
            self.mul(out[i], 1 - single.lc, be[i - 1])
        self.assign(out_len.lc, (1 - single.lc) + length.lc + is_zero.lc)


# ------------------------------------------------------------------------------------------------ rlp/empty_account.circom
EMPTY_TAIL = [160, 86, 232, 31, 23, 27, 204, 85, 166, 255, 131, 69, 230, 146, 192, 248, 110, 91, 72, 224, 27, 153, 108, 173, 192, 1, 98, 47, 181, 227, 99, 180, 33,
              160, 197, 210, 70, 1, 134, 247, 35, 60, 146, 126, 125, 178, 220, 199, 3, 192, 229, 0, 182, 83, 202, 130, 39, 59, 123, 250, 216, 4, 93, 133, 164, 112]


class RlpEmptyAccount(Template):       # :20-134
    def build(self, mb):
        assert mb <= 31
        balance = self.input("balance"); out = self.output("out", 4 + mb + 66); out_len = self.output("outLen")
        pn = self.signal("prefixedNonceAndBalanceRlp", 4 + mb); pnl = self.signal("prefixedNonceAndBalanceRlpLen")
        br = self.signal("balanceRlp", mb + 1); brl = self.signal("balanceRlpLen"); nbl = self.signal("nonceAndBalanceRlpLen"); sc = self.signal("storageAndCodeHashRlp", 66)
        self.assign(pn[2], 0x80)
        ri = self.comp("RlpInteger_41", RlpInteger.get(mb)); self.assign(ri["in"].lc, balance.lc); self.copy(br, ri["out"]); self.assign(brl.lc, ri["outLen"].lc)
        for i in range(mb + 1):
            self.assign(pn[i + 3], br[i])
        self.assign(nbl.lc, 1 + brl.lc)
        self.assign(pnl.lc, 2 + nbl.lc)
        for i, v in enumerate(EMPTY_TAIL):
            self.assign(sc[i], v)
        self.assign(pn[0], 0xf7 + 1)
        self.assign(pn[1], nbl.lc + 66)
        cc = self.comp("concat", Concat.get(4 + mb, 66))
        self.copy(cc["a"], pn); self.assign(cc["aLen"].lc, pnl.lc); self.copy(cc["b"], sc); self.assign(cc["bLen"].lc, 66)
        self.copy(out, cc["out"]); self.assign(out_len.lc, cc["outLen"].lc)


# ------------------------------------------------------------------------------------------------ rlp/merkle_patricia_trie_leaf.circom
class TruncatedAddressHash(Template):  # :50-90
    def build(self, b):
        n2 = 2 * b
        nib = self.input("addressHashNibbles", n2); nib_len = self.input("addressHashNibblesLen"); out = self.output("out", b + 1); out_len = self.output("outLen")
        div = self.signal("div"); rem = self.signal("rem"); shifted = self.signal("shifted", n2); on = self.signal("outNibbles", n2 + 2)
        self.signal("temp", n2 - 1)                                   # declared, never assigned (:76): unconstrained wires
        a = self.comp("AssertLessEqThan_57", AssertLessEqThan.get(7)); self.assign(a["a"].lc, nib_len.lc); self.assign(a["b"].lc, n2)
        d = self.comp("Divide_60", Divide.get(7)); self.assign(d["a"].lc, nib_len.lc); self.assign(d["b"].lc, 2)
        self.assign(div.lc, d["out"].lc); self.assign(rem.lc, d["rem"].lc)
        sl = self.comp("ShiftLeft_64", ShiftLeft.get(n2)); self.copy(sl["in"], nib); self.assign(sl["count"].lc, n2 - nib_len.lc); self.copy(shifted, sl["out"])
        self.assign(on[0], 2 + rem.lc)
        self.mul(on[1], rem.lc, shifted[0])
        for i in range(n2):
            if i < n2 - 1:
                m = self.comp(f"Mux1_81[{i}]", Mux1.get())
                self.assign(m["c"][0], shifted[i]); self.assign(m["c"][1], shifted[i + 1]); self.assign(m["s"].lc, rem.lc); self.assign(on[i + 2], m["out"].lc)
            else:
                self.mul(on[i + 2], 1 - rem.lc, shifted[i])
        nb = self.comp("Nibbles2Bytes_88", Nibbles2Bytes.get(b + 1)); self.copy(nb["nibbles"], on); self.copy(out, nb["bytes"])
        self.assign(out_len.lc, 1 + div.lc)


class RlpMerklePatriciaTrieLeaf(Template):  # :102-189
    def build(self, ab, bb):
        max_acc = 4 + bb + 66; max_val = 2 + max_acc; max_key = 1 + ab; max_pk = 2 + 1 + max_key; max_out = max_pk + max_val
        nib = self.input("addressHashNibbles", 2 * ab); nib_len = self.input("addressHashNibblesLen"); balance = self.input("balance")
        out = self.output("out", max_out); out_len = self.output("outLen")
        key = self.signal("key", max_key); key_len = self.signal("keyLen"); acc = self.signal("rlpEmptyAccount", max_acc); acc_len = self.signal("rlpEmptyAccountLen")
        pk = self.signal("prefixedKeyRlp", max_pk); pk_len = self.signal("prefixedKeyRlpLen"); val = self.signal("valueRlp", max_val); val_len = self.signal("valueRlpLen")
        t = self.comp("TruncatedAddressHash_148", TruncatedAddressHash.get(ab))
        self.copy(t["addressHashNibbles"], nib); self.assign(t["addressHashNibblesLen"].lc, nib_len.lc); self.copy(key, t["out"]); self.assign(key_len.lc, t["outLen"].lc)
        g = self.comp("AssertGreaterEqThan_151", AssertGreaterEqThan.get(16)); self.assign(g["a"].lc, key_len.lc); self.assign(g["b"].lc, 2)
        e = self.comp("RlpEmptyAccount_155", RlpEmptyAccount.get(bb)); self.assign(e["balance"].lc, balance.lc); self.copy(acc, e["out"]); self.assign(acc_len.lc, e["outLen"].lc)
        self.assign(val[0], 0xb7 + 1); self.assign(val[1], acc_len.lc)
        for i in range(max_acc):
            self.assign(val[i + 2], acc[i])
        self.assign(val_len.lc, 2 + acc_len.lc)
        self.assign(pk[0], 0xf7 + 1); self.assign(pk[1], (key_len.lc + 1) + val_len.lc); self.assign(pk[2], 0x80 + key_len.lc)
        for i in range(max_key):
            self.assign(pk[i + 3], key[i])
        self.assign(pk_len.lc, 3 + key_len.lc)
        c = self.comp("Concat_183", Concat.get(max_pk, max_val))
        self.copy(c["a"], pk); self.assign(c["aLen"].lc, pk_len.lc); self.copy(c["b"], val); self.assign(c["bLen"].lc, val_len.lc)
        self.copy(out, c["out"]); self.assign(out_len.lc, c["outLen"].lc)


class IsInRange(Template):             # :196-207
    def build(self, nbits):
        lo = self.input("lower"); v = self.input("value"); hi = self.input("upper"); out = self.output("out")
        a = self.signal("lowerLteValue"); b = self.signal("valueLteUpper")
        for line, s in ((201, lo), (202, v), (203, hi)):
            x = self.comp(f"AssertBits_{line}", AssertBits.get(nbits)); self.assign(x["in"].lc, s.lc)
        l1 = self.comp("LessEqThan_204", LessEqThan.get(nbits)); self.assign(l1["in"][0], lo.lc); self.assign(l1["in"][1], v.lc); self.assign(a.lc, l1["out"].lc)
        l2 = self.comp("LessEqThan_205", LessEqThan.get(nbits)); self.assign(l2["in"][0], v.lc); self.assign(l2["in"][1], hi.lc); self.assign(b.lc, l2["out"].lc)
        self.mul(out.lc, a.lc, b.lc)


class LeafDetector(Template):          # :247-294
    def build(self, n):
        layer = self.input("layer", n); layer_len = self.input("layerLen"); is_leaf = self.output("isLeaf")
        s = {name: self.signal(name) for name in (
            "leafPrefixIsF8", "totalLength", "isConsistentWithLayerLen", "keyPrefix", "keyPrefixIsValid", "keyIsMultiByte", "keyExtraLen", "keyLen",
            "valueWrapperPrefix", "valueWrapperPrefixIsB8", "valueWrapperLen", "valuePrefix", "valuePrefixIsF8", "valueLen", "isValueWrapperLenConsistent",
            "isKeyValueLenEqualWithLayerLen")}
        a = self.comp("AssertLessEqThan_252", AssertLessEqThan.get(16)); self.assign(a["a"].lc, layer_len.lc); self.assign(a["b"].lc, n)

        def iseq(line, x, y, dst):
            e = self.comp(f"IsEqual_{line}", IsEqual.get()); self.assign(e["in"][0], x); self.assign(e["in"][1], y); self.assign(s[dst].lc, e["out"].lc)

        def select(line, idx, dst):
            c = self.comp(f"Selector_{line}", Selector.get(n)); self.copy(c["vals"], layer); self.assign(c["select"].lc, idx); self.assign(s[dst].lc, c["out"].lc)

        iseq(255, layer[0], 0xf8, "leafPrefixIsF8")
        self.assign(s["totalLength"].lc, layer[1])
        iseq(257, s["totalLength"].lc + 2, layer_len.lc, "isConsistentWithLayerLen")
        self.assign(s["keyPrefix"].lc, layer[2])
        le = self.comp("LessEqThan_261", LessEqThan.get(16)); self.assign(le["in"][0], s["keyPrefix"].lc); self.assign(le["in"][1], 0xb7)
        self.assign(s["keyPrefixIsValid"].lc, le["out"].lc)
        r = self.comp("IsInRange_264", IsInRange.get(16))
        self.assign(r["lower"].lc, 0x81); self.assign(r["value"].lc, s["keyPrefix"].lc); self.assign(r["upper"].lc, 0xb7); self.assign(s["keyIsMultiByte"].lc, r["out"].lc)
        self.mul(s["keyExtraLen"].lc, s["keyIsMultiByte"].lc, s["keyPrefix"].lc - 0x80)
        kl = s["keyLen"].lc
        self.assign(kl, 1 + s["keyExtraLen"].lc)
        select(279, 2 + kl + 0, "valueWrapperPrefix")
        iseq(280, s["valueWrapperPrefix"].lc, 0xb8, "valueWrapperPrefixIsB8")
        select(281, 2 + kl + 1, "valueWrapperLen")
        select(283, 2 + kl + 2, "valuePrefix")
        iseq(284, s["valuePrefix"].lc, 0xf8, "valuePrefixIsF8")
        select(285, 2 + kl + 3, "valueLen")
        iseq(286, s["valueWrapperLen"].lc, s["valueLen"].lc + 2, "isValueWrapperLenConsistent")
        iseq(287, kl + s["valueLen"].lc + 6, layer_len.lc, "isKeyValueLenEqualWithLayerLen")
        m = self.comp("MultiAND_289", MultiAND.get(7))
        for i, name in enumerate(("leafPrefixIsF8", "isConsistentWithLayerLen", "keyPrefixIsValid", "valueWrapperPrefixIsB8", "isValueWrapperLenConsistent",
                                  "valuePrefixIsF8", "isKeyValueLenEqualWithLayerLen")):
            self.assign(m["in"][i], s[name].lc)
        self.assign(is_leaf.lc, m["out"].lc)


# ------------------------------------------------------------------------------------------------ burn_address.circom / proof_of_work.circom
class BurnAddress(Template):           # burn_address.circom:47-58
    def build(self):
        bk = self.input("burnKey"); ra = self.input("revealAmount"); bec = self.input("burnExtraCommitment"); out = self.output("addressBytes", 20)
        h = self.signal("hash"); hb = self.signal("hashBytes", 32)
        p = self.comp("Poseidon_55", Poseidon.get(4))
        self.assign(p["inputs"][0], POSEIDON_PREFIX + 0); self.assign(p["inputs"][1], bk.lc); self.assign(p["inputs"][2], ra.lc); self.assign(p["inputs"][3], bec.lc)
        self.assign(h.lc, p["out"].lc)
        c = self.comp("Num2BigEndianBytes_56", Num2BigEndianBytes.get(32)); self.assign(c["in"].lc, h.lc); self.copy(hb, c["out"])
        f = self.comp("Fit_57", Fit.get(32, 20)); self.copy(f["in"], hb); self.copy(out, f["out"])


class BurnAddressHash(Template):       # burn_address.circom:67-83
    def build(self):
        bk = self.input("burnKey"); ra = self.input("revealAmount"); bec = self.input("burnExtraCommitment"); out = self.output("addressHashNibbles", 64)
        ab = self.signal("addressBytes", 20); blk = self.signal("addressBytesBlock", 136); ah = self.signal("addressHash", 32)
        b = self.comp("BurnAddress_74", BurnAddress.get())
        self.assign(b["burnKey"].lc, bk.lc); self.assign(b["revealAmount"].lc, ra.lc); self.assign(b["burnExtraCommitment"].lc, bec.lc); self.copy(ab, b["addressBytes"])
        f = self.comp("Fit_78", Fit.get(20, 136)); self.copy(f["in"], ab); self.copy(blk, f["out"])
        k = self.comp("KeccakBytes_79", KeccakBytes.get(1)); self.copy(k["in"], blk); self.assign(k["inLen"].lc, 20); self.copy(ah, k["out"])
        n = self.comp("Bytes2Nibbles_82", Bytes2Nibbles.get(32)); self.copy(n["in"], ah); self.copy(out, n["out"])


class EIP7503(Template):               # proof_of_work.circom:11-21
    def build(self):
        out = self.output("out", 8)
        for i, ch in enumerate(b"EIP-7503"):
            self.assign(out[i], ch)


class ConcatFixed4(Template):          # proof_of_work.circom:28-48
    def build(self, A, B, C, D):
        a = self.input("a", A); b = self.input("b", B); c = self.input("c", C); d = self.input("d", D); out = self.output("out", A + B + C + D)
        pos = 0
        for sig, n in ((a, A), (b, B), (c, C), (d, D)):
            for i in range(n):
                self.assign(out[pos + i], sig[i])
            pos += n


class ProofOfWorkChecker(Template):    # proof_of_work.circom:54-81
    def build(self):
        bk = self.input("burnKey"); ra = self.input("revealAmount"); bec = self.input("burnExtraCommitment"); mzb = self.input("minimumZeroBytes")
        kb = self.signal("burnKeyBytes", 32); rb = self.signal("revealAmountBytes", 32); eb = self.signal("burnExtraCommitmentBytes", 32); eip = self.signal("eip7503", 8)
        hin = self.signal("hasherInput", 104); blk = self.signal("burnKeyBlock", 136); kk = self.signal("burnKeyKeccak", 32); sbz = self.signal("shouldBeZero", 32)
        for line, src, dst in ((60, bk, kb), (61, ra, rb), (62, bec, eb)):
            c = self.comp(f"Num2BigEndianBytes_{line}", Num2BigEndianBytes.get(32)); self.assign(c["in"].lc, src.lc); self.copy(dst, c["out"])
        e = self.comp("EIP7503_63", EIP7503.get()); self.copy(eip, e["out"])
        cf = self.comp("ConcatFixed4_66", ConcatFixed4.get(32, 32, 32, 8))
        self.copy(cf["a"], kb); self.copy(cf["b"], rb); self.copy(cf["c"], eb); self.copy(cf["d"], eip); self.copy(hin, cf["out"])
        f = self.comp("Fit_70", Fit.get(104, 136)); self.copy(f["in"], hin); self.copy(blk, f["out"])
        k = self.comp("KeccakBytes_71", KeccakBytes.get(1)); self.copy(k["in"], blk); self.assign(k["inLen"].lc, 104); self.copy(kk, k["out"])
        fl = self.comp("Filter_73", Filter.get(32)); self.assign(fl["in"].lc, mzb.lc); self.copy(sbz, fl["out"])
        for i in range(32):
            self.constrain(kk[i], sbz[i], 0)


# ------------------------------------------------------------------------------------------------ proof_of_burn.circom
class ProofOfBurn(Template):           # :34-212
    def build(self, L, NB, HB, min_nib, amount_bytes, pow_zero, max_intended, max_actual):
        assert amount_bytes <= 31
        LB, HBy, AB8 = NB * 136, HB * 136, amount_bytes * 8
        commitment = self.output("commitment")
        burn_key = self.input("burnKey"); actual = self.input("actualBalance"); intended = self.input("intendedBalance"); reveal = self.input("revealAmount")
        bec = self.input("burnExtraCommitment"); nln = self.input("numLeafAddressNibbles"); layers = self.input("layers", L, LB); layer_lens = self.input("layerLens", L)
        num_layers = self.input("numLayers"); header = self.input("blockHeader", HBy); header_len = self.input("blockHeaderLen"); relax = self.input("byteSecurityRelax")
        pec = self.input("_proofExtraCommitment")
        rem_coin = self.signal("remainingCoin"); nullifier = self.signal("nullifier"); ahn = self.signal("addressHashNibbles", 64); block_root = self.signal("blockRoot", 32)
        state_root = self.signal("stateRoot", 32); nul_b = self.signal("nullifierBytes", 32); rc_b = self.signal("remainingCoinBytes", 32); ra_b = self.signal("revealAmountBytes", 32)
        bec_b = self.signal("burnExtraCommitmentBytes", 32); ec_b = self.signal("extraCommitmentBytes", 32); last_layer = self.signal("lastLayer", LB)
        last_len = self.signal("lastLayerLen"); layer_exists = self.signal("layerExists", L); sub_chk = self.signal("substringCheckers", L - 1)
        layer_k = self.signal("layerKeccaks", L, 32); red_k = self.signal("reducedLayerKeccaks", L, 31); is_leaf = self.signal("isLeaf", L)
        is_last_leaf = self.signal("isLastLayerLeaf"); leaf = self.signal("leaf", 139); leaf_len = self.signal("leafLen")

        def two(tpl, name, x, y):
            c = self.comp(name, tpl); self.assign(c["a"].lc, x); self.assign(c["b"].lc, y)

        two(AssertLessEqThan.get(AB8), "AssertLessEqThan_84", intended.lc, max_intended)
        two(AssertLessEqThan.get(AB8), "AssertLessEqThan_85", actual.lc, max_actual)
        two(AssertLessEqThan.get(AB8), "AssertLessEqThan_86", intended.lc, actual.lc)
        two(AssertLessEqThan.get(16), "AssertLessEqThan_90", relax.lc * 2, min_nib)
        two(AssertGreaterEqThan.get(16), "AssertGreaterEqThan_91", nln.lc, min_nib - relax.lc * 2)
        ab = self.comp("AssertBits_96", AssertBits.get(AB8)); self.assign(ab["in"].lc, reveal.lc)
        two(AssertLessEqThan.get(AB8), "AssertLessEqThan_97", reveal.lc, intended.lc)
        for i in range(L):
            two(AssertLessThan.get(16), f"AssertLessThan_101[{i}]", layer_lens[i], LB * 8)
            a = self.comp(f"AssertByteString_102[{i}]", AssertByteString.get(LB))
            for k in range(LB):
                self.assign(a["in"][k], layers[i, k])
        two(AssertLessThan.get(16), "AssertLessThan_105", header_len.lc, HBy * 8)
        a = self.comp("AssertByteString_106", AssertByteString.get(HBy)); self.copy(a["in"], header)
        p3 = self.comp("Poseidon_113", Poseidon.get(3))
        self.assign(p3["inputs"][0], POSEIDON_PREFIX + 2); self.assign(p3["inputs"][1], burn_key.lc); self.assign(p3["inputs"][2], intended.lc - reveal.lc)
        self.assign(rem_coin.lc, p3["out"].lc)
        p2 = self.comp("Poseidon_116", Poseidon.get(2))
        self.assign(p2["inputs"][0], POSEIDON_PREFIX + 1); self.assign(p2["inputs"][1], burn_key.lc); self.assign(nullifier.lc, p2["out"].lc)
        bah = self.comp("BurnAddressHash_119", BurnAddressHash.get())
        self.assign(bah["burnKey"].lc, burn_key.lc); self.assign(bah["revealAmount"].lc, reveal.lc); self.assign(bah["burnExtraCommitment"].lc, bec.lc)
        self.copy(ahn, bah["addressHashNibbles"])
        kh = self.comp("KeccakBytes_122", KeccakBytes.get(HB)); self.copy(kh["in"], header); self.assign(kh["inLen"].lc, header_len.lc); self.copy(block_root, kh["out"])
        for i in range(32):
            self.assign(state_root[i], header[91 + i])
        for line, src, dst in ((132, nullifier, nul_b), (133, rem_coin, rc_b), (134, reveal, ra_b), (135, bec, bec_b), (136, pec, ec_b)):
            c = self.comp(f"Num2BigEndianBytes_{line}", Num2BigEndianBytes.get(32)); self.assign(c["in"].lc, src.lc); self.copy(dst, c["out"])
        pc = self.comp("PublicCommitment_137", PublicCommitment.get(6))
        for j, sig in enumerate((block_root, nul_b, rc_b, ra_b, bec_b, ec_b)):
            for k in range(32):
                self.assign(pc["in"][j, k], sig[k])
        self.assign(commitment.lc, pc["out"].lc)
        sa = self.comp("SelectorArray1D_142", SelectorArray1D.get(L, LB)); self.copy(sa["arrays"], layers); self.assign(sa["select"].lc, num_layers.lc - 1)
        self.copy(last_layer, sa["out"])
        sl = self.comp("Selector_146", Selector.get(L)); self.copy(sl["vals"], layer_lens); self.assign(sl["select"].lc, num_layers.lc - 1); self.assign(last_len.lc, sl["out"].lc)
        fl = self.comp("Filter_150", Filter.get(L)); self.assign(fl["in"].lc, num_layers.lc); self.copy(layer_exists, fl["out"])
        n_leaves = 0
        for i in range(L):
            ld = self.comp(f"LeafDetector_159[{i}]", LeafDetector.get(LB))
            for k in range(LB):
                self.assign(ld["layer"][k], layers[i, k])
            self.assign(ld["layerLen"].lc, layer_lens[i]); self.assign(is_leaf[i], ld["isLeaf"].lc)
            n_leaves = is_leaf[i] + n_leaves
            kb = self.comp(f"KeccakBytes_163[{i}]", KeccakBytes.get(NB))
            for k in range(LB):
                self.assign(kb["in"][k], layers[i, k])
            self.assign(kb["inLen"].lc, layer_lens[i])
            for k in range(32):
                self.assign(layer_k[i, k], kb["out"][k])
            ft = self.comp(f"Fit_166[{i}]", Fit.get(32, 31))
            for k in range(32):
                self.assign(ft["in"][k], layer_k[i, k])
            for k in range(31):
                self.assign(red_k[i, k], ft["out"][k])
            if i > 0:
                sc = self.comp(f"SubstringCheck_170[{i}]", SubstringCheck.get(LB, 31))
                for k in range(31):
                    self.assign(sc["subInput"][k], red_k[i, k])
                self.assign(sc["mainLen"].lc, layer_lens[i - 1])
                for k in range(LB):
                    self.assign(sc["mainInput"][k], layers[i - 1, k])
                self.assign(sub_chk[i - 1], sc["out"].lc)
                self.constrain(1 - sub_chk[i - 1], layer_exists[i], 0)
        self.eq(n_leaves, 1)
        ld = self.comp("LeafDetector_187", LeafDetector.get(LB)); self.copy(ld["layer"], last_layer); self.assign(ld["layerLen"].lc, last_len.lc)
        self.assign(is_last_leaf.lc, ld["isLeaf"].lc)
        self.eq(is_last_leaf.lc, 1)
        for i in range(32):
            self.eq(layer_k[0, i], state_root[i])
        rl = self.comp("RlpMerklePatriciaTrieLeaf_198", RlpMerklePatriciaTrieLeaf.get(32, amount_bytes))
        self.copy(rl["addressHashNibbles"], ahn); self.assign(rl["addressHashNibblesLen"].lc, nln.lc); self.assign(rl["balance"].lc, actual.lc)
        self.copy(leaf, rl["out"]); self.assign(leaf_len.lc, rl["outLen"].lc)
        for i in range(139):
            self.eq(leaf[i], last_layer[i])
        self.eq(leaf_len.lc, last_len.lc)
        pw = self.comp("ProofOfWorkChecker_211", ProofOfWorkChecker.get())
        self.assign(pw["burnKey"].lc, burn_key.lc); self.assign(pw["revealAmount"].lc, reveal.lc); self.assign(pw["burnExtraCommitment"].lc, bec.lc)
        self.assign(pw["minimumZeroBytes"].lc, pow_zero + relax.lc)
