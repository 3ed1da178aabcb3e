"""The two circuits with a `component main` in the reference (circuits/spend.circom, circuits/proof_of_burn.circom) and the
top-level gadgets they use (public_commitment, burn_address, proof_of_work), as circuit-model templates."""
from __future__ import annotations

from .core import Circuit, Template
from .keccak import KeccakBytes
from .lib import Poseidon
from .utils import AssertByteString, AssertGreaterEqThan, BigEndianBytes2Num, Fit, Flatten, Num2BigEndianBytes

POSEIDON_PREFIX = 5265656504298861414514317065875120428884240036965045859626767452974705356670    # constants.circom:3-5


class PublicCommitment(Template):      # public_commitment.circom:18-42
    def build(self, n):
        inp = self.input("in", n, 32); out = self.output("out")
        nb = n * 32 // 136 + (1 if (n * 32) % 136 else 0)
        assert nb * 136 - n * 32 >= 1
        flat = self.signal("flattenIn", n * 32); block = self.signal("block", nb * 136); hsh = self.signal("hash", 32); red = self.signal("reducedHash", 31)
        for i in range(n):
            a = self.comp(f"AssertByteString_24[{i}]", AssertByteString.get(32))
            for k in range(32):
                self.assign(a["in"][k], inp[i, k])
        fl = self.comp("Flatten_34", Flatten.get(n, 32)); self.copy(fl["in"], inp); self.copy(flat, fl["out"])
        ft = self.comp("Fit_35", Fit.get(n * 32, nb * 136)); self.copy(ft["in"], flat); self.copy(block, ft["out"])
        kb = self.comp("KeccakBytes_36", KeccakBytes.get(nb)); self.copy(kb["in"], block); self.assign(kb["inLen"].lc, n * 32); self.copy(hsh, kb["out"])
        f2 = self.comp("Fit_40", Fit.get(32, 31)); self.copy(f2["in"], hsh); self.copy(red, f2["out"])
        be = self.comp("BigEndianBytes2Num_41", BigEndianBytes2Num.get(31)); self.copy(be["in"], red); self.assign(out.lc, be["out"].lc)


class Spend(Template):                 # spend.circom:32-53
    def build(self, max_amount_bytes):
        assert max_amount_bytes <= 31
        burn_key = self.input("burnKey"); balance = self.input("balance"); wd = self.input("withdrawnBalance"); extra = self.input("extraCommitment")
        commitment = self.output("commitment")
        coin = self.signal("coin"); rem = self.signal("remainingCoin")
        coin_b = self.signal("coinBytes", 32); wd_b = self.signal("withdrawnBalanceBytes", 32); rem_b = self.signal("remainingCoinBytes", 32)
        ex_b = self.signal("extraCommmitmentBytes", 32)          # (sic: the reference's spelling, spend.circom:49)
        ge = self.comp("AssertGreaterEqThan_41", AssertGreaterEqThan.get(max_amount_bytes * 8))
        self.assign(ge["a"].lc, balance.lc); self.assign(ge["b"].lc, wd.lc)
        p1 = self.comp("Poseidon_43", Poseidon.get(3))
        self.assign(p1["inputs"][0], POSEIDON_PREFIX + 2); self.assign(p1["inputs"][1], burn_key.lc); self.assign(p1["inputs"][2], balance.lc)
        self.assign(coin.lc, p1["out"].lc)
        p2 = self.comp("Poseidon_44", Poseidon.get(3))
        self.assign(p2["inputs"][0], POSEIDON_PREFIX + 2); self.assign(p2["inputs"][1], burn_key.lc); self.assign(p2["inputs"][2], balance.lc - wd.lc)
        self.assign(rem.lc, p2["out"].lc)
        for line, src, dst in ((46, coin, coin_b), (47, wd, wd_b), (48, rem, rem_b), (49, extra, ex_b)):
            c = self.comp(f"Num2BigEndianBytes_{line}", Num2BigEndianBytes.get(32)); self.assign(c["in"].lc, src.lc); self.copy(dst, c["out"])
        pc = self.comp("PublicCommitment_50", PublicCommitment.get(4))
        for j, sig in enumerate((coin_b, wd_b, rem_b, ex_b)):
            for k in range(32):
                self.assign(pc["in"][j, k], sig[k])
        self.assign(commitment.lc, pc["out"].lc)


def circuit(main: str) -> Circuit:
    """'Spend(31)' / 'ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)' -> Circuit (memoised templates)"""
    from ..witness import parse_main
    name, params = parse_main(main)
    if name == "Spend":
        return Circuit(Spend.get(*params))
    if name == "ProofOfBurn":
        from .proof_of_burn import ProofOfBurn
        return Circuit(ProofOfBurn.get(*params))
    raise NotImplementedError(name)
