"""Pure-Python stand-ins for the third-party packages the reference's tests import.

TEST INFRASTRUCTURE ONLY.  The reference's known-answer tests
(/root/reference/tests/testcases/*.py) compute their expectations with `web3`,
`rlp` and `eth_abi`, none of which is installed here (no network).  This module
restates the three public specifications they rely on:

  * Keccak-256 (original Keccak padding 0x01, *not* NIST SHA-3's 0x06) --
    what `web3.Web3.keccak` returns (reference call sites:
    tests/testcases/keccak.py:59-60, public_commitment.py:10-12, burn_address.py:19-23).
  * RLP encoding of ints / byte strings / lists (Ethereum yellow paper app. B) --
    `rlp.encode` (tests/testcases/rlp/empty_account.py:5-19).
  * `eth_abi.packed.encode_packed(["uint256"]*n, vals)` = concatenated 32-byte
    big-endian words (tests/testcases/public_commitment.py:11).

The Keccak-f[1600] permutation is cross-checked against hashlib.sha3_256 (same
permutation, different domain byte) in tests/test_refshim.py, and against the one
Keccak-256 constant the reference hard-codes (block root at
tests/testcases/proof_of_burn.py:22).
"""

from __future__ import annotations

import sys
import types

_MASK = (1 << 64) - 1

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]

# rho offsets indexed [x][y], FIPS-202 table 2
_RHO = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]


def _rotl(v: int, n: int) -> int:
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _MASK if n else v


def keccak_f1600(a: list[int]) -> list[int]:
    """Keccak-f[1600] on 25 lanes, lane index x + 5*y."""
    a = list(a)
    for rnd in range(24):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ _rotl(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = _rotl(a[x + 5 * y], _RHO[x][y])
        a = [b[i] ^ ((~b[(i % 5 + 1) % 5 + 5 * (i // 5)]) & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) for i in range(25)]
        a = [v & _MASK for v in a]
        a[0] ^= _RC[rnd]
    return a


def _sponge(data: bytes, rate: int, dom: int, outlen: int) -> bytes:
    p = bytearray(data)
    p.append(dom)
    while len(p) % rate:
        p.append(0)
    p[-1] |= 0x80
    st = [0] * 25
    for off in range(0, len(p), rate):
        blk = p[off:off + rate]
        for i in range(rate // 8):
            st[i] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        st = keccak_f1600(st)
    out = b"".join(v.to_bytes(8, "little") for v in st)
    return out[:outlen]


def keccak256(data: bytes) -> bytes:
    return _sponge(bytes(data), 136, 0x01, 32)


def sha3_256(data: bytes) -> bytes:
    """NIST variant, only used to cross-check the permutation against hashlib."""
    return _sponge(bytes(data), 136, 0x06, 32)


# --------------------------------------------------------------------------- RLP

def _int_to_min_bytes(v: int) -> bytes:
    return b"" if v == 0 else v.to_bytes((v.bit_length() + 7) // 8, "big")


def _rlp_len_prefix(n: int, short_base: int) -> bytes:
    if n <= 55:
        return bytes([short_base + n])
    ln = _int_to_min_bytes(n)
    return bytes([short_base + 55 + len(ln)]) + ln


def rlp_encode(item) -> bytes:
    if isinstance(item, int):
        item = _int_to_min_bytes(item)
    if isinstance(item, (bytes, bytearray)):
        item = bytes(item)
        if len(item) == 1 and item[0] < 0x80:
            return item
        return _rlp_len_prefix(len(item), 0x80) + item
    if isinstance(item, (list, tuple)):
        body = b"".join(rlp_encode(x) for x in item)
        return _rlp_len_prefix(len(body), 0xC0) + body
    raise TypeError(type(item))


def rlp_decode(data: bytes):
    """Minimal decoder (single item) -- used by the synthetic-input self checks."""
    def dec(b, pos):
        p = b[pos]
        if p < 0x80:
            return bytes([p]), pos + 1
        if p <= 0xB7:
            n = p - 0x80
            return bytes(b[pos + 1:pos + 1 + n]), pos + 1 + n
        if p <= 0xBF:
            ll = p - 0xB7
            n = int.from_bytes(b[pos + 1:pos + 1 + ll], "big")
            return bytes(b[pos + 1 + ll:pos + 1 + ll + n]), pos + 1 + ll + n
        if p <= 0xF7:
            n = p - 0xC0
            start = pos + 1
        else:
            ll = p - 0xF7
            n = int.from_bytes(b[pos + 1:pos + 1 + ll], "big")
            start = pos + 1 + ll
        out, q = [], start
        while q < start + n:
            it, q = dec(b, q)
            out.append(it)
        return out, start + n
    item, end = dec(bytes(data), 0)
    assert end == len(data)
    return item


def encode_packed(types_, vals) -> bytes:
    assert all(t == "uint256" for t in types_)
    return b"".join(int(v).to_bytes(32, "big") for v in vals)


# ------------------------------------------------------------- sys.modules shims

def install_shims() -> None:
    """Register fake `web3`, `rlp`, `eth_abi` modules exposing exactly the calls the
    reference testcases make, so they can be imported unmodified."""

    class _Web3:
        @staticmethod
        def keccak(primitive=None, hexstr=None, text=None):
            if hexstr is not None:
                primitive = bytes.fromhex(hexstr[2:] if hexstr.startswith("0x") else hexstr)
            if text is not None:
                primitive = text.encode()
            return keccak256(bytes(primitive))

        @staticmethod
        def to_bytes(primitive=None, hexstr=None):
            if hexstr is not None:
                return bytes.fromhex(hexstr[2:] if hexstr.startswith("0x") else hexstr)
            if isinstance(primitive, int):
                return _int_to_min_bytes(primitive) or b"\x00"
            return bytes(primitive)

    web3 = types.ModuleType("web3")
    web3.Web3 = _Web3
    rlp = types.ModuleType("rlp")
    rlp.encode = rlp_encode
    rlp.decode = rlp_decode
    eth_abi = types.ModuleType("eth_abi")
    packed = types.ModuleType("eth_abi.packed")
    packed.encode_packed = encode_packed
    eth_abi.packed = packed
    sys.modules.setdefault("web3", web3)
    sys.modules.setdefault("rlp", rlp)
    sys.modules.setdefault("eth_abi", eth_abi)
    sys.modules.setdefault("eth_abi.packed", packed)
