"""Synthetic, VALID proof_of_burn inputs at any trie depth -- the host input producer of the path
(reference tests/main.py:47-178, which needs a live JSON-RPC chain; here the account proof is constructed).

Recipe: per distinct burn key: draw amounts within the circuit's bounds (proof_of_burn.circom:84-97),
search the proof-of-work key natively (keccak(BE32(key)|BE32(reveal)|BE32(extra)|"EIP-7503") with 2+relax leading
zero bytes, tests/main.py:47-56), derive the burn address = Poseidon4(...)[:20] (burn_address.circom:55-57) and its
Keccak; per witness: build a depth-D Merkle-Patricia path (D-1 branch nodes with random siblings, then the leaf
RLP([hex-prefix(key), RLP([0, balance, EMPTY_STORAGE, EMPTY_CODE])])), and a block header whose bytes 91..122 are the
state root (proof_of_burn.circom:125-129).  Expected commitments are computed with the reference's formula
(tests/testcases/proof_of_burn.py:18-36, public_commitment.py:5-17).
"""
from __future__ import annotations

import ctypes
import os
import random
from dataclasses import dataclass, field

from .witness import P, keccak256, load_library

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POSEIDON_PREFIX = 5265656504298861414514317065875120428884240036965045859626767452974705356670   # constants.circom:3-5
EMPTY_STORAGE = bytes.fromhex("56e81f171bcc55a6ff8345e692c0f86e5b48e01b996cadc001622fb5e363b421")  # empty_account.circom:9
EMPTY_CODE = bytes.fromhex("c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470")     # empty_account.circom:10
MAIN = (16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)   # circuits/main_proof_of_burn.circom:27


def poseidon(*inputs: int) -> int:
    """host-side Poseidon (circomlib parameters) for the input producer: burn address, nullifier, remaining coin"""
    from . import poseidon_host
    return poseidon_host.poseidon(list(inputs))


def _min_bytes(v: int) -> bytes:
    return b"" if v == 0 else v.to_bytes((v.bit_length() + 7) // 8, "big")


def rlp(item) -> bytes:
    if isinstance(item, int):
        item = _min_bytes(item)
    if isinstance(item, (bytes, bytearray)):
        item = bytes(item)
        if len(item) == 1 and item[0] < 0x80:
            return item
        return _len_prefix(len(item), 0x80) + item
    body = b"".join(rlp(x) for x in item)
    return _len_prefix(len(body), 0xC0) + body


def _len_prefix(n: int, base: int) -> bytes:
    if n <= 55:
        return bytes([base + n])
    ln = _min_bytes(n)
    return bytes([base + 55 + len(ln)]) + ln


def hex_prefix_leaf(nibbles) -> bytes:
    """leaf key encoding: even -> 0x20 | pairs, odd -> 0x3n | pairs (merkle_patricia_trie_leaf.circom:15-48)"""
    if len(nibbles) % 2 == 0:
        out = [0x20]
        rest = nibbles
    else:
        out = [0x30 + nibbles[0]]
        rest = nibbles[1:]
    out += [rest[i] * 16 + rest[i + 1] for i in range(0, len(rest), 2)]
    return bytes(out)


def pow_search(start_key: int, reveal: int, extra: int, zero_bytes: int, max_tries: int = 1 << 32, device: int | None = None) -> int:
    """first burn key >= start_key whose keccak(key | reveal | extra | "EIP-7503") starts with zero_bytes zero bytes
    (reference tests/main.py:47-56); device = GPU ordinal runs the search as a HIP kernel (same result)"""
    postfix = reveal.to_bytes(32, "big") + extra.to_bytes(32, "big") + b"EIP-7503"
    out = ctypes.create_string_buffer(32)
    lib = load_library()
    if device is None:
        tries = lib.pob_pow_search(start_key.to_bytes(32, "big"), postfix, len(postfix), zero_bytes, max_tries, out)
    else:
        tries = lib.pob_pow_search_gpu(device, start_key.to_bytes(32, "big"), postfix, len(postfix), zero_bytes, max_tries, out)
    if tries < 0:
        raise RuntimeError("proof-of-work search exhausted")
    return int.from_bytes(out.raw, "big")


@dataclass
class BurnKey:
    key: int
    reveal: int
    extra: int
    intended: int
    actual: int
    address: bytes
    address_hash: bytes
    nullifier: int
    remaining_coin: int
    relax: int


def make_burn_key(rng: random.Random, relax: int, pow_zero: int = 2, max_intended: int = 10 ** 19, max_actual: int = 10 ** 20,
                  pow_device: int | None = None) -> BurnKey:
    extra = rng.randrange(P)
    intended = rng.randrange(1, max_intended + 1)
    reveal = rng.randrange(0, intended + 1)
    actual = intended + rng.randrange(0, min(max_actual - intended, 10 ** 17) + 1)
    key = pow_search(rng.randrange(P - (1 << 64)), reveal, extra, pow_zero + relax, device=pow_device)
    addr = poseidon(POSEIDON_PREFIX + 0, key, reveal, extra).to_bytes(32, "big")[:20]
    return BurnKey(key, reveal, extra, intended, actual, addr, keccak256(addr),
                   poseidon(POSEIDON_PREFIX + 1, key), poseidon(POSEIDON_PREFIX + 2, key, intended - reveal), relax)


def account_proof(rng: random.Random, bk: BurnKey, depth: int):
    """[root branch, ..., leaf] such that keccak(node[i+1]) appears in node[i]; returns (layers, leaf nibble count)"""
    nibbles = [b for byte in bk.address_hash for b in (byte >> 4, byte & 15)]
    rest = nibbles[depth - 1:]
    account = rlp([0, bk.actual, EMPTY_STORAGE, EMPTY_CODE])
    leaf = rlp([hex_prefix_leaf(rest), account])
    layers = [leaf]
    child = leaf
    for k in range(depth - 2, -1, -1):
        slots = [b""] * 17
        others = [i for i in range(16) if i != nibbles[k]]
        rng.shuffle(others)
        for i in others[:rng.randrange(8, 16)]:          # >= 9 hashed children: payload >= 256 B => node starts with 0xf9
            slots[i] = rng.randbytes(32)
        slots[nibbles[k]] = keccak256(child)
        child = rlp(slots)
        assert child[0] == 0xF9 and 31 <= len(child) <= 543
        layers.append(child)
    layers.reverse()
    return layers, len(rest)


def block_header(rng: random.Random, state_root: bytes) -> bytes:
    fields = [rng.randbytes(32), rng.randbytes(32), rng.randbytes(20), state_root, rng.randbytes(32), rng.randbytes(32),
              rng.randbytes(256), 0, rng.randrange(1 << 24), 30_000_000, rng.randrange(30_000_000), 1_750_000_000 + rng.randrange(1 << 20),
              rng.randbytes(rng.randrange(0, 32)), rng.randbytes(32), rng.randbytes(8), rng.randrange(1, 1 << 36), rng.randbytes(32),
              rng.randrange(1 << 20), rng.randrange(1 << 20), rng.randbytes(32)]
    hdr = rlp(fields)
    assert hdr[91:123] == state_root
    return hdr


def expected_commitment(vals) -> int:
    """keccak(abi.encodePacked(uint256...)) >> 8 (public_commitment.circom:38-41; tests/testcases/public_commitment.py:5-17)"""
    return int.from_bytes(keccak256(b"".join(int(v).to_bytes(32, "big") for v in vals))[:31], "big")


@dataclass
class Batch:
    inputs: list = field(default_factory=list)
    commitments: list = field(default_factory=list)
    distinct_keys: int = 0
    depth: int = 0


def synthetic_batch(n: int, depth: int = 10, seed: int = 0xB0B, distinct_keys: int = 8, params=MAIN, first: int = 0,
                    pow_device: int | None = None) -> Batch:
    """witnesses [first, first + n) of the global synthetic batch of this seed: witness g depends only on (seed, g), so a rank's
    slice is the same whoever generates it (the multi-GPU tests compare a 2-rank run with a single-rank run of the same seeds).
    pow_device: run the proof-of-work searches as the HIP kernel on that GPU (3-zero-byte searches of the 16-layer shape)."""
    L, NB, HB, min_nib, amount_bytes, pow_zero, max_intended, max_actual = params
    LB, HBy = 136 * NB, 136 * HB
    assert 2 <= depth <= L
    leaf_nibbles = 64 - (depth - 1)
    relax = 0 if leaf_nibbles >= min_nib else (min_nib - leaf_nibbles + 1) // 2
    nkeys = max(1, distinct_keys)
    used = sorted({(first + w) % nkeys for w in range(n)})
    keys = {k: make_burn_key(random.Random(seed * 1_000_003 + k), relax, pow_zero, min(max_intended, 256 ** amount_bytes - 1), max_actual, pow_device)
            for k in used}
    out = Batch(distinct_keys=len(keys), depth=depth)
    for w in range(first, first + n):
        rng = random.Random(seed + w)
        bk = keys[w % nkeys]
        layers, nn = account_proof(rng, bk, depth)
        hdr = block_header(rng, keccak256(layers[0]))
        assert len(hdr) < HBy and all(len(x) < LB for x in layers) and nn == leaf_nibbles
        proof_extra = rng.randrange(P)
        inp = {
            "burnKey": str(bk.key), "actualBalance": str(bk.actual), "intendedBalance": str(bk.intended), "revealAmount": str(bk.reveal),
            "burnExtraCommitment": str(bk.extra), "numLeafAddressNibbles": str(nn),
            "layers": [list(x) + [0] * (LB - len(x)) for x in layers] + [[0] * LB] * (L - depth),
            "layerLens": [len(x) for x in layers] + [256] * (L - depth),        # tests/main.py:148-150 pads with 256
            "numLayers": depth, "blockHeader": list(hdr) + [0] * (HBy - len(hdr)), "blockHeaderLen": len(hdr),
            "byteSecurityRelax": relax, "_proofExtraCommitment": str(proof_extra),
        }
        out.inputs.append(inp)
        out.commitments.append(expected_commitment([int.from_bytes(keccak256(hdr), "big"), bk.nullifier, bk.remaining_coin, bk.reveal, bk.extra, proof_extra]))
    return out


def synthetic_spend_batch(n: int, seed: int = 0x5BE4D, first: int = 0, amount_bytes: int = 31) -> Batch:
    """witnesses [first, first + n) of a synthetic Spend(amount_bytes) batch (spend.circom:32-53): witness g depends only on (seed, g); commitment =
    PublicCommitment([coin, withdrawnBalance, remainingCoin, extraCommitment]) with coin = Poseidon(COIN_PREFIX, burnKey, balance)"""
    out = Batch(distinct_keys=n, depth=0)
    for w in range(first, first + n):
        rng = random.Random(seed * 7919 + w)
        key, balance = rng.randrange(P), rng.randrange(1, 256 ** min(amount_bytes, 12))
        withdrawn, extra = rng.randrange(balance + 1), rng.randrange(P)
        coin, rem = poseidon(POSEIDON_PREFIX + 2, key, balance), poseidon(POSEIDON_PREFIX + 2, key, balance - withdrawn)
        out.inputs.append({"burnKey": str(key), "balance": str(balance), "withdrawnBalance": str(withdrawn), "extraCommitment": str(extra)})
        out.commitments.append(expected_commitment([coin, withdrawn, rem, extra]))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Real account proof -> input.json (reference tests/main.py:65-178, which obtains the proof from a JSON-RPC node)
def rlp_decode(data: bytes):
    """minimal RLP decoder: bytes -> nested lists of bytes (yellow paper app. B; the reference uses the `rlp` package).  Every declared
    length is checked against the buffer: a truncated or over-long encoding raises ValueError, nothing is silently cut by slicing"""
    data = bytes(data)

    def take(pos, n, what):
        if pos + n > len(data):
            raise ValueError(f"RLP: {what} of {n} bytes at offset {pos} overruns the {len(data)}-byte buffer")
        return data[pos:pos + n]

    def item(pos):
        if pos >= len(data):
            raise ValueError("RLP: unexpected end of input")
        b0 = data[pos]
        if b0 < 0x80:
            return data[pos:pos + 1], pos + 1
        if b0 < 0xB8:
            n = b0 - 0x80
            return take(pos + 1, n, "string"), pos + 1 + n
        if b0 < 0xC0:
            ll = b0 - 0xB7
            n = int.from_bytes(take(pos + 1, ll, "length"), "big")
            return take(pos + 1 + ll, n, "string"), pos + 1 + ll + n
        if b0 < 0xF8:
            n, start = b0 - 0xC0, pos + 1
        else:
            ll = b0 - 0xF7
            n, start = int.from_bytes(take(pos + 1, ll, "length"), "big"), pos + 1 + ll
        take(start, n, "list payload")
        out, p = [], start
        while p < start + n:
            x, p = item(p)
            out.append(x)
        if p != start + n:
            raise ValueError("RLP: list payload overruns")
        return out, start + n
    v, end = item(0)
    if end != len(data):
        raise ValueError("RLP: trailing bytes")
    return v


def leaf_address_nibbles(leaf_node: bytes) -> int:
    """number of address-hash nibbles stored in the leaf of an account proof = `numLeafAddressNibbles`
    (reference tests/main.py:69-82: hex-prefix flag 2 -> even, 3 -> odd path)"""
    node = rlp_decode(leaf_node)
    if not isinstance(node, list) or len(node) != 2 or not isinstance(node[0], bytes):
        raise ValueError("last proof node is not a leaf (a leaf is a 2-item list [hex-prefix key, value])")
    key = node[0]
    if len(key) == 0:
        raise ValueError("last proof node has an empty key")
    flag = key[0] >> 4
    if flag == 2:
        return 2 * (len(key) - 1)
    if flag == 3:
        return 2 * (len(key) - 1) + 1
    raise ValueError("last proof node is not a leaf (hex-prefix flag %d)" % flag)


def from_account_proof(account_proof, header_rlp: bytes, burn_key: int, actual_balance: int, intended_balance: int, reveal_amount: int,
                       burn_extra_commitment: int, proof_extra_commitment: int, byte_security_relax: int = 0, params=MAIN) -> dict:
    """eth_getProof's `accountProof` (list of RLP node byte strings, root first) + the RLP block header -> the circuit's input.json
    dict, as the reference producer builds it (tests/main.py:65-82 leaf / nibble count, :84-122 header, :134-150 zero padding of
    layers and header, unused layerLens = 256, :160-178 the dict)."""
    L, NB, HB = params[0], params[1], params[2]
    LB, HBy = 136 * NB, 136 * HB
    layers = [bytes(x) for x in account_proof]
    if not 1 <= len(layers) <= L:
        raise ValueError(f"{len(layers)} proof nodes, circuit takes 1..{L}")
    if any(len(x) >= LB for x in layers):
        raise ValueError(f"a proof node has {max(len(x) for x in layers)} bytes, circuit takes < {LB}")
    if len(header_rlp) >= HBy:
        raise ValueError(f"block header has {len(header_rlp)} bytes, circuit takes < {HBy}")
    hdr_fields = rlp_decode(bytes(header_rlp))
    if len(hdr_fields) < 4 or len(hdr_fields[3]) != 32 or bytes(header_rlp[91:123]) != hdr_fields[3]:
        raise ValueError("block header: the state root is not at bytes 91..122 (proof_of_burn.circom:125-129)")
    if keccak256(layers[0]) != hdr_fields[3]:
        raise ValueError("the first proof node does not hash to the header's state root")
    return {
        "burnKey": str(burn_key), "actualBalance": str(actual_balance), "intendedBalance": str(intended_balance), "revealAmount": str(reveal_amount),
        "burnExtraCommitment": str(burn_extra_commitment), "numLeafAddressNibbles": str(leaf_address_nibbles(layers[-1])),
        "layers": [list(x) + [0] * (LB - len(x)) for x in layers] + [[0] * LB] * (L - len(layers)),
        "layerLens": [len(x) for x in layers] + [256] * (L - len(layers)),
        "numLayers": len(layers), "blockHeader": list(header_rlp) + [0] * (HBy - len(header_rlp)), "blockHeaderLen": len(header_rlp),
        "byteSecurityRelax": byte_security_relax, "_proofExtraCommitment": str(proof_extra_commitment),
    }
