"""Stored O1-style keep maps (which O0 wires survive, circuit_model/o1.py) of the instantiations that matter: deriving the map of the
production circuit from the model takes ~3 minutes and ~12 GB (216 M wires through scipy's connected components), but the result is
highly regular (the same Keccak-f pattern 84 times): delta-encoded and xz-compressed it is 14 KB.

    python -m proof_of_burn_amd.circuit_model keepmap "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"      # (re)generate

File format: b"POBKEEP1" | u64 n_wires (W) | u64 n_keep | sha256 of the uint32 LE array (32 B) | xz(delta-encoded uint32 LE array).
Which representative circom's own --O1 keeps is NOT pinned (no circom here); the map is data."""
from __future__ import annotations

import hashlib
import lzma
import os
import re
import struct

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
MAGIC = b"POBKEEP1"


def path_of(main: str) -> str:
    return os.path.join(DATA, "o1_keep_" + re.sub(r"[^0-9A-Za-z]+", "_", main.replace(" ", "").replace("**", "e")).strip("_") + ".xz")


def save(main: str, keep: np.ndarray, n_wires: int) -> str:
    keep = np.ascontiguousarray(keep, dtype=np.uint32)
    delta = np.diff(keep, prepend=np.uint32(0)).astype(np.uint32)
    os.makedirs(DATA, exist_ok=True)
    p = path_of(main)
    with open(p, "wb") as f:
        f.write(MAGIC + struct.pack("<QQ", n_wires, keep.size) + hashlib.sha256(keep.tobytes()).digest() + lzma.compress(delta.tobytes(), preset=9))
    return p


def load(main: str):
    """-> (keep uint32[n_keep] sorted, n_wires); FileNotFoundError if no map is stored for this instantiation"""
    with open(path_of(main), "rb") as f:
        raw = f.read()
    if raw[:8] != MAGIC:
        raise ValueError("not a keep map")
    n_wires, n_keep = struct.unpack("<QQ", raw[8:24])
    keep = np.cumsum(np.frombuffer(lzma.decompress(raw[56:]), dtype=np.uint32), dtype=np.uint64).astype(np.uint32)
    if keep.size != n_keep or hashlib.sha256(keep.tobytes()).digest() != raw[24:56]:
        raise ValueError("keep map is corrupt")
    return keep, int(n_wires)


def generate(main: str) -> str:
    from . import circuit
    from .o1 import reduce_map
    c = circuit(main)
    m = reduce_map(c)
    return save(main, m.keep, c.n_wires)
