"""ctypes front-end of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

`run_main(main, inputs)` plays the role of tests/test.py:57-74 of the reference for one case:
the template-instantiation string is what the reference writes after `component main =`
(tests/test.py:31), `inputs` is the input.json dict; the return value is the list of output
signals (witness[1..nOut], tests/test.py:40-47) or None when any assert/=== failed
(tests/test.py:65-68).
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# declaration order of `signal input`s per template (the order the O0 witness stores them in)
INPUT_ORDER = {
    "ProofOfBurn": ["burnKey", "actualBalance", "intendedBalance", "revealAmount", "burnExtraCommitment",
                    "numLeafAddressNibbles", "layers", "layerLens", "numLayers", "blockHeader", "blockHeaderLen",
                    "byteSecurityRelax", "_proofExtraCommitment"],            # proof_of_burn.circom:43-72
    "Spend": ["burnKey", "balance", "withdrawnBalance", "extraCommitment"],     # spend.circom:33-36
    "EIP7503": [],
    "ConcatFixed4": ["a", "b", "c", "d"],
    "ProofOfWorkChecker": ["burnKey", "revealAmount", "burnExtraCommitment", "minimumZeroBytes"],
    "PublicCommitment": ["in"],
    "Poseidon": ["inputs"],
    "Divide": ["a", "b"],
    "SubstringCheck": ["mainInput", "mainLen", "subInput"],
    "ShiftLeft": ["in", "count"], "ShiftRight": ["in", "count"], "Mask": ["in", "count"],
    "Concat": ["a", "aLen", "b", "bLen"],
    "Selector": ["vals", "select"], "SelectorArray1D": ["arrays", "select"], "SelectorArray2D": ["arrays", "select"],
    "BigEndianBytes2Num": ["in"], "LittleEndianBytes2Num": ["in"], "Bytes2Nibbles": ["in"], "Nibbles2Bytes": ["nibbles"],
    "Num2BigEndianBytes": ["in"], "Num2LittleEndianBytes": ["in"], "Num2BitsSafe": ["in"],
    "Pad": ["in", "inLen"], "KeccakBytes": ["in", "inLen"],
    "BurnAddress": ["burnKey", "revealAmount", "burnExtraCommitment"],
    "BurnAddressHash": ["burnKey", "revealAmount", "burnExtraCommitment"],
    "AssertBits": ["in"], "AssertByteString": ["in"],
    "AssertLessEqThan": ["a", "b"], "AssertLessThan": ["a", "b"], "AssertGreaterEqThan": ["a", "b"],
    "Filter": ["in"], "Fit": ["in"], "Reverse": ["in"], "Flatten": ["in"], "Reshape": ["in"],
    "RlpInteger": ["in"], "CountBytes": ["bytes"], "RlpEmptyAccount": ["balance"],
    "TruncatedAddressHash": ["addressHashNibbles", "addressHashNibblesLen"],
    "IsInRange": ["lower", "value", "upper"],
    "LeafDetector": ["layer", "layerLen"],
    "RlpMerklePatriciaTrieLeaf": ["addressHashNibbles", "addressHashNibblesLen", "balance"],
    "Keccakf": ["in"],
}


def parse_main(main: str):
    """'ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)' -> ('ProofOfBurn', [4,4,5,20,31,2,10**18,10**19])"""
    m = re.fullmatch(r"\s*(\w+)\s*\((.*)\)\s*;?\s*", main, flags=re.S)
    if not m:
        raise ValueError(main)
    args = m.group(2).strip()
    params = []
    if args:
        for a in args.split(","):
            if not re.fullmatch(r"[\d\s\*\+\-\(\)/x0-9a-fA-F]+", a):
                raise ValueError(a)
            params.append(int(eval(a, {"__builtins__": {}})))  # circom constant expressions: ints, **, *, +, -
    return m.group(1), params


def to_field(v) -> int:
    """input.json value -> Fr: JSON ints or decimal / 0x strings, reduced mod p."""
    if isinstance(v, bool):
        return int(v)
    if isinstance(v, int):
        return v % P
    if isinstance(v, str):
        return int(v, 16 if v.lower().startswith("0x") else 10) % P
    raise TypeError(f"unsupported input value {v!r}")


def flatten(v, out):
    if isinstance(v, (list, tuple)):
        for x in v:
            flatten(x, out)
    else:
        out.append(to_field(v))


def limbs(vals):
    arr = (ctypes.c_uint64 * (4 * len(vals)))()
    for i, v in enumerate(vals):
        for k in range(4):
            arr[4 * i + k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return arr


class OracleResult(ctypes.Structure):
    _fields_ = [("witness", ctypes.POINTER(ctypes.c_uint64)), ("nwitness", ctypes.c_uint64), ("noutputs", ctypes.c_uint64),
                ("ninputs_expected", ctypes.c_uint64), ("failed", ctypes.c_int32), ("unknown", ctypes.c_int32),
                ("msg", ctypes.c_char * 160)]


_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("pob_oracle.c", "fr.h", "poseidon_consts.h")]
    if os.environ.get("ORACLE_SAN") == "1":      # AddressSanitizer + UBSan build of the oracle (tools/run_sanitizers.py: clang's runtime is preloaded)
        so = os.path.join(ORACLE_DIR, "liboracle_san.so")
        if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang", "-O1", "-g", "-fPIC", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                   "-fno-omit-frame-pointer", "-shared-libasan", "-shared", "-o", so, os.path.join(ORACLE_DIR, "pob_oracle.c")])
        return so
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_run.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ctypes.c_uint64,
                                    ctypes.POINTER(OracleResult)]
        _lib.oracle_run.restype = ctypes.c_int
        _lib.oracle_wtns_size.restype = ctypes.c_uint64
        _lib.oracle_wtns_into.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        _lib.oracle_wtns_write.argtypes = [ctypes.c_char_p]
        _lib.oracle_keccak256.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]
        _lib.oracle_fail_sites.restype = ctypes.c_char_p
    return _lib


class Run:
    """Result of one oracle execution; the witness memory stays valid until the next run()."""

    def __init__(self, res: OracleResult, rc: int):
        self.rc = rc
        self.failed = bool(res.failed) or rc != 0
        self.msg = res.msg.decode(errors="replace")
        self.nwitness = int(res.nwitness)
        self.noutputs = int(res.noutputs)
        self._res = res

    def value(self, i: int) -> int:
        w = self._res.witness
        return w[4 * i] | (w[4 * i + 1] << 64) | (w[4 * i + 2] << 128) | (w[4 * i + 3] << 192)

    def outputs(self):
        return [self.value(1 + i) for i in range(self.noutputs)]

    def witness_bytes(self) -> bytes:
        return ctypes.string_at(self._res.witness, 32 * self.nwitness)

    def witness_numpy(self):
        import numpy as np
        buf = (ctypes.c_uint8 * (32 * self.nwitness)).from_address(ctypes.addressof(self._res.witness.contents))
        return np.frombuffer(buf, dtype=np.uint8)

    def wtns_numpy(self):
        import numpy as np
        n = lib().oracle_wtns_size()
        out = np.empty(n, dtype=np.uint8)
        assert lib().oracle_wtns_into(out.ctypes.data, n) == 0
        return out


def run(main: str, inputs: dict, capacity: int = 0) -> Run:
    name, params = parse_main(main)
    flat = []
    given = dict(inputs)
    for key in INPUT_ORDER[name]:
        if key not in given:
            raise KeyError(f"missing input {key}")
        flatten(given.pop(key), flat)
    if given:
        raise KeyError(f"unexpected inputs {sorted(given)}")
    if capacity == 0:
        capacity = 1 << 22
        if name == "ProofOfBurn":
            capacity = 2_600_000 * (params[0] * params[1] + params[2] + 5) + (4 << 20)
        elif name in ("Spend", "KeccakBytes", "PublicCommitment", "BurnAddressHash", "ProofOfWorkChecker", "Keccakf"):
            capacity = 2_700_000 * 4
    res = OracleResult()
    pl = limbs([p % (1 << 256) for p in params])
    il = limbs(flat)
    rc = lib().oracle_run(name.encode(), pl, len(params), il, len(flat), capacity, ctypes.byref(res))
    if rc == -2:
        raise NotImplementedError(res.msg.decode())
    return Run(res, rc)


def fail_sites():
    """EVERY failing assert / === site of the last run as (template, line) pairs, in execution order (the oracle does not stop at the first)"""
    txt = lib().oracle_fail_sites().decode()
    return tuple((a, int(b)) for a, b in (x.split(":") for x in txt.split(";") if x))


def run_main(main: str, inputs: dict):
    r = run(main, inputs)
    return None if r.failed else r.outputs()


def keccak256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().oracle_keccak256(bytes(data), len(data), out)
    return out.raw


# ---------------------------------------------------------------------------- many oracle runs at once (the GPU tests: ~2 s and 6.9 GB per production witness, one thread each)
def payload_digest(arr) -> str:
    """128-bit digest of a canonical payload (numpy uint8 / bytes-like): what two processes compare instead of shipping gigabytes to each other"""
    import xxhash
    h = xxhash.xxh3_128()
    mv = memoryview(arr).cast("B")
    step = 1 << 26
    for i in range(0, len(mv), step):
        h.update(mv[i:i + step])
    return h.hexdigest()


class Brief:
    """what a worker process hands back of one oracle run: verdict, message, outputs, wire count, digest of the payload (valid witnesses, on request)"""
    __slots__ = ("failed", "msg", "outs", "nwitness", "digest", "sites")

    def __init__(self, failed, msg, outs, nwitness, digest, sites=()):
        self.failed, self.msg, self.outs, self.nwitness, self.digest, self.sites = failed, msg, outs, nwitness, digest, sites

    def outputs(self):
        return self.outs


def _brief_worker(args):
    main, inputs, want_digest = args
    r = run(main, inputs)
    d = payload_digest(r.witness_numpy()) if (want_digest and not r.failed) else None
    out = (r.failed, r.msg, None if r.failed else r.outputs(), r.nwitness, d, fail_sites())
    lib().oracle_free()
    return out


class OraclePool:
    """worker PROCESSES (spawned: the parent holds a HIP context; the oracle keeps one global witness, so threads cannot share it) that run the oracle beside whatever
    the test does on the GPU: `jobs = pool.submit(main, inputs, digest=True)` returns at once, `pool.collect(jobs)` gives one Brief per input, in order"""

    def __init__(self, procs: int | None = None):
        import multiprocessing as mp
        import os
        if procs is None:
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                quota = os.cpu_count() if q == "max" else max(1, int(q) // int(per))
            except Exception:
                quota = os.cpu_count() or 2
            procs = max(2, min(8, quota - 1))
        build()                                   # (once, in the parent: the workers only load the library)
        self.pool = mp.get_context("spawn").Pool(procs)

    def submit(self, main: str, inputs, digest: bool = False):
        return [self.pool.apply_async(_brief_worker, ((main, inp, digest),)) for inp in inputs]

    def collect(self, jobs, timeout: float = 900.0):
        return [Brief(*j.get(timeout)) for j in jobs]

    def close(self):
        self.pool.terminate(); self.pool.join()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
