"""Host side of the MI355X witness generator: the reference's calculator interface over the C ABI.

Mirrors the reference boundary:
  * which circuit = the `component main = ...;` string (reference tests/test.py:31,
    circuits/main_proof_of_burn.circom:27, circuits/main_spend.circom:6);
  * inputs = the input.json dict with the circuit's `signal input` names
    (proof_of_burn.circom:43-72, spend.circom:33-36; producer tests/main.py:160-178), values as JSON ints or
    decimal / 0x strings, reduced mod p;
  * out = public output signals as ints, or failure (reference: any stderr output, tests/test.py:65-68),
    and an iden3 `.wtns` file (reference Makefile:4-5: `./main_proof_of_burn input.json witness.wtns`).

All arithmetic happens in libpob_hip.so (hand-written HIP, include/pob_hip.h).  There is NO CPU fallback:
if the library is missing or no GPU is visible, construction fails.
"""
from __future__ import annotations

import ctypes
import json
import os
import re
from dataclasses import dataclass
from typing import Iterable, Sequence

import numpy as np

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POB_LIB_PATH") or os.path.join(_HERE, "csrc", "libpob_hip.so")      # (POB_LIB_PATH: another BUILD of the same library, for A/B runs on one box)

POB_FR_INPUTS = ["burnKey", "actualBalance", "intendedBalance", "revealAmount", "burnExtraCommitment", "_proofExtraCommitment"]
POB_SM_INPUTS = ["numLeafAddressNibbles", "layers", "layerLens", "numLayers", "blockHeader", "blockHeaderLen", "byteSecurityRelax"]
SPEND_FR_INPUTS = ["burnKey", "balance", "withdrawnBalance", "extraCommitment"]

# failure-code template ids (csrc/policy.hpp)
_TPL = {1: "Num2Bits", 2: "IsZero", 3: "AliasCheck", 4: "AssertLessThan", 5: "AssertLessEqThan", 6: "AssertGreaterEqThan",
        7: "Divide", 8: "Selector", 9: "ProofOfWorkChecker", 10: "ProofOfBurn", 11: "input", 12: "CompConstant", 13: "misc"}
FAIL_INPUT_RANGE = (11 << 12) | 1

# Gadget-level mains (reference tests/test.py:146-201: `component main = T(params);` around one template of circuits/utils): the main's
# input signals in declaration order as (name, class, number of elements); class "f" = field-valued (packed as 32-byte canonical values),
# "s" = byte / length / selector signals the HIP gadgets compute in int32.  csrc/gadget_mains.hpp reads them in this order.
GADGET_INPUTS = {
    "EIP7503": lambda p: [],
    "ConcatFixed4": lambda p: [("a", "s", p[0]), ("b", "s", p[1]), ("c", "s", p[2]), ("d", "s", p[3])],
    "ProofOfWorkChecker": lambda p: [("burnKey", "f", 1), ("revealAmount", "f", 1), ("burnExtraCommitment", "f", 1), ("minimumZeroBytes", "s", 1)],
    "PublicCommitment": lambda p: [("in", "s", 32 * p[0])],
    "Poseidon": lambda p: [("inputs", "f", p[0])],
    "Divide": lambda p: [("a", "s", 1), ("b", "s", 1)],
    "SubstringCheck": lambda p: [("mainInput", "s", p[0]), ("mainLen", "s", 1), ("subInput", "s", p[1])],
    "ShiftLeft": lambda p: [("in", "s", p[0]), ("count", "s", 1)],
    "ShiftRight": lambda p: [("in", "s", p[0]), ("count", "s", 1)],
    "Mask": lambda p: [("in", "s", p[0]), ("count", "s", 1)],
    "Concat": lambda p: [("a", "s", p[0]), ("aLen", "s", 1), ("b", "s", p[1]), ("bLen", "s", 1)],
    "Selector": lambda p: [("vals", "s", p[0]), ("select", "s", 1)],
    "SelectorArray1D": lambda p: [("arrays", "s", p[0] * p[1]), ("select", "s", 1)],
    "SelectorArray2D": lambda p: [("arrays", "s", p[0] * p[1] * p[2]), ("select", "s", 1)],
    "BigEndianBytes2Num": lambda p: [("in", "s", p[0])],
    "LittleEndianBytes2Num": lambda p: [("in", "s", p[0])],
    "Bytes2Nibbles": lambda p: [("in", "s", p[0])],
    "Nibbles2Bytes": lambda p: [("nibbles", "s", 2 * p[0])],
    "Num2BigEndianBytes": lambda p: [("in", "f", 1)],
    "Num2LittleEndianBytes": lambda p: [("in", "f", 1)],
    "Num2BitsSafe": lambda p: [("in", "f", 1)],
    "Pad": lambda p: [("in", "s", p[0] * p[1]), ("inLen", "s", 1)],
    "KeccakBytes": lambda p: [("in", "s", 136 * p[0]), ("inLen", "s", 1)],
    "BurnAddress": lambda p: [("burnKey", "f", 1), ("revealAmount", "f", 1), ("burnExtraCommitment", "f", 1)],
    "BurnAddressHash": lambda p: [("burnKey", "f", 1), ("revealAmount", "f", 1), ("burnExtraCommitment", "f", 1)],
    "AssertBits": lambda p: [("in", "f", 1)],
    "AssertByteString": lambda p: [("in", "s", p[0])],
    "AssertLessEqThan": lambda p: [("a", "s", 1), ("b", "s", 1)],
    "AssertLessThan": lambda p: [("a", "s", 1), ("b", "s", 1)],
    "AssertGreaterEqThan": lambda p: [("a", "s", 1), ("b", "s", 1)],
    "Filter": lambda p: [("in", "s", 1)],
    "Fit": lambda p: [("in", "s", p[0])],
    "Reverse": lambda p: [("in", "s", p[0])],
    "Flatten": lambda p: [("in", "s", p[0] * p[1])],
    "Reshape": lambda p: [("in", "s", p[0] * p[1])],
    "RlpInteger": lambda p: [("in", "f", 1)],
    "CountBytes": lambda p: [("bytes", "s", p[0])],
    "RlpEmptyAccount": lambda p: [("balance", "f", 1)],
    "TruncatedAddressHash": lambda p: [("addressHashNibbles", "s", 2 * p[0]), ("addressHashNibblesLen", "s", 1)],
    "IsInRange": lambda p: [("lower", "s", 1), ("value", "s", 1), ("upper", "s", 1)],
    "LeafDetector": lambda p: [("layer", "s", p[0]), ("layerLen", "s", 1)],
    "RlpMerklePatriciaTrieLeaf": lambda p: [("addressHashNibbles", "s", 2 * p[0]), ("addressHashNibblesLen", "s", 1), ("balance", "f", 1)],
}


def circuit_of(name: str, params):
    """(circuit id, parameter list) for pob_plan_info / pob_open: the two production circuits, or a gadget-level main (template id first)"""
    if name == "ProofOfBurn":
        return 0, list(params)
    if name == "Spend":
        return 1, list(params)
    if name not in GADGET_INPUTS:
        raise NotImplementedError(f"{name} is not a template of the reference's circuits")
    npar = ctypes.c_int()
    tid = load_library().pob_gadget_template(name.encode(), ctypes.byref(npar))
    if tid < 0:
        raise NotImplementedError(f"{name}: unknown to libpob_hip.so")
    if npar.value != len(params):
        raise ValueError(f"{name} takes {npar.value} template parameters, got {len(params)}")
    return 2, [tid] + list(params)


def _limbs(params):
    arr = (ctypes.c_uint64 * (4 * max(len(params), 1)))()
    for i, v in enumerate(params):
        for k in range(4):
            arr[4 * i + k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return arr


class PobInfo(ctypes.Structure):
    _fields_ = [("n_witness", ctypes.c_uint64), ("n_bit", ctypes.c_uint64), ("n_sm", ctypes.c_uint64), ("n_fr", ctypes.c_uint64),
                ("n_fr_inputs", ctypes.c_uint32), ("n_sm_inputs", ctypes.c_uint32), ("n_outputs", ctypes.c_uint32),
                ("n_units", ctypes.c_uint32), ("n_sponges", ctypes.c_uint32), ("n_perms", ctypes.c_uint32),
                ("n_stages", ctypes.c_uint32), ("max_batch", ctypes.c_uint32),
                ("group_bytes", ctypes.c_uint64), ("keccak_bit_wires", ctypes.c_uint64), ("n_derived", ctypes.c_uint64), ("n_alias", ctypes.c_uint64),
                ("kchk_rounds", ctypes.c_uint32), ("kgc_rounds", ctypes.c_uint32)]


# one result record (include/pob_hip.h POB_RECORD_BYTES = 44)
RECORD_DTYPE = np.dtype([("status", "<u4"), ("check_status", "<u4"), ("bad_wire", "<u4"), ("commitment", "u1", (32,))])
NOT_EVALUATED, CLEAN = 0xFFFFFFFE, 0xFFFFFFFF
# the byte form of the small inputs (include/pob_hip.h pob_upload_inputs8): per witness POB_EXC_CAP exception slots {index, int32 value}, unused = POB_EXC_NONE
EXC_CAP, EXC_NONE, E_RANGE, E_NOMEM = 32, 0xFFFFFFFF, -6, -3
EXC_DTYPE = np.dtype([("k", "<u4"), ("v", "<i4")])

_lib = None


def load_library() -> ctypes.CDLL:
    """dlopen libpob_hip.so and declare the prototypes of include/pob_hip.h.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, u32p, u8p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint8)
    lib.pob_open.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(vp)]
    lib.pob_close.argtypes = [vp]
    lib.pob_close.restype = None
    lib.pob_get_info.argtypes = [vp, ctypes.POINTER(PobInfo)]
    lib.pob_plan_info.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.POINTER(PobInfo)]
    lib.pob_strerror.argtypes = [vp]
    lib.pob_strerror.restype = ctypes.c_char_p
    lib.pob_gadget_template.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    lib.pob_gadget_template.restype = ctypes.c_int
    lib.pob_upload_inputs.argtypes = [vp, vp, vp, ctypes.c_uint32]
    lib.pob_upload_inputs_async.argtypes = [vp, vp, vp, ctypes.c_uint32, vp]
    lib.pob_upload_inputs8.argtypes = [vp, vp, vp, vp, ctypes.c_uint32]
    lib.pob_upload_inputs8_async.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, vp]
    lib.pob_narrow_inputs.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
    lib.pob_pack_json_batch8.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint64),
                                         ctypes.c_uint32, ctypes.c_int, vp, vp, vp, vp, ctypes.c_char_p, ctypes.c_uint32]
    lib.pob_host_alloc.argtypes = [ctypes.POINTER(vp), ctypes.c_uint64]
    lib.pob_pack_json.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, vp, vp, vp, ctypes.c_char_p, ctypes.c_uint32]
    lib.pob_pack_json_batch.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.c_uint32, ctypes.c_int, vp, vp, vp, ctypes.c_char_p, ctypes.c_uint32]
    lib.pob_host_free.argtypes = [vp]
    lib.pob_host_free.restype = None
    lib.pob_results_fetch.argtypes = [vp]
    lib.pob_results_wait.argtypes = [vp, ctypes.POINTER(vp), u32p]
    lib.pob_emit_begin_reduced.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.c_uint64]
    lib.pob_write_wtns_reduced.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint64, ctypes.c_char_p]
    lib.pob_reduced_map_pin.argtypes = [vp, vp, ctypes.c_uint64]
    lib.pob_emit_measure_ex.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    lib.pob_generate.argtypes = [vp, vp]
    lib.pob_constraint_check.argtypes = [vp, vp]
    lib.pob_sync.argtypes = [vp]
    lib.pob_set_partner.argtypes = [vp, vp]
    lib.pob_results.argtypes = [vp, vp, vp, vp, vp]
    lib.pob_results_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    lib.pob_results_records_device.argtypes = [vp, ctypes.POINTER(vp)]
    lib.pob_gather_records.argtypes = [vp, vp, vp, vp, ctypes.c_uint32]
    lib.pob_emit_witness.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint64]
    lib.pob_write_wtns.argtypes = [vp, ctypes.c_uint32, ctypes.c_char_p]
    lib.pob_emit_begin.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint64]
    lib.pob_emit_next.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.pob_emit_queue.argtypes = [vp, ctypes.c_uint32]
    lib.pob_emit_measure.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    lib.pob_time_kernel.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
    lib.pob_probe_check_kernel.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    lib.pob_debug_xor_bits.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64]
    lib.pob_debug_store_fault.argtypes = [vp, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    lib.pob_debug_poke.argtypes = [vp, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.pob_debug_emit_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.pob_emit_selfcheck.argtypes = [vp, ctypes.c_int]
    lib.pob_emit_selfcheck_alias.argtypes = [vp, vp, ctypes.c_uint64]
    lib.pob_set_inorder.argtypes = [vp, ctypes.c_int]
    lib.pob_emit_selfcheck_result.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
    lib.pob_debug_fr_inv.argtypes = [ctypes.c_int, vp, ctypes.c_uint32, vp, vp]
    lib.pob_debug_ref.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.pob_keccak256.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p]
    lib.pob_keccak256.restype = None
    lib.pob_pow_search.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_char_p]
    lib.pob_pow_search.restype = ctypes.c_int64
    lib.pob_pow_search_gpu.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_char_p]
    lib.pob_pow_search_gpu.restype = ctypes.c_int64
    _lib = lib
    return lib


EXPORTED_SYMBOLS = ["pob_plan_info", "pob_gadget_template", "pob_open", "pob_close", "pob_get_info", "pob_strerror", "pob_upload_inputs", "pob_upload_inputs_async", "pob_host_alloc", "pob_host_free", "pob_pack_json", "pob_pack_json_batch",
                    "pob_upload_inputs8", "pob_upload_inputs8_async", "pob_narrow_inputs", "pob_pack_json_batch8",
                    "pob_results_fetch", "pob_results_wait", "pob_emit_begin_reduced", "pob_reduced_map_pin", "pob_write_wtns_reduced", "pob_emit_measure_ex", "pob_generate",
                    "pob_constraint_check", "pob_sync", "pob_set_partner", "pob_results", "pob_results_device", "pob_results_records_device", "pob_gather_records", "pob_emit_witness",
                    "pob_write_wtns", "pob_emit_begin", "pob_emit_next", "pob_emit_queue", "pob_emit_measure", "pob_time_kernel", "pob_probe_check_kernel", "pob_debug_xor_bits", "pob_debug_store_fault", "pob_debug_stream_create", "pob_debug_stream_destroy", "pob_debug_poke", "pob_debug_ref", "pob_debug_emit_counters", "pob_debug_fr_inv", "pob_emit_selfcheck", "pob_emit_selfcheck_alias", "pob_emit_selfcheck_result", "pob_set_inorder", "pob_keccak256", "pob_pow_search", "pob_pow_search_gpu"]


def plan_info(main: str) -> PobInfo:
    """wire / storage-class counts of an instantiation from the host-side layout planner (no GPU needed)"""
    return plan_info_of(*parse_main(main))


_plan_cache: dict = {}


def plan_info_of(name: str, params) -> PobInfo:
    """wire / class / input counts of an instantiation (the layout planner walks the whole circuit: cached per instantiation)"""
    key = (name, tuple(int(v) for v in params))
    if key in _plan_cache:
        return _plan_cache[key]
    info = _plan_cache[key] = _plan_info_uncached(name, params)
    return info


def _plan_info_uncached(name: str, params) -> PobInfo:
    circuit, cparams = circuit_of(name, params)
    info = PobInfo()
    rc = load_library().pob_plan_info(circuit, _limbs(cparams), len(cparams), ctypes.byref(info))
    if rc != 0:
        raise ValueError(f"pob_plan_info rc={rc}: unsupported instantiation {name}{tuple(params)}")
    return info


def keccak256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    load_library().pob_keccak256(bytes(data), len(data), out)
    return out.raw


def parse_main(main: str):
    """'ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)' -> ('ProofOfBurn', [16, 4, ...]) (reference tests/test.py:31)."""
    m = re.fullmatch(r"\s*(\w+)\s*\((.*)\)\s*;?\s*", main, flags=re.S)
    if not m:
        raise ValueError(f"not a template instantiation: {main!r}")
    params = []
    if m.group(2).strip():
        for a in m.group(2).split(","):
            if not re.fullmatch(r"[\d\s\*\+\-\(\)]+", a):
                raise ValueError(f"unsupported template argument {a!r}")
            params.append(int(eval(a, {"__builtins__": {}})))
    return m.group(1), params


def to_field(v) -> int:
    """input.json scalar -> canonical Fr (the emitted loader accepts JSON numbers and base-10 / 0x strings, mod p)."""
    if isinstance(v, bool):
        return int(v)
    if isinstance(v, (int, np.integer)):
        return int(v) % P
    if isinstance(v, str):
        return int(v, 16 if v.lower().startswith("0x") else 10) % P
    raise TypeError(f"unsupported input value {v!r}")


def _scalar(v):
    """the reference's tests pass scalars as 1-element arrays too (tests/testcases/divide.py:4)"""
    while isinstance(v, (list, tuple)) and len(v) == 1:
        v = v[0]
    return to_field(v)


def _flat(v, out):
    if isinstance(v, (list, tuple, np.ndarray)):
        for x in v:
            _flat(x, out)
    else:
        out.append(to_field(v))


def _main_of(main):
    return parse_main(main) if isinstance(main, str) else (main[0], list(main[1]))


def pack_inputs(main, inputs: Sequence[dict], info: PobInfo | None = None):
    """list of input.json dicts -> (fr[n][nfr][32] uint8, sm[n][nsm] int32, forced_status[n]): the emitted loader (loadJson) in Python.
    The byte arrays (layers, blockHeader: 99 % of an input) go through numpy in one conversion per key; only values that numpy
    cannot hold as int64 (strings, huge ints -- the loader's mod-p path) fall back to per-element parsing.  Needs no GPU."""
    name, params = _main_of(main)
    info = info or plan_info_of(name, params)
    if name not in ("ProofOfBurn", "Spend"):
        return _pack_gadget_inputs(name, params, inputs, info)
    n = len(inputs)
    nfr, nsm = info.n_fr_inputs, info.n_sm_inputs
    fr = np.zeros((n, nfr, 32), dtype=np.uint8)
    sm = np.zeros((n, max(nsm, 1)), dtype=np.int32)
    forced = np.zeros(n, dtype=np.uint32)
    fr_names = POB_FR_INPUTS if name == "ProofOfBurn" else SPEND_FR_INPUTS
    sm_names = POB_SM_INPUTS if name == "ProofOfBurn" else []
    shapes = {}
    if name == "ProofOfBurn":
        shapes = {"layers": params[0] * params[1] * 136, "layerLens": params[0], "blockHeader": params[2] * 136}
    want = set(fr_names) | set(sm_names)
    for w, d in enumerate(inputs):
        keys = set(d.keys())
        if keys != want:
            raise KeyError(f"input {w}: missing {sorted(want - keys)} unexpected {sorted(keys - want)}")
        for k, key in enumerate(fr_names):
            fr[w, k] = np.frombuffer(_scalar(d[key]).to_bytes(32, "little"), dtype=np.uint8)
        col = 0
        for name in sm_names:
            cnt = shapes.get(name, 1)
            a = None
            if name in shapes:
                try:                       # fast path: a (nested) list of plain non-negative ints
                    a = np.asarray(d[name])
                    # only integer / bool data takes the shortcut: floats (1.7 would be truncated to 1) and strings (numpy's own
                    # parsing rules) go through to_field element by element, which refuses / parses them like the loader does
                    a = a.astype(np.int64).reshape(-1) if a.dtype.kind in "iub" else None
                    if a is not None and a.size and int(a.min()) < 0:
                        a = None
                except (ValueError, TypeError, OverflowError):
                    a = None
            if a is None:
                vals = []
                if name in shapes:
                    _flat(d[name], vals)
                else:
                    vals = [_scalar(d[name])]
                a = np.array([v if v < (1 << 31) else -1 for v in vals], dtype=np.int64)
            if a.size != cnt:
                raise ValueError(f"input {w}: {name} has {a.size} elements, circuit expects {cnt}")
            big = (a < 0) | (a >= (1 << 31))
            if big.any():      # not representable as a small input: every such input is range-checked in-circuit
                forced[w] = FAIL_INPUT_RANGE
                a = np.where(big, 0x7FFFFFFF, a)
            sm[w, col:col + cnt] = a.astype(np.int32)
            col += cnt
        assert col == nsm
    return fr, sm, forced


def _pack_gadget_inputs(name, params, inputs: Sequence[dict], info: PobInfo):
    """the loader for a gadget-level main: every input signal of the template must be present with its declared number of elements
    (the emitted loader aborts otherwise: KeyError / ValueError here); values are reduced mod p like loadJson does.  An "s" signal whose
    value does not fit int32 cannot be represented by the HIP gadgets: NotImplementedError (loud, never a wrong answer)."""
    spec = GADGET_INPUTS[name](params)
    n = len(inputs)
    nfr = sum(c for _, k, c in spec if k == "f")
    nsm = sum(c for _, k, c in spec if k == "s")
    if (nfr, nsm) != (info.n_fr_inputs, info.n_sm_inputs):
        raise RuntimeError(f"{name}: the loader's input table ({nfr} field / {nsm} small) disagrees with the planner ({info.n_fr_inputs} / {info.n_sm_inputs})")
    fr = np.zeros((n, max(nfr, 1), 32), dtype=np.uint8)
    sm = np.zeros((n, max(nsm, 1)), dtype=np.int32)
    want = {nm for nm, _, _ in spec}
    for w, d in enumerate(inputs):
        keys = set(d.keys())
        if keys != want:
            raise KeyError(f"input {w}: missing {sorted(want - keys)} unexpected {sorted(keys - want)}")
        cf = cs = 0
        for nm, kind, cnt in spec:
            vals = []
            _flat(d[nm], vals)
            if len(vals) != cnt:
                raise ValueError(f"input {w}: {nm} has {len(vals)} elements, {name} expects {cnt}")
            for v in vals:
                if kind == "f":
                    fr[w, cf] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
                    cf += 1
                else:
                    sv = v if v < (1 << 31) else v - P          # small negative values arrive as p - |v|
                    if not -(1 << 31) <= sv < (1 << 31):
                        raise NotImplementedError(f"input {w}: {nm} = {v} does not fit the int32 class the HIP gadget path keeps this signal in")
                    sm[w, cs] = sv
                    cs += 1
    return fr, sm, np.zeros(n, dtype=np.uint32)


class TextBatch:
    """n input.json texts as the (pointer, length) arrays the native loader takes, built once: a service that packs the same buffers again (bench.py cycles
    its batches) does not pay Python's per-text handling per call"""

    def __init__(self, texts: Sequence[bytes | str]):
        self.raw = [t if isinstance(t, bytes) else (t.encode() if isinstance(t, str) else bytes(t)) for t in texts]
        self.n = len(self.raw)
        self.ptrs = (ctypes.c_char_p * self.n)(*self.raw)
        self.lens = (ctypes.c_uint64 * self.n)(*[len(t) for t in self.raw])

    def __len__(self):
        return self.n


def pack_json(main, texts: "Sequence[bytes | str] | TextBatch", threads: int = 0, out: "PinnedInputs | None" = None, info: PobInfo | None = None):
    """input.json TEXTS -> (fr, sm, forced) through the native loader (pob_pack_json_batch: hand-written parser, `threads` host
    threads, 0 = all cores) -- same acceptance and the same bits as pack_inputs on the parsed dicts; out = pinned arrays to fill in place.
    Needs no GPU."""
    name, params = _main_of(main)
    info = info or plan_info_of(name, params)
    n = len(texts)
    nfr, nsm = info.n_fr_inputs, info.n_sm_inputs
    if out is not None:
        fr, sm, forced = out.fr, (None if out.compact else out.sm), out.forced
        assert fr.shape[0] == n
    else:
        fr = np.zeros((n, nfr, 32), dtype=np.uint8)
        sm = np.zeros((n, max(nsm, 1)), dtype=np.int32)
        forced = np.zeros(n, dtype=np.uint32)
    tb = texts if isinstance(texts, TextBatch) else TextBatch(texts)
    ptrs, lens = tb.ptrs, tb.lens
    circuit = 0 if name == "ProofOfBurn" else 1
    arr = (ctypes.c_uint64 * (4 * len(params)))()
    for i, v in enumerate(params):
        for k in range(4):
            arr[4 * i + k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    err = ctypes.create_string_buffer(512)
    if out is not None and out.compact:
        # straight into the byte form (pob_pack_json_batch8); a witness with more than EXC_CAP values outside 0..255 sends the batch through the int32 form
        rc = load_library().pob_pack_json_batch8(circuit, arr, len(params), ptrs, lens, n, threads, fr.ctypes.data, out.sm8.ctypes.data, out.exc.ctypes.data, forced.ctypes.data, err, 512)
        out.bytes_ok = rc == 0
        if rc == 0:
            return fr, None, forced
        if rc != E_RANGE:
            _raise_loader(rc, err)
        out.ensure_sm()
        sm = out.sm
    rc = load_library().pob_pack_json_batch(circuit, arr, len(params), ptrs, lens, n, threads, fr.ctypes.data, sm.ctypes.data, forced.ctypes.data, err, 512)
    if rc != 0:
        _raise_loader(rc, err)
    return fr, sm, forced


def _raise_loader(rc: int, err):
    """the loader's error as the exception the Python loader raises for the same input: a missing / unexpected key is a KeyError, a malformed value a ValueError; running out of
    memory (POB_E_NOMEM) says nothing about the input and is a MemoryError"""
    msg = err.value.decode(errors="replace")
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise (KeyError if "missing [" in msg else ValueError)(msg)


def pack_json8(main, texts: Sequence[bytes | str], threads: int = 0):
    """input.json TEXTS -> (fr, sm8, exc, forced): the native loader straight into the byte form (pob_pack_json_batch8), or None when a witness has more
    than EXC_CAP small inputs outside 0..255 (then pack_json's int32 rows are the form to upload).  Needs no GPU."""
    name, params = _main_of(main)
    info = plan_info_of(name, params)
    n, nfr, nsm = len(texts), info.n_fr_inputs, max(info.n_sm_inputs, 1)
    fr = np.zeros((n, nfr, 32), dtype=np.uint8); sm8 = np.zeros((n, nsm), dtype=np.uint8); exc = np.zeros((n, EXC_CAP), dtype=EXC_DTYPE); forced = np.zeros(n, dtype=np.uint32)
    tb = texts if isinstance(texts, TextBatch) else TextBatch(texts)
    ptrs, lens = tb.ptrs, tb.lens
    arr = (ctypes.c_uint64 * (4 * len(params)))()
    for i, v in enumerate(params):
        for k in range(4):
            arr[4 * i + k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    err = ctypes.create_string_buffer(512)
    rc = load_library().pob_pack_json_batch8(0 if name == "ProofOfBurn" else 1, arr, len(params), ptrs, lens, n, threads, fr.ctypes.data, sm8.ctypes.data, exc.ctypes.data,
                                             forced.ctypes.data, err, 512)
    if rc == E_RANGE:
        return None
    if rc != 0:
        _raise_loader(rc, err)
    return fr, sm8, exc, forced


def widen_inputs(sm8: np.ndarray, exc: np.ndarray) -> np.ndarray:
    """the int32 rows a byte-form batch stands for (what the device's widening pass writes)"""
    sm = sm8.astype(np.int32)
    w, e = np.nonzero(exc["k"] != EXC_NONE)
    sm[w, exc["k"][w, e]] = exc["v"][w, e]
    return sm


def narrow_inputs(sm: np.ndarray):
    """int32 rows [n][nsm] -> (sm8 uint8 [n][nsm], exc [n][EXC_CAP]) of the byte form, or None when a witness has more than EXC_CAP values outside 0..255
    (pob_narrow_inputs; no GPU)"""
    sm = np.ascontiguousarray(sm, dtype=np.int32)
    n, nsm = sm.shape
    sm8 = np.empty((n, nsm), dtype=np.uint8)
    exc = np.empty((n, EXC_CAP), dtype=EXC_DTYPE)
    rc = load_library().pob_narrow_inputs(sm.ctypes.data, n, nsm, sm8.ctypes.data, exc.ctypes.data)
    if rc == E_RANGE:
        return None
    if rc != 0:
        raise ValueError("pob_narrow_inputs")
    return sm8, exc


@dataclass
class Result:
    status: int                 # 0 = ok; else (template id << 12) | source line of the first failing assert
    outputs: list | None        # public output signals (None when failed) -- tests/test.py:40-47,65-68
    check_status: int | None = None      # with_check: 0 = every === holds, else the first failing site; None = not asked for / not evaluated
    bad_wire: int | None = None          # with_check: lowest wire whose stored value contradicts its definition (None = none / not evaluated)
    evaluated: bool | None = None        # with_check: False when no constraint_check has run on this batch (check_status / bad_wire are None then)

    @property
    def ok(self) -> bool:
        return self.status == 0

    def message(self) -> str:
        if self.status == 0:
            return ""
        return f"Failed assert in template {_TPL.get(self.status >> 12, '?')} line {self.status & 0xFFF}"


class WitnessCalculator:
    """One (GPU, circuit instantiation) calculator: `./main_proof_of_burn` / `./main_spend` as an object."""

    def __init__(self, main: str, max_batch: int = 64, device: int = 0):
        name, params = parse_main(main) if isinstance(main, str) else main
        self.name, self.params = name, list(params)
        if name == "ProofOfBurn":
            if len(params) != 8:
                raise ValueError("ProofOfBurn(maxNumLayers, maxNodeBlocks, maxHeaderBlocks, minLeafAddressNibbles, amountBytes, "
                                 "powMinimumZeroBytes, maxIntendedBalance, maxActualBalance)")
            circuit = 0
            self.L, self.NB, self.HB = params[0], params[1], params[2]
        elif name == "Spend":
            circuit = 1
        self.lib = load_library()
        circuit, cparams = circuit_of(name, params)       # (a gadget-level main: circuit 2, template id in front of the parameters)
        self.is_gadget = circuit == 2
        self.device = device
        self.h = ctypes.c_void_p()
        rc = self.lib.pob_open(device, circuit, _limbs(cparams), len(cparams), max_batch, ctypes.byref(self.h))
        if rc != 0:
            msg = self.lib.pob_strerror(self.h).decode() if self.h else "pob_open failed"
            raise RuntimeError(f"pob_open: {msg} (rc={rc})")
        self.info = PobInfo()
        self._ck(self.lib.pob_get_info(self.h, ctypes.byref(self.info)))
        self.max_batch = max_batch
        self.n = 0
        self._forced = None
        self._next = None                    # (n, forced) of an uploaded batch that has not been generated yet
        self._forced_fetched = None

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(f"libpob_hip: {self.lib.pob_strerror(self.h).decode()} (rc={rc})")

    def close(self):
        if getattr(self, "h", None):
            self.lib.pob_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nwitness(self) -> int:
        return int(self.info.n_witness)

    # ------------------------------------------------------------------ input packing (the emitted loadJson)
    def pack(self, inputs: Sequence[dict]):
        """list of input.json dicts -> (fr[n][nfr][32] uint8, sm[n][nsm] int32, forced_status[n]) (module-level pack_inputs)"""
        return pack_inputs((self.name, self.params), inputs, self.info)

    def pack_json(self, texts: "Sequence[bytes | str] | TextBatch", threads: int = 0, out: "PinnedInputs | None" = None):
        """input.json TEXTS -> (fr, sm, forced) through the native loader (module-level pack_json)"""
        return pack_json((self.name, self.params), texts, threads, out, self.info)

    # ------------------------------------------------------------------ the calculator
    def upload(self, inputs: Sequence[dict]):
        fr, sm, forced = self.pack(inputs)
        self.upload_packed(fr, sm, forced)

    def upload_packed(self, fr: np.ndarray, sm: np.ndarray, forced: np.ndarray | None = None):
        n = fr.shape[0]
        if n > self.max_batch:
            raise ValueError(f"batch {n} exceeds max_batch {self.max_batch}")
        fr = np.ascontiguousarray(fr, dtype=np.uint8)
        sm = np.ascontiguousarray(sm, dtype=np.int32)
        self._ck(self.lib.pob_upload_inputs(self.h, fr.ctypes.data, sm.ctypes.data, n))
        self._next = (n, np.array(forced, dtype=np.uint32) if forced is not None else np.zeros(n, dtype=np.uint32))

    def upload_pinned_async(self, pin: "PinnedInputs", stream: int | None = None) -> int:
        """service-loop upload of a PinnedInputs batch: the byte form when the batch holds it (pob_upload_inputs8_async: 11.4 KB per production witness on
        the wire instead of 43.8), else the int32 rows; returns the bytes that cross PCIe"""
        n = pin.fr.shape[0]
        if n > self.max_batch:
            raise ValueError(f"batch {n} exceeds max_batch {self.max_batch}")
        if pin.compact and pin.bytes_ok:
            self._ck(self.lib.pob_upload_inputs8_async(self.h, pin.fr.ctypes.data, pin.sm8.ctypes.data, pin.exc.ctypes.data, n, ctypes.c_void_p(stream) if stream else None))
            self._next = (n, np.array(pin.forced, dtype=np.uint32))
            return pin.fr.nbytes + pin.sm8.nbytes + pin.exc.nbytes
        self.upload_packed_async(pin.fr, pin.sm, pin.forced, stream)
        return pin.fr.nbytes + pin.sm.nbytes

    def upload_packed8(self, fr: np.ndarray, sm8: np.ndarray, exc: np.ndarray, forced: np.ndarray | None = None):
        """synchronous upload of the byte form (pob_upload_inputs8; narrow_inputs makes it from int32 rows)"""
        n = fr.shape[0]
        if n > self.max_batch:
            raise ValueError(f"batch {n} exceeds max_batch {self.max_batch}")
        fr = np.ascontiguousarray(fr, dtype=np.uint8); sm8 = np.ascontiguousarray(sm8, dtype=np.uint8); exc = np.ascontiguousarray(exc, dtype=EXC_DTYPE)
        assert sm8.shape[0] == n and exc.shape == (n, EXC_CAP)
        self._ck(self.lib.pob_upload_inputs8(self.h, fr.ctypes.data, sm8.ctypes.data, exc.ctypes.data, n))
        self._next = (n, np.array(forced, dtype=np.uint32) if forced is not None else np.zeros(n, dtype=np.uint32))

    def upload_packed_async(self, fr: np.ndarray, sm: np.ndarray, forced: np.ndarray | None = None, stream: int | None = None):
        """service-loop upload (pob_upload_inputs_async): fr / sm must live in pinned memory (PinnedInputs) and stay untouched until the
        batch's results are in; the copy is ordered behind this calculator's previous generation, the next generate() behind the copy"""
        n = fr.shape[0]
        if n > self.max_batch:
            raise ValueError(f"batch {n} exceeds max_batch {self.max_batch}")
        assert fr.dtype == np.uint8 and sm.dtype == np.int32 and fr.flags.c_contiguous and sm.flags.c_contiguous
        self._ck(self.lib.pob_upload_inputs_async(self.h, fr.ctypes.data, sm.ctypes.data, n, ctypes.c_void_p(stream) if stream else None))
        self._next = (n, np.array(forced, dtype=np.uint32) if forced is not None else np.zeros(n, dtype=np.uint32))

    def generate(self, stream: int | None = None):
        """enqueue the generation of the uploaded batch (which becomes the current one: the inputs are double-buffered, a batch may be
        uploaded while the previous one is still being generated / evaluated / read); without a new upload: the same inputs again"""
        self._ck(self.lib.pob_generate(self.h, ctypes.c_void_p(stream) if stream else None))
        if self._next is not None:
            self.n, self._forced = self._next
            self._next = None

    def constraint_check(self, stream: int | None = None):
        self._ck(self.lib.pob_constraint_check(self.h, ctypes.c_void_p(stream) if stream else None))

    def sync(self):
        self._ck(self.lib.pob_sync(self.h))

    def set_partner(self, other: "WitnessCalculator | None"):
        """two-batch pipeline (include/pob_hip.h pob_set_partner): link this calculator with the one working on the neighbouring batch"""
        self._ck(self.lib.pob_set_partner(self.h, other.h if other is not None else None))

    def fetch_records(self):
        """enqueue the D2H copy of this batch's result records behind its evaluation (pob_results_fetch); does not block"""
        self._ck(self.lib.pob_results_fetch(self.h))
        self._forced_fetched = self._forced

    def wait_records(self) -> np.ndarray:
        """block on THAT copy only and return the records as a structured array (fields status, check_status, bad_wire, commitment[32]);
        the view aliases the handle's pinned buffer: valid until the fetch after the next one"""
        p, n = ctypes.c_void_p(), ctypes.c_uint32()
        self._ck(self.lib.pob_results_wait(self.h, ctypes.byref(p), ctypes.byref(n)))
        buf = (ctypes.c_uint8 * (RECORD_DTYPE.itemsize * n.value)).from_address(p.value)
        rec = np.frombuffer(buf, dtype=RECORD_DTYPE)
        f = self._forced_fetched
        if f is not None and f.any():        # inputs the loader could not represent (FAIL_INPUT_RANGE): failed whatever the device computed
            rec = rec.copy()
            rec["status"] = np.where(f != 0, f, rec["status"])
        return rec

    def results(self, with_check: bool = False) -> list[Result]:
        n = self.n
        status = np.zeros(n, dtype=np.uint32)
        outs = np.zeros((n, 32), dtype=np.uint8)
        chk = np.zeros(n, dtype=np.uint32)
        bad = np.zeros(n, dtype=np.uint32)
        self._ck(self.lib.pob_results(self.h, status.ctypes.data, outs.ctypes.data,
                                      chk.ctypes.data if with_check else None, bad.ctypes.data if with_check else None))
        res = []
        for i in range(n):
            st = int(self._forced[i]) or int(status[i])
            if self.is_gadget:                 # the main's outputs are the first wires of its witness (tests/test.py:40-47)
                out = None if st else self.output_signals(i)
            else:
                out = None if st else [int.from_bytes(outs[i].tobytes(), "little")]
            r = Result(st, out)
            if with_check:
                r.evaluated = int(chk[i]) != NOT_EVALUATED
                if r.evaluated:
                    r.check_status = 0 if chk[i] == CLEAN else int(chk[i])
                    r.bad_wire = None if bad[i] == CLEAN else int(bad[i])
            res.append(r)
        return res

    def calculate(self, inputs: dict | Sequence[dict], check: bool = False) -> list[Result]:
        """input.json dict(s) -> per-witness Result; the whole batch runs on the GPU."""
        if isinstance(inputs, dict):
            inputs = [inputs]
        self.upload(inputs)
        self.generate()
        if check:
            self.constraint_check()
        return self.results(with_check=check)

    def output_signals(self, idx: int) -> list:
        """witness[1 .. nOutputs] of witness idx, read through the emitter's first window (what the reference's patched main.cpp prints)"""
        nout = int(self.info.n_outputs)
        if nout == 0:
            return []
        for w0, view in self.witness_windows(idx, max(nout + 1, min(self.nwitness, 1 << 16))):
            assert w0 == 0 and view.size >= 32 * (nout + 1)
            b = view[32:32 * (nout + 1)].tobytes()
            return [int.from_bytes(b[32 * k:32 * k + 32], "little") for k in range(nout)]
        raise RuntimeError("empty witness")

    def witness_payload(self, idx: int = 0, out: np.ndarray | None = None) -> np.ndarray:
        """canonical 32-byte LE values of witness `idx` (the .wtns section 2 payload) as a uint8 array; `out`: a caller's buffer of that size to fill (a caller that
        looks at many witnesses reuses one: a fresh 6.9 GB array is two seconds of page faults)"""
        if out is None:
            out = np.empty(32 * self.nwitness, dtype=np.uint8)
        elif out.dtype != np.uint8 or out.size != 32 * self.nwitness or not out.flags.c_contiguous:
            raise ValueError("out: a contiguous uint8 array of 32 bytes per wire")
        self._ck(self.lib.pob_emit_witness(self.h, idx, out.ctypes.data, out.nbytes))
        return out

    def write_wtns(self, idx: int, path: str):
        self._ck(self.lib.pob_write_wtns(self.h, idx, os.fsencode(path)))

    def _keep_array(self, keep) -> np.ndarray:
        """a ReducedMap (circuit_model/o1.py) or an array of kept O0 wire indices -> the PRIVATE, read-only uint32 copy the library has pinned
        (pob_reduced_map_pin) for this map.  The library recognises a pinned map by address and length and then promises itself that its contents have not
        changed, so what is pinned is never the caller's array (the caller may mutate it in place): a private copy, recognised by a digest of the caller's contents
        (crc32 + adler32 over the 86 MB of the production map: ~40 ms per call, against ~9 ms of device-side hashing per emission that the pin saves and a silently stale map
        that it prevents) AND an element-wise comparison on a digest hit.  The library holds ONE pin, so this object keeps ONE copy: the map pinned last (a second map
        replaces it -- an earlier map handed in again is copied and pinned again, never served from a stale cache entry that the library would hash on every emission)."""
        import zlib
        k = np.ascontiguousarray(getattr(keep, "keep", keep), dtype=np.uint32)
        if k.ndim != 1 or k.size == 0:
            raise ValueError("keep: a non-empty 1-D array of wire indices")
        cur = getattr(self, "_keep_pin", None)           # (digest, the private read-only copy the library has pinned)
        if cur is not None and k is cur[1]:              # our own pinned copy handed back (witness_payload_reduced -> witness_windows): immutable, no digest needed
            return k
        key = (k.size, zlib.crc32(k), zlib.adler32(k))
        if cur is not None and cur[0] == key and np.array_equal(cur[1], k):
            return cur[1]
        own = np.array(k, dtype=np.uint32, copy=True)
        own.setflags(write=False)
        self._ck(self.lib.pob_reduced_map_pin(self.h, own.ctypes.data, own.size))
        self._keep_pin = (key, own)
        return own

    def write_wtns_reduced(self, idx: int, path: str, keep):
        """O1-style reduced .wtns (circuit_model.o1.reduce_map): only the surviving wires are expanded on the GPU and cross PCIe
        (pob_write_wtns_reduced); `keep` = sorted O0 wire indices (wire 0 first) or an object with a .keep array"""
        k = self._keep_array(keep)
        self._ck(self.lib.pob_write_wtns_reduced(self.h, idx, k.ctypes.data, k.size, os.fsencode(path)))

    def witness_windows(self, idx: int = 0, window_wires: int = 0, keep=None):
        """stream the canonical payload of witness idx: yields (first_wire, uint8 view [n_wires * 32]) per window; a view is valid
        until the next iteration (it aliases the handle's pinned buffer).  keep: the reduced payload (positions count kept wires)"""
        if keep is None:
            self._ck(self.lib.pob_emit_begin(self.h, idx, window_wires))
        else:
            k = self._keep_array(keep)
            self._ck(self.lib.pob_emit_begin_reduced(self.h, idx, k.ctypes.data, k.size, window_wires))
        p, w0, wn = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        while True:
            self._ck(self.lib.pob_emit_next(self.h, ctypes.byref(p), ctypes.byref(w0), ctypes.byref(wn)))
            if wn.value == 0:
                return
            buf = (ctypes.c_uint8 * (32 * wn.value)).from_address(p.value)
            yield w0.value, np.frombuffer(buf, dtype=np.uint8)

    def set_inorder(self, on=True):
        """every launch of this calculator on the caller's stream, in dependency order, no side streams (pob_set_inorder): for jobs that keep several calculators in flight.
        on = 3: the same with the FUSED launch -- the Poseidon blocks share a launch with the header's / layers' sponge chain, which does not depend on them;
        on | 4: the Keccak round blocks are evaluated by the launch that writes them (k_rounds_gc: the evaluation's loads come from L2 / the Infinity Cache), and
        constraint_check skips its round kernel unless a debug poke touched the vector in between"""
        self._ck(self.lib.pob_set_inorder(self.h, int(on)))

    def emit_selfcheck(self, enable: bool = True):
        """every following emission (O0 or reduced) evaluates the derived wires' own relations on the values written into its windows (pob_emit_selfcheck)"""
        self._ck(self.lib.pob_emit_selfcheck(self.h, 1 if enable else 0))

    def emit_selfcheck_alias(self, alias):
        """reduced emissions: the class representative of every O0 wire (circuit_model/o1.py O1Map or its .alias array; -1 = pinned to a constant), so that the check
        evaluates a site whose wires were dropped on their representatives (pob_emit_selfcheck_alias); None clears it"""
        if alias is None:
            self._sc_alias = None
            self._ck(self.lib.pob_emit_selfcheck_alias(self.h, None, 0))
            return
        if hasattr(alias, "const_wires"):        # an O1Map: representatives, and the constants coded as -1 - c (c < 2^30) / INT32_MIN
            a = np.array(alias.alias, dtype=np.int64)
            cv = np.array([(-1 - v) if v < (1 << 30) else -(1 << 31) for v in alias.const_values], dtype=np.int64)
            a[np.asarray(alias.const_wires, dtype=np.int64)] = cv
            a = a.astype(np.int32)
        else:
            a = np.ascontiguousarray(alias, dtype=np.int32)
        self._sc_alias = a                       # (the library reads it when the next reduced emission begins)
        self._ck(self.lib.pob_emit_selfcheck_alias(self.h, a.ctypes.data, a.size))

    def emit_selfcheck_result(self) -> dict:
        """of the last complete self-checked emission: relations checked / skipped (wires in two windows) and the lowest violated wire (None = none)"""
        c, s, w = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32()
        self._ck(self.lib.pob_emit_selfcheck_result(self.h, ctypes.byref(c), ctypes.byref(s), ctypes.byref(w)))
        return {"checked": int(c.value), "skipped": int(s.value), "first_bad_wire": None if w.value == 0xFFFFFFFF else int(w.value)}

    def emit_queue(self, next_idx: int):
        """announce the witness emitted after the current / next one (pob_emit_queue): its first window is expanded behind the current one's last"""
        self._ck(self.lib.pob_emit_queue(self.h, next_idx))

    def witness_payload_reduced(self, idx: int, keep, window_wires: int = 0) -> np.ndarray:
        """the reduced payload (32 B per kept wire) as one array"""
        k = self._keep_array(keep)
        out = np.empty(32 * k.size, dtype=np.uint8)
        for w0, view in self.witness_windows(idx, window_wires, keep=k):
            out[32 * w0:32 * w0 + view.size] = view
        return out

    def emit_throughput(self, first_idx: int = 0, count: int = 1, window_wires: int = 0, keep=None):
        """(seconds, bytes) of `count` witnesses emitted back to back into pinned host memory (keep: the reduced form).  The first
        emission of a calculator also allocates its window buffers: call twice for the steady state."""
        sec, nb = ctypes.c_double(), ctypes.c_uint64()
        k = None if keep is None else self._keep_array(keep)
        self._ck(self.lib.pob_emit_measure_ex(self.h, first_idx, count, window_wires, k.ctypes.data if k is not None else None, k.size if k is not None else 0,
                                              ctypes.byref(sec), ctypes.byref(nb)))
        return sec.value, nb.value

    def time_kernel(self, which: int, iters: int = 5, stream: int | None = None) -> float:
        ms = ctypes.c_float()
        self._ck(self.lib.pob_time_kernel(self.h, which, iters, ctypes.c_void_p(stream) if stream else None, ctypes.byref(ms)))
        return float(ms.value)

    def probe_check_kernel(self, enable: bool = True, read: bool = False) -> float | None:
        """HIP events around the Keccak round evaluation kernel of every following constraint_check (pob_probe_check_kernel); read=True
        returns the duration (ms) of the last one, whose batch must be complete"""
        ms = ctypes.c_float()
        self._ck(self.lib.pob_probe_check_kernel(self.h, 1 if enable else 0, ctypes.byref(ms) if read else None))
        return float(ms.value) if read else None

    # ------------------------------------------------------------------ test hooks of the constraint evaluator
    CLASS_BIT, CLASS_SM, CLASS_FR = 0, 1, 2

    def class_sizes(self) -> dict:
        return {self.CLASS_BIT: int(self.info.n_bit), self.CLASS_SM: int(self.info.n_sm), self.CLASS_FR: int(self.info.n_fr)}

    def poke(self, cls: int, index: int, lane: int, xor_mask: int = 1, sub: int = 0, group: int = 0):
        """XOR one stored value of one witness of the resident vector (storage class, rank in the class, lane)"""
        self._ck(self.lib.pob_debug_poke(self.h, cls, group, index, sub, lane, xor_mask))

    UNKNOWN_WIRE = 0xFFFFFFFF

    def store_fault(self, index: int, lane_mask: int, group: int = 0, cls: int = 0):
        """arm ONE corrupted store of the next generation (pob_debug_store_fault; calculators with set_inorder(... | 4)): the word of storage class cls at rank `index`.
        Returns the wire the evaluation that rides with the generation must report for the witnesses of lane_mask -- known for a round block's BIT word and an input's SM
        row; UNKNOWN_WIRE for any other word: a store of a G unit if a riding unit stores it at all (then exactly the witnesses of lane_mask are flagged, at the wire
        itself), nothing happens otherwise -- or None if no such word exists"""
        w = ctypes.c_uint32()
        rc = self.lib.pob_debug_store_fault(self.h, cls, group, index, lane_mask, ctypes.byref(w))
        if rc == -1 and "not a stored word" in self.lib.pob_strerror(self.h).decode():
            return None
        self._ck(rc)
        return int(w.value)

    def emit_counters(self, reset: bool = True) -> dict:
        """IsZero.inv wires written by the emitter since the last reset, by path (pob_debug_emit_counters)"""
        out = (ctypes.c_uint64 * 4)()
        self._ck(self.lib.pob_debug_emit_counters(self.h, out, 1 if reset else 0))
        return {"table": int(out[0]), "fermat": int(out[1]), "field_nonzero": int(out[2]), "field_zero": int(out[3])}

    def debug_ref(self, name: str, k: int = 0):
        cls, idx, wire = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = self.lib.pob_debug_ref(self.h, name.encode(), k, ctypes.byref(cls), ctypes.byref(idx), ctypes.byref(wire))
        if rc != 0:
            raise KeyError(f"no debug ref {name}[{k}]")
        return cls.value, idx.value, wire.value

    def records_device_ptr(self) -> int:
        """device pointer of the packed per-witness result records {u32 status, u32 check_status, u32 bad_wire, u8 commitment[32]} (44 B each)"""
        a = ctypes.c_void_p()
        self._ck(self.lib.pob_results_records_device(self.h, ctypes.byref(a)))
        return a.value


    def gather_records_rccl(self, comm: int, out_ptr: int, n_per_rank: int, stream: int = 0):
        """the multi-GPU path's one collective through the C ABI (pob_gather_records): all-gather of this calculator's device records over the caller's ncclComm_t into
        device memory at out_ptr (nranks x n_per_rank x 44 bytes), on `stream`, ordered behind the batch's evaluation"""
        self._ck(self.lib.pob_gather_records(self.h, ctypes.c_void_p(comm), ctypes.c_void_p(stream), ctypes.c_void_p(out_ptr), n_per_rank))

    def results_device_ptrs(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self._ck(self.lib.pob_results_device(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value


class PinnedInputs:
    """packed inputs of one batch in pinned host memory (pob_host_alloc), the source of upload_pinned_async / upload_packed_async.
    compact (default): the small inputs in the byte form -- sm8 [n][nsm] uint8 + exc [n][EXC_CAP] -- which pack_json fills natively and fill() narrows into;
    the int32 rows (sm) are allocated only when a batch does not fit the byte form (bytes_ok False) or compact is off."""

    def __init__(self, calc: "WitnessCalculator", n: int, compact: bool = True):
        self.lib = calc.lib
        self.n, self.nfr, self.nsm = n, int(calc.info.n_fr_inputs), max(int(calc.info.n_sm_inputs), 1)
        self.compact = bool(compact) and int(calc.info.n_sm_inputs) > 0
        self.bytes_ok = False
        self._p = []
        self.fr = self._alloc((n, self.nfr, 32), np.uint8)
        self.sm = self.sm8 = self.exc = None
        if self.compact:
            self.sm8 = self._alloc((n, self.nsm), np.uint8)
            self.exc = self._alloc((n, EXC_CAP), EXC_DTYPE)
        else:
            self.sm = self._alloc((n, self.nsm), np.int32)
            if int(calc.info.n_sm_inputs) == 0:
                self.sm[...] = 0             # (the dummy column of a circuit without small inputs: the loader never writes it)
        self.forced = np.zeros(n, dtype=np.uint32)

    def _alloc(self, shape, dt):
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        p = ctypes.c_void_p()
        if self.lib.pob_host_alloc(ctypes.byref(p), nbytes) != 0:
            raise MemoryError("pob_host_alloc")
        self._p.append(p)
        return np.frombuffer((ctypes.c_uint8 * nbytes).from_address(p.value), dtype=dt).reshape(shape)

    def ensure_sm(self):
        if self.sm is None:
            self.sm = self._alloc((self.n, self.nsm), np.int32)

    def fill(self, fr, sm, forced=None):
        self.fr[...] = fr
        self.forced[...] = 0 if forced is None else forced
        if self.compact:
            sm = np.ascontiguousarray(sm, dtype=np.int32)
            rc = self.lib.pob_narrow_inputs(sm.ctypes.data, self.n, self.nsm, self.sm8.ctypes.data, self.exc.ctypes.data)
            self.bytes_ok = rc == 0
            if rc == 0:
                return self
            self.ensure_sm()
        self.sm[...] = sm
        return self

    def widened(self) -> np.ndarray:
        """the int32 rows this batch stands for (what the device widens the byte form into)"""
        if not (self.compact and self.bytes_ok):
            return np.array(self.sm)
        return widen_inputs(self.sm8, self.exc)

    def free(self):
        for p in self._p:
            self.lib.pob_host_free(p)
        self._p = []


def wtns_header(nwitness: int) -> bytes:
    """iden3 .wtns preamble; pob_write_wtns writes the same bytes natively."""
    return (b"wtns" + (2).to_bytes(4, "little") + (2).to_bytes(4, "little") + (1).to_bytes(4, "little") + (40).to_bytes(8, "little")
            + (32).to_bytes(4, "little") + P.to_bytes(32, "little") + nwitness.to_bytes(4, "little")
            + (2).to_bytes(4, "little") + (32 * nwitness).to_bytes(8, "little"))


def calculate_witness(main: str, input_json: str | dict, wtns_path: str | None = None, device: int = 0) -> Result:
    """`./<circuit> input.json witness.wtns` (reference Makefile:4-5) as a function."""
    inp = input_json
    if isinstance(input_json, (str, os.PathLike)):
        with open(input_json) as f:
            inp = json.load(f)
    calc = WitnessCalculator(main, max_batch=1, device=device)
    try:
        r = calc.calculate(inp)[0]
        if r.ok and wtns_path:
            calc.write_wtns(0, wtns_path)
        return r
    finally:
        calc.close()
