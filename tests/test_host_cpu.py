"""CPU-side tests of the product: the C-ABI library loads and exports every symbol of include/pob_hip.h, the host
planner's layout (written independently of the oracle) yields the oracle's wire counts, input packing mirrors the
emitted loader.  No compute call is made without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import proof_of_burn_amd as pkg
from proof_of_burn_amd import witness as W
from tests import oracle_ffi as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(pkg.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "pob_hip.h")).read()
    declared = set(re.findall(r"\b(pob_\w+)\s*\(", hdr))
    assert declared == set(pkg.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None


@pytest.mark.parametrize("main,w", [
    ("Spend(31)", 2_603_360),
    ("ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)", 64_355_038),
    ("ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)", 215_907_954),
])
def test_planner_wire_count_matches_model(main, w):
    info = pkg.plan_info(main)
    assert info.n_witness == w                              # SURVEY.md app. C / BASELINE.md section 2
    assert info.n_bit + info.n_sm + info.n_fr + info.n_sb + 1 == w      # every wire has exactly one storage class (+ the constant wire)


def test_planner_matches_oracle_on_other_instantiations():
    for main in ("ProofOfBurn(2, 1, 1, 20, 31, 2, 10 ** 18, 10 ** 19)", "ProofOfBurn(3, 2, 2, 20, 30, 2, 10 ** 18, 10 ** 19)", "Spend(16)"):
        name, prm = pkg.parse_main(main)
        if name == "ProofOfBurn":
            L, NB, HB = prm[:3]
            inp = {k: 0 for k in W.POB_FR_INPUTS + ["numLeafAddressNibbles", "numLayers", "blockHeaderLen", "byteSecurityRelax"]}
            inp.update(layers=[[0] * (136 * NB)] * L, layerLens=[0] * L, blockHeader=[0] * (136 * HB))
        else:
            inp = {k: 0 for k in W.SPEND_FR_INPUTS}
        assert O.run(main, inp).nwitness == pkg.plan_info(main).n_witness, main


def test_host_keccak_matches_reference_constant():
    with open(os.path.join(ROOT, "tests", "golden", "test_pob_input.json")) as f:
        pob = json.load(f)
    hdr = bytes(pob["blockHeader"][:pob["blockHeaderLen"]])
    # the one Keccak-256 value hard-coded in the reference (tests/testcases/proof_of_burn.py:22)
    assert pkg.keccak256(hdr).hex() == "e36499b50da290131c3fa32d4f60717c8c529ae1bc3a216f32d05c05fe80368d"
    assert pkg.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"


def test_parse_main_and_values():
    assert pkg.parse_main("ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)") == ("ProofOfBurn", [16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20])
    assert W.to_field("0x10") == 16 and W.to_field(str(W.P + 5)) == 5 and W.to_field(2 ** 256 - 1) == (2 ** 256 - 1) % W.P
    assert W._scalar([7]) == 7


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pkg.WitnessCalculator("Spend(31)", max_batch=1)     # hipSetDevice/hipMalloc fail: no CPU fallback exists


def test_wtns_header_equals_oracle_writer():
    r = O.run("Poseidon(2)", {"inputs": [1, 2]})
    assert bytes(r.wtns_numpy()[:76]) == pkg.wtns_header(r.nwitness)


def test_plan_rejects_unsupported_and_oversized_instantiations():
    """template parameters are validated limb-exactly and an instantiation whose wire / class counts would overflow the layout's
    32-bit offsets is refused up front instead of wrapping silently (C ABI: POB_E_ARG; here: ValueError from plan_info)"""
    from proof_of_burn_amd import plan_info
    ok = plan_info("ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)")
    assert int(ok.n_witness) == 215_907_954 and int(ok.n_bit) < (1 << 29)
    for bad in ("ProofOfBurn(32, 8, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",           # 276 Keccak-f permutations: ~690 M BIT wires
                "ProofOfBurn(64, 16, 32, 50, 31, 2, 10 ** 19, 10 ** 20)",
                f"ProofOfBurn({2 ** 64 + 16}, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",     # upper limbs must be zero
                "ProofOfBurn(16, 4, 16, 500, 31, 2, 10 ** 19, 10 ** 20)",                  # minLeafAddressNibbles > 64
                "ProofOfBurn(16, 4, 16, 50, 32, 2, 10 ** 19, 10 ** 20)",                   # amountBytes > 31
                "ProofOfBurn(16, 4, 16, 50, 31, 99, 10 ** 19, 10 ** 20)",                  # powMinimumZeroBytes > 32
                "ProofOfBurn(1, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)",
                "Spend(32)", "Spend(0)"):
        with pytest.raises(ValueError):
            plan_info(bad)
    # the balance bounds are field elements: values >= p are reduced, not truncated
    assert int(plan_info(f"ProofOfBurn(4, 4, 5, 20, 31, 2, {W.P + 10 ** 18}, 10 ** 19)").n_witness) == 64_355_038
