"""The synthetic input producer (proof_of_burn_amd/inputs.py) yields inputs the ORACLE accepts, with the commitment the
reference's formula predicts -- at the fixture's instantiation sizes so it runs in seconds on CPU."""
import random

from proof_of_burn_amd import inputs as G
from tests import oracle_ffi as O
from tests import refshim


def test_rlp_and_hex_prefix_against_refshim():
    rng = random.Random(1)
    for _ in range(20):
        item = [rng.randbytes(rng.randrange(0, 40)) for _ in range(rng.randrange(0, 18))] + [rng.randrange(1 << 70)]
        assert G.rlp(item) == refshim.rlp_encode(item)
    assert G.hex_prefix_leaf([1, 2, 3, 4]) == bytes([0x20, 0x12, 0x34]) and G.hex_prefix_leaf([2, 3, 4]) == bytes([0x32, 0x34])


def test_pow_search_matches_reference_vectors():
    # tests/testcases/proof_of_work.py: burnKey 812 has one leading zero byte, 47109 two (revealAmount 234, extra 345)
    assert G.pow_search(811, 234, 345, 1) == 812
    assert G.pow_search(47108, 234, 345, 2) == 47109


def test_synthetic_small_instantiation_is_valid():
    params = (4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)
    b = G.synthetic_batch(2, depth=3, seed=7, distinct_keys=1, params=params)
    main = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"
    for inp, c in zip(b.inputs, b.commitments):
        assert O.run_main(main, inp) == [c]
    bad = dict(b.inputs[0]); bad["numLeafAddressNibbles"] = str(int(bad["numLeafAddressNibbles"]) - 1)
    assert O.run_main(main, bad) is None


def test_from_account_proof_rebuilds_the_reference_fixture():
    """the producer path of reference tests/main.py:65-178: feed the fixture's own UNPADDED proof nodes and header (what
    eth_getProof / the block RPC return) and get the fixture's input.json back -- nibble count from the leaf's hex-prefix,
    zero padding, unused layerLens = 256"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "test_pob_input.json")) as f:
        fix = json.load(f)
    params = (4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)
    nodes = [bytes(fix["layers"][i][:fix["layerLens"][i]]) for i in range(fix["numLayers"])]
    header = bytes(fix["blockHeader"][:fix["blockHeaderLen"]])
    assert G.leaf_address_nibbles(nodes[-1]) == int(fix["numLeafAddressNibbles"])
    got = G.from_account_proof(nodes, header, int(fix["burnKey"]), int(fix["actualBalance"]), int(fix["intendedBalance"]), int(fix["revealAmount"]),
                               int(fix["burnExtraCommitment"]), int(fix["_proofExtraCommitment"]), fix["byteSecurityRelax"], params=params)
    assert set(got) == set(fix)
    for k in fix:
        if isinstance(fix[k], list):
            assert got[k] == fix[k], k
        else:
            assert int(got[k]) == int(fix[k]), k
    main = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"
    assert O.run_main(main, got) == O.run_main(main, fix) and O.run_main(main, got) is not None
    # and a synthetic proof goes through the same parser: producer and parser agree
    b = G.synthetic_batch(1, depth=3, seed=11, distinct_keys=1, params=params)
    inp = b.inputs[0]
    nodes = [bytes(inp["layers"][i][:inp["layerLens"][i]]) for i in range(inp["numLayers"])]
    again = G.from_account_proof(nodes, bytes(inp["blockHeader"][:inp["blockHeaderLen"]]), int(inp["burnKey"]), int(inp["actualBalance"]),
                                 int(inp["intendedBalance"]), int(inp["revealAmount"]), int(inp["burnExtraCommitment"]), int(inp["_proofExtraCommitment"]),
                                 inp["byteSecurityRelax"], params=params)
    assert again["layers"] == inp["layers"] and again["layerLens"] == inp["layerLens"] and int(again["numLeafAddressNibbles"]) == int(inp["numLeafAddressNibbles"])
    # malformed: state root not where the circuit reads it
    import pytest
    with pytest.raises(ValueError):
        G.from_account_proof(nodes, bytes(inp["blockHeader"][1:inp["blockHeaderLen"]]), 1, 1, 1, 1, 1, 1, params=params)


def test_pack_vectorised_equals_elementwise():
    """the Python loader (witness.pack_inputs): the numpy fast path and the per-element (string / huge int) path agree"""
    import numpy as np
    from proof_of_burn_amd import witness as W
    params = (4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)
    b = G.synthetic_batch(2, depth=2, seed=3, distinct_keys=1, params=params)

    main = "ProofOfBurn(4, 4, 5, 20, 31, 2, 10 ** 18, 10 ** 19)"
    a = W.pack_inputs(main, b.inputs)
    slow = [dict(d, layers=[[str(x) for x in row] for row in d["layers"]], blockHeader=[hex(x) for x in d["blockHeader"]]) for d in b.inputs]
    c = W.pack_inputs(main, slow)
    for x, y in zip(a, c):
        assert np.array_equal(x, y)
    big = dict(b.inputs[0]); big["layerLens"] = list(big["layerLens"]); big["layerLens"][3] = 2 ** 40
    assert W.pack_inputs(main, [big])[2][0] == W.FAIL_INPUT_RANGE


def test_rlp_decode_checks_every_length():
    """truncated / over-long encodings and malformed leaves raise instead of being cut by slicing"""
    import pytest
    good = G.rlp([G.hex_prefix_leaf([1, 2, 3]), b"value"])
    assert G.rlp_decode(good) == [G.hex_prefix_leaf([1, 2, 3]), b"value"] and G.leaf_address_nibbles(good) == 3
    for bad in (good[:-1], good + b"\x00", bytes([0xB8]), bytes([0x85, 1, 2]), bytes([0xC5, 0x83, 1]), bytes([0xF8, 0x40, 1, 2]), b""):
        with pytest.raises(ValueError):
            G.rlp_decode(bad)
    for node in (G.rlp([b"", b"v"]), G.rlp([b"\x20", b"a", b"b"]), G.rlp(b"\x20\x12"), G.rlp([[b"\x20"], b"v"]), G.rlp([b"\x10", b"v"])):
        with pytest.raises(ValueError):
            G.leaf_address_nibbles(node)
