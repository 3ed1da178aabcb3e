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
