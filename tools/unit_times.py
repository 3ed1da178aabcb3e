#!/usr/bin/env python3
"""Per-unit-kind kernel times (each kind alone on the GPU): generation and constraint evaluation.
    python tools/unit_times.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proof_of_burn_amd import WitnessCalculator, inputs as gen  # noqa: E402

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
KINDS = ("U_POB_INPUT U_POB_RANGE U_POB_LAYER_ASSERT U_POB_HDR_ASSERT U_POB_POSEIDONS U_BAH_PRE U_BAH_POST U_KB_HEAD U_KB_RANGE U_KB_SELROW "
         "U_KB_POST U_POB_N2B U_PC_PRE U_PC_POST U_POB_LASTLAYER U_POB_LASTLAYER_RANGE U_POB_LASTLEN U_POB_LEAF U_POB_LAYER_POST U_SC_M U_SC_RANGE "
         "U_SC_SUMS U_POB_LASTLEAF U_RL_A U_RL_SLROW U_RL_ACC U_RL_B U_POW_PRE U_POW_POST U_POB_FINAL U_ABS_RANGE U_LD_HEAD U_LD_SELR U_LD_TAIL "
         "U_POB_INPUT_FR U_RL_ACC_B U_RL_ACC_C U_SP_INPUT U_SP_HEAD U_SC_MI CK_POS_SEG CK_SR_COLS CK_SL_ROWS CK_N2BE CK_CAT CK_RL_B2 CK_RL_B3 U_POS_WIDE").split()      # = circuits.hpp UnitKind, from 1
FAMS = "F_MISC F_RANGE F_SELROW F_LD F_RL F_SC F_POS F_N2B".split()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    batch = gen.synthetic_batch(B, depth=10, seed=0xB0B, distinct_keys=16)
    calc = WitnessCalculator(MAIN, max_batch=B)
    res = calc.calculate(batch.inputs, check=True)
    assert all(r.ok for r in res)
    print(f"{'kind':24s} {'gen ms':>9s} {'check ms':>9s}")
    tg = tc = 0.0
    for k, name in enumerate(KINDS, start=1):
        g = calc.time_kernel(200 + k, iters=3)
        c = calc.time_kernel(100 + k, iters=3)
        if g or c:
            print(f"{name:24s} {g:9.3f} {c:9.3f}")
            tg += g; tc += c
    print(f"{'sum':24s} {tg:9.3f} {tc:9.3f}")
    print("evaluation kernel of each family, alone on the device:")
    for f, name in enumerate(FAMS):
        print(f"{name:24s} {'':9s} {calc.time_kernel(300 + f, iters=3):9.3f}")
    for w, name in ((0, "k_rounds<GEN>"), (1, "k_rounds<CHECK>"), (2, "all G check launches"), (3, "k_chain<GEN>")):
        print(f"{name:24s} {calc.time_kernel(w, iters=3):9.3f}")
    calc.close()


if __name__ == "__main__":
    main()
