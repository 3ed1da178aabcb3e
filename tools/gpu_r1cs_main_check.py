#!/usr/bin/env python3
"""The independent R1CS referee (proof_of_burn_amd/circuit_model) on a GPU-EMITTED .wtns of the production instantiation: one witness of a
synthetic 10-layer batch is generated and evaluated on the GPU, written as a 6.9 GB .wtns file by pob_write_wtns, read back from disk and
checked against all 215 962 292 rows.  Prints the summary committed as profiles/round3_r1cs_main_check_gpu.txt.
    python tools/gpu_r1cs_main_check.py [scratch_dir]"""
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proof_of_burn_amd import WitnessCalculator, inputs as gen  # noqa: E402
from proof_of_burn_amd.circuit_model import check as CK, circuit  # noqa: E402

MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"


def main():
    scratch = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
    batch = gen.synthetic_batch(4, depth=10, seed=0xB0B, distinct_keys=2)
    calc = WitnessCalculator(MAIN, max_batch=4)
    res = calc.calculate(batch.inputs, check=True)
    assert all(r.ok and r.check_status == 0 and r.bad_wire is None for r in res)
    assert [r.outputs[0] for r in res] == batch.commitments
    path = os.path.join(scratch, "main_gpu.wtns")
    t0 = time.time()
    calc.write_wtns(2, path)
    t_w = time.time() - t0
    calc.close()
    t0 = time.time()
    c = circuit(MAIN)
    print(f"model built in {time.time() - t0:.1f} s: wires {c.n_wires} (planner {calc.nwitness}), constraints {c.n_constraints}, outputs {c.n_outputs}, inputs {c.n_inputs}")
    t0 = time.time()
    w = CK.Witness.from_wtns(path)
    bad = CK.check_witness(c, w)
    print(f"GPU-emitted .wtns of witness 2 of a synthetic 10-layer batch ({os.path.getsize(path)} bytes, written in {t_w:.1f} s by pob_write_wtns, commitment "
          f"{batch.commitments[2]}): {len(bad)} violated constraints of {c.n_constraints} (read + checked in {time.time() - t0:.1f} s)")
    print("peak RSS GB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6)
    os.remove(path)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
