import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from proof_of_burn_amd import WitnessCalculator, inputs as gen
MAIN = "ProofOfBurn(16, 4, 16, 50, 31, 2, 10 ** 19, 10 ** 20)"
batch = gen.synthetic_batch(4, depth=10, seed=0xB0B, distinct_keys=4)
calc = WitnessCalculator(MAIN, max_batch=4)
res = calc.calculate(batch.inputs, check=True)
for r in res: print(r.ok, r.check_status, r.bad_wire, hex(r.check_status or 0))
