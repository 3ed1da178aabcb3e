"""Shared bodies of the constraint-evaluator corruption tests.  They run on the GPU (`tests/test_gpu_parity.py`, `-m gpu`) and, at
reduced sizes, on the CPU through the HIP-on-fibers shim (`tests/test_hostsim_cpu.py`): same product code either way.

Method: a batch of 64 IDENTICAL valid witnesses = one group (one wavefront, lane = witness).  One pass pokes a DIFFERENT stored
wire in each of the lanes 1..63 and leaves lane 0 alone; then one evaluation must flag exactly the poked lanes -- a failing lane
never disturbs another, and lane 0 is the control.  Storage classes: BIT (1 bit), SM (int32 rows: bytes, lengths, and the
IsZero.inv hints stored as their operand), FR (8 x 32-bit limb
planes, Montgomery: Poseidon state, SubstringCheck M[] / exists operands, balances).
"""
from __future__ import annotations

import numpy as np

BIT, SM, FR = 0, 1, 2
NAMES = {BIT: "BIT", SM: "SM", FR: "FR"}


def _flagged(r) -> bool:
    return r.bad_wire is not None or r.check_status != 0


def open_identical_batch(pkg, main: str, inp: dict, n: int = 64):
    calc = pkg.WitnessCalculator(main, max_batch=n)
    res = calc.calculate([inp] * n, check=True)
    assert all(r.ok and not _flagged(r) for r in res), "clean batch must evaluate clean"
    return calc


def sweep(calc, cls: int, indices, rng, lanes_per_pass: int = 63):
    """poke every index of `indices` (storage ranks of class cls), lanes_per_pass per evaluation; returns the mis-detections"""
    missed = []
    indices = list(indices)
    n = calc.n
    for p0 in range(0, len(indices), lanes_per_pass):
        chunk = indices[p0:p0 + lanes_per_pass]
        pokes = []
        for k, idx in enumerate(chunk):
            lane = 1 + k
            if cls == FR:
                sub, mask = int(rng.integers(0, 8)), 1 << int(rng.integers(0, 28))
            elif cls == SM:
                sub, mask = 0, 1 << int(rng.integers(0, 4))
            else:
                sub, mask = 0, 1
            calc.poke(cls, idx, lane, mask, sub)
            pokes.append((idx, lane, mask, sub))
        calc.constraint_check()
        res = calc.results(with_check=True)
        for idx, lane, mask, sub in pokes:
            calc.poke(cls, idx, lane, mask, sub)            # restore
        want = [False] * n
        for _, lane, _, _ in pokes:
            want[lane] = True
        got = [_flagged(r) for r in res]
        if got != want:
            for idx, lane, mask, sub in pokes:
                if not got[lane]:
                    missed.append((NAMES[cls], idx, lane, mask, sub, "not detected"))
            for lane in range(n):
                if got[lane] and not want[lane]:
                    missed.append((NAMES[cls], None, lane, None, None, f"clean lane flagged: bad_wire={res[lane].bad_wire} status={res[lane].check_status}"))
    return missed


def uniform_sweep(calc, counts: dict, seed: int = 7):
    """counts: class -> number of uniformly drawn storage ranks"""
    rng = np.random.default_rng(seed)
    sizes = calc.class_sizes()
    missed, done = [], {}
    for cls, cnt in counts.items():
        if sizes[cls] == 0 or cnt == 0:
            continue
        idx = sorted(set(rng.integers(0, sizes[cls], cnt).tolist()))
        done[NAMES[cls]] = len(idx)
        missed += sweep(calc, cls, idx, rng)
    calc.constraint_check()
    assert all(not _flagged(r) for r in calc.results(with_check=True)), "restored vector must evaluate clean again"
    return missed, done


def named_pokes(calc, names):
    """names: list of (debug-ref name, k).  Each poke alone (lane 1), lanes 0 and 2 are controls; the reported bad wire must not
    lie after the poked wire (its own definition is checked) nor far before it (only the enclosing component reads it)."""
    bad = []
    for name, k in names:
        cls, idx, wire = calc.debug_ref(name, k)
        mask = 1 if cls == BIT else 4
        calc.poke(cls, idx, 1, mask, 0)
        calc.constraint_check()
        res = calc.results(with_check=True)
        calc.poke(cls, idx, 1, mask, 0)
        ok = _flagged(res[1]) and not _flagged(res[0]) and not _flagged(res[2])
        if ok and res[1].bad_wire is not None and not (wire - 16384 <= res[1].bad_wire <= wire):
            ok = False
        if not ok:
            bad.append((name, k, NAMES[cls], idx, wire, res[1].bad_wire, res[1].check_status, _flagged(res[0]), _flagged(res[2])))
    calc.constraint_check()
    assert all(not _flagged(r) for r in calc.results(with_check=True))
    return bad
