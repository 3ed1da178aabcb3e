"""python -m proof_of_burn_amd.circuit_model <command> "<Template(params)>" ...

    info   MAIN                      wire / constraint counts of the model
    sym    MAIN out.sym              circom-style .sym (signal, witness index, component, name) of the --O0 numbering
    r1cs   MAIN out.r1cs             iden3 binary .r1cs (every <== and ===)
    check  MAIN witness.wtns         evaluate every constraint on a witness (`snarkjs wtns check` equivalent); exit 1 on violations
    o1     MAIN in.wtns out.wtns     O1-style reduced witness (signal-to-signal / constant copies dropped)
    keepmap MAIN                     derive that map and store it compressed under circuit_model/data/ (keepmap.py: the device-side
                                     reduced emission of the bench and the GPU tests read it from there)
"""
import sys

import numpy as np

from . import check as CK
from .circuits import circuit
from .core import P


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 2:
        print(__doc__, file=sys.stderr)
        return 2
    if argv[0] == "keepmap":
        from . import keepmap
        print(keepmap.generate(argv[1]))
        return 0
    cmd, c = argv[0], circuit(argv[1])
    if cmd == "info":
        print(f"wires {c.n_wires} constraints {c.n_constraints} outputs {c.n_outputs} inputs {c.n_inputs}")
    elif cmd == "sym":
        print("sha256", c.write_sym(argv[2]))
    elif cmd == "r1cs":
        c.write_r1cs(argv[2])
    elif cmd == "check":
        bad = CK.check_witness(c, CK.Witness.from_wtns(argv[2]))
        for row, wires in bad:
            print(f"constraint {row} violated (wires {wires[:8]})", file=sys.stderr)
        print("OK" if not bad else f"{len(bad)}+ violations")
        return 1 if bad else 0
    elif cmd == "o1":
        from .o1 import reduce_map
        from ..witness import wtns_header
        m = reduce_map(c)
        data = np.fromfile(argv[2], dtype=np.uint8)
        red = m.reduce(data[76:])
        with open(argv[3], "wb") as f:
            f.write(wtns_header(len(m.keep)))
            f.write(red.tobytes())
        print(f"kept {len(m.keep)} of {c.n_wires} wires ({m.n_linear_rows} of {m.n_rows} constraints are linear)")
    else:
        print(__doc__, file=sys.stderr)
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
