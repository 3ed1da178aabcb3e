"""proof_of_burn_amd -- MI355X-native witness generation for the WORM proof_of_burn / spend circuits.

The package holds only what the hot path needs: `csrc/` (hand-written HIP for gfx950 + the C ABI declared in
include/pob_hip.h) and `witness.py`, the host-side mirror of the reference calculator interface
(input.json dict -> outputs / failure / .wtns).  See DESIGN.md.
"""
from .witness import (P, Result, WitnessCalculator, calculate_witness, keccak256, load_library, parse_main, plan_info,  # noqa: F401
                      wtns_header, EXPORTED_SYMBOLS, LIB_PATH, PinnedInputs, TextBatch, RECORD_DTYPE)
