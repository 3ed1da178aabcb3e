"""Symbolic model of the circuits at --O0: wire numbering, signal names (.sym), rank-1 constraints (.r1cs), an independent
constraint checker for .wtns payloads, and the O1-style reduced witness map.  See core.py."""
from .core import Circuit, P  # noqa: F401
from .circuits import circuit  # noqa: F401
