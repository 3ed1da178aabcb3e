"""AddressSanitizer + UndefinedBehaviorSanitizer over the product's own kernels and host scheduler (the unchanged sources of libpob_hip.so on the
HIP-on-fibers shim, tests/hostsim) and over the C oracle: the quick set of tools/run_sanitizers.py (Spend suite incl. the window / queue / reduced
emission paths, gadget mains through the run shim, the emitter's inverse paths).  The full set -- fixture instantiation, evaluator sweeps, every gadget
main, the pipelined service loop -- is `python tools/run_sanitizers.py full`; its log is committed under profiles/."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_quick_set_under_asan_and_ubsan():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_sanitizers.py"), "quick"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "findings: 0" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
