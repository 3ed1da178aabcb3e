import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden_suites():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "suites.json")) as f:
        return json.load(f)
