import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle runs on torch-CPU: a GPU box grants 16 cores per GPU but shows all 256 of the host, and a thread per
    # visible core oversubscribes that share by orders of magnitude (same rule as bench.py's cpu_baseline leg)
    import torch
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, visible)))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


@pytest.fixture(scope="session")
def reference_available():
    return os.path.isdir(os.path.join(REFERENCE, "models"))
