import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's HIP runtime has to come up before libgulon_hip.so touches the device in this process (the
    # other order leaves torch with "No HIP GPUs are available"; sharded.HipEngine documents the same):
    # tests that hand torch device tensors to the C ABI share the process with tests that only use ctypes
    if os.path.exists("/dev/kfd"):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()


def bits(a):
    """View float32 array as uint32 for bit-exact comparison."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
