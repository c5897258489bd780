import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import crlib
    return crlib.Oracle()


@pytest.fixture(scope="session")
def gpu():
    # torch first: a process holds ONE HIP runtime, the first one loaded (torch bundles its own copy of libamdhip64);
    # when libcrgpu.so brings in the system's before torch is imported, torch later finds "no HIP GPUs"
    import torch
    torch.cuda.init()
    import comprox_amd
    g = comprox_amd.CrGpu(0)
    yield g
    g.close()
