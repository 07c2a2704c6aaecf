import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """tests that depend on the box's RCCL bootstrap outside this process (a C program creating its own communicators) run
    LAST: the suite is run with -x, and an infrastructure hiccup there must not hide the parity tests behind it."""
    late = [it for it in items if "dist_consumer" in it.name]
    if late:
        items[:] = [it for it in items if "dist_consumer" not in it.name] + late


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # the HIP path must be the one that runs: fail loudly if the library is absent
    from gptorch_amd import _native
    _native.lib()
    return torch.device("cuda:0")
