import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _collect_between_gpu_tests(request):
    """GPU tests build trainers that hold reference cycles (model hooks pointing back at the trainer) together with captured
    hipGraphs, pinned staging blocks and graph memory pools.  Left to the cyclic collector they die at an arbitrary moment of
    a LATER test - their hipGraphExecDestroy / pool frees then run beside that test's captures and replays (one segmentation
    fault inside hipGraphLaunch in ~10 suite runs on this stack, always right behind a test that had dropped several trainers).
    Collecting at the test boundary, with the device idle, keeps every test's garbage inside its own slot."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
