import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Every test process -- and the rank processes the data-parallel tests spawn, which inherit the environment -- gets the 16
# hardware queues `import lidog_amd` asks for, set here BEFORE anything starts the HIP runtime.  Run on its own
# (`pytest -k unequal`), a two-rank test used to spawn workers with the runtime's default of 4: the step's streams then
# share queues, and two ranks spin-waiting for each other's statistics message on one GPU hang (seen in round 5, every
# build back to round 4; inside the full suite an earlier `import lidog_amd` in the parent had always set it).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
