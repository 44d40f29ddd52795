import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wheat-3dgs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import _poison                      # W3D_TEST_FILL=zeros|ff|nan|small|rand|unit: patterned torch.empty (tests/_poison.py)
    _poison.install_from_env()
    import cpu_twins                    # torch stand-ins for the HIP-backed operations: the product has no CPU path (tests/cpu_twins.py)
    cpu_twins.install()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
