import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def engine_mod():
    import blackhole_simulation_amd as bh
    if not os.path.exists(bh.library_path()):
        # a checkout without the built artefact (they are git-ignored): compile it with hipcc --
        # the tests never fall back to anything else
        bh.build_library()
    return bh
