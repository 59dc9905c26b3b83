import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import bbme_oracle
    bbme_oracle.build()
    bbme_oracle.lib()
    return bbme_oracle


@pytest.fixture(scope="session")
def bbme():
    """The product package; libbbme.so is built in-tree with hipcc when missing."""
    from blockbasedmotionestimation_amd import build as _build
    _build.build()
    import blockbasedmotionestimation_amd as pkg
    return pkg


GOLDEN = os.path.join(ROOT, "tests", "golden")
