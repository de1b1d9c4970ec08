import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import mvoracle
    mvoracle.lib()
    return mvoracle


@pytest.fixture(scope="session")
def mv():
    """the product binding; GPU tests fail loudly if the HIP library is missing"""
    import mvtools_amd
    mvtools_amd.lib()
    return mvtools_amd
