import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vapoursynth-mvtools_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Order of the GPU run: the parity suite first, then the filter shell (tests/test_vs_shim.py), the one-GPU sharding case LAST -- with
    `-x` a failure of a late file must not hide the cases of another (round 4: 31 shell cases were never reached behind a sharding failure)."""
    last = [it for it in items if "test_sharding.py" in it.nodeid and it.get_closest_marker("gpu") is not None]
    if last:
        items[:] = [it for it in items if it not in last] + last


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


@pytest.fixture(scope="session")
def fakedev(tmp_path_factory, oracle):
    """path of the TEST DOUBLE of the device layer (tests/fakedev/mvx_fakedev.c), built into a temporary directory: LD_PRELOADed in front of
    libmvtools_amd.so it lets the real VapourSynth plugin run in the real mini host on a CPU-only machine (the oracle computes).  Test
    infrastructure; the product never loads it."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("fakedev") / "libmvx_fakedev.so")
    odir = os.path.join(root, "oracle")
    subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-shared", "-fPIC", "-I" + os.path.join(root, "include"), "-I" + odir,
                           os.path.join(root, "tests", "fakedev", "mvx_fakedev.c"), "-o", so, "-L" + odir, "-lmvoracle", "-Wl,-rpath," + odir, "-ldl", "-lpthread"])
    return so
