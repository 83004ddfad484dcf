import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently skip: only skip when the marker was not requested
    if "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="needs an MI355X; run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords and not _has_gpu():
            item.add_marker(skip)


@pytest.fixture(scope="session")
def cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle_c
    oracle_c.build()
    return oracle_c


@pytest.fixture(scope="session")
def built_lib():
    """libtmx.so built in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tendermintx_amd", "csrc")])
    from tendermintx_amd import _lib
    return _lib.lib()
