import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (checker only)."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def H():
    """The product package; building it is part of the fixture (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    import hpsdf_loader
    return hpsdf_loader.load()


@pytest.fixture(scope="session")
def ctx(H):
    c = H.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def golden():
    out = {}
    for n in ("tables", "blocks", "kats"):
        with open(os.path.join(GOLDEN, n + ".json")) as f:
            out[n] = json.load(f)
    return out


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a
