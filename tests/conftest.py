import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/fx_oracle.c) -- checker only, never the product path."""
    import fxoracle
    fxoracle.lib()
    return fxoracle


def load_golden(name):
    import json
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def fixture_bytes(fn):
    import gzip
    p = os.path.join(DATA, fn)
    return gzip.open(p).read() if fn.endswith(".gz") else open(p, "rb").read()
