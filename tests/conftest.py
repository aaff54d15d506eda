import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """The product context (libkmc_hip.so on device 0). No fallback: a missing library or GPU is a failure."""
    from kmc_amd import capi

    c = capi.Context((0,))
    yield c
    c.close()


@pytest.fixture(scope="session")
def ref_bins():
    """Paths of the prebuilt reference binaries (oracle/_ref), or None where they were never built."""
    d = os.path.join(ROOT, "oracle", "_ref")
    need = ["kmc", "kmc_oracle", "kmc_hip", "kmc_tools"]
    if all(os.path.exists(os.path.join(d, n)) for n in need):
        return {n: os.path.join(d, n) for n in need}
    return None
