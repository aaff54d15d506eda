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
    """Paths of the prebuilt reference binaries (oracle/_ref: the checkers) and of the product drop-in (kmc_amd/bin/kmc_hip), or None where they
    were never built."""
    d = os.path.join(ROOT, "oracle", "_ref")
    paths = {n: os.path.join(d, n) for n in ("kmc", "kmc_oracle", "kmc_tools")}
    paths["kmc_hip"] = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hip")
    if all(os.path.exists(p) for p in paths.values()):
        return paths
    return None
