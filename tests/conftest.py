import os
import sys
import subprocess
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neural-color-transfer_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bind
    return oracle_bind.load()


@pytest.fixture(scope="session")
def ctx():
    """GPU context. Fails loudly (no fallback) when the HIP extension or the device is missing."""
    import nct
    c = nct.Context(0)
    yield c
    c.close()
