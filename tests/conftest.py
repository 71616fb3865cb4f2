import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cognitive-radio-network_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure libcrnsense.so and the oracle exist (build() cross-compiles without a GPU)."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "cognitive-radio-network_amd", "libcrnsense.so")
    orc = os.path.join(ROOT, "oracle", "libcrn_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        g.build()
    return True
