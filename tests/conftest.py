import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the product library and the oracle once per session (CPU-only: hipcc cross-compiles)."""
    from oracle import orc
    from rustlight_amd import build as rl_build

    rl_build.build()
    orc.build()
    return True


@pytest.fixture(scope="session")
def cbox64(built):
    from rustlight_amd import scenes

    return scenes.cbox(64, 64)


@pytest.fixture(scope="session")
def orc_cbox64(cbox64):
    from oracle import orc

    return orc.Scene(cbox64)
