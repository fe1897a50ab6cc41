import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with `-m gpu`")


@pytest.fixture(scope="session")
def native_lib():
    """The in-tree gfx950 library; (re)built on demand with hipcc, which cross-compiles without a GPU."""
    from gaussian_gan_decoder_amd import build, _capi
    build.build()
    return _capi.load()
