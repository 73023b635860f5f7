import os
import sys
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    from onnxstream_amd import osgpu
    g = osgpu.Gpu(0)   # raises loudly if libosgpu.so is missing or no GPU is visible: no CPU fallback
    yield g
    g.close()
