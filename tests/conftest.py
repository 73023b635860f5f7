import os
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


def pytest_runtest_setup(item):
    """GPU-box rule (round 5): the parity tests proper need the reference oracle (oracle/_ref/libonnxstream_ref.so travels with the snapshot: git-ignored, not
    gpurun-ignored).  A box where it did not arrive must FAIL the GPU suite, not report green with every parity leg skipped; OSA_ALLOW_NO_ORACLE=1 turns the
    failure back into the per-test skips (a developer box that cannot build the oracle)."""
    if item.get_closest_marker("gpu") is None or os.environ.get("OSA_ALLOW_NO_ORACLE") == "1":
        return
    from oracle import ref as oref
    if not oref.available():
        pytest.fail("oracle/_ref/libonnxstream_ref.so is missing on a GPU box: the parity legs of the GPU suite would be skipped "
                    "(build it where /root/reference is mounted -- __graft_entry__.build() -- and let it travel; OSA_ALLOW_NO_ORACLE=1 to skip instead)", pytrace=False)


@pytest.fixture(scope="session")
def gpu():
    from onnxstream_amd import osgpu
    g = osgpu.Gpu(0)   # raises loudly if libosgpu.so is missing or no GPU is visible: no CPU fallback
    yield g
    g.close()
