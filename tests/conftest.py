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
    """Make sure the native library and the oracle exist (cheap no-op when up to date)."""
    import __graft_entry__ as g
    if not (os.path.exists(os.path.join(ROOT, "htslib_amd", "libhtsgpu.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from tests import refutil
    return refutil.Oracle()


@pytest.fixture(scope="session")
def engine(built):
    # (no torch import here: the first one on a fresh box pages in for a minute or two, and the engine is plain HIP)
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no GPU")
    from htslib_amd import _native as nat
    return nat.Engine(0)
