import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def product():
    """The MI355X library through the reference's own C API binding.  No fallback: a missing
    library is a failure, not a skip."""
    import helpers
    from srla_amd import capi
    assert os.path.exists(helpers.PRODUCT_SO), "srla_amd/libsrla_mi355x.so is not built (run __graft_entry__.build())"
    return capi.EncoderLib(helpers.PRODUCT_SO)
