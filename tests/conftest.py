import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "julia-spira_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()   # builds oracle/libspira_oracle.so with gcc when missing
    return oracle_py


@pytest.fixture(scope="session")
def binding():
    from spira_hip import _binding
    if not os.path.exists(_binding.LIB_PATH):
        _binding.build_library()   # hipcc cross-compiles without a GPU
    _binding.lib()
    return _binding


@pytest.fixture(scope="session")
def gpu(binding):
    """The HIP path itself; fails loudly (never skips to a fallback) when no device is usable."""
    n = binding.device_count()
    assert n >= 1, "no HIP device visible: " + binding.lib().spira_last_error().decode()
    return binding
