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
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def sar():
    """The product package with libsar_hip.so loaded (fails loudly if it is not built)."""
    import strange_attractor_renderer_amd as S
    from strange_attractor_renderer_amd import _abi, build
    if not os.path.exists(_abi.LIB_PATH) or not os.path.exists(_abi.HOOKS_PATH):
        build.build_library()          # same recipe as __graft_entry__.build(); the product itself never builds lazily
    # the suite turns A/B and test options (include/sar_test_hooks.h): it runs on the HOOKS build — the product's own object files
    # plus sar_runtime_set_test_option. (SAR_LIBRARY still names any other build.) smoke() and bench.py load the product.
    _abi.use_hooks_build()
    S.load_library()
    return S


@pytest.fixture(scope="session")
def gpu(sar):
    n = sar.device_count()
    if n <= 0:
        pytest.fail("gpu-marked test needs a HIP device and found none (no CPU fallback exists)")
    return n
