import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs the HIP library)")


def scene_path(name: str) -> str:
    return os.path.join(GOLDEN, "scenes", name + ".toml")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker)."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def rt():
    """The product package; loading the C ABI fails loudly when the library has not been built."""
    import raytracer_rs_amd
    from raytracer_rs_amd import _lib

    _lib.lib()
    return raytracer_rs_amd


@pytest.fixture(scope="session")
def gpu(rt):
    """A HIP device must be present for tests marked gpu; no fallback is acceptable there."""
    import ctypes

    from raytracer_rs_amd import _lib

    n = ctypes.c_int(0)
    rc = _lib.lib().rt_device_count(ctypes.byref(n))
    assert rc == 0 and n.value >= 1, "tests marked gpu need a HIP device"
    return n.value
