import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "no_canary: run this gpu test without the guard-banded allocations of tests/_guard.py (dispatch-count / timing tests)")


@pytest.fixture(autouse=True)
def canary(request):
    """Every `-m gpu` test runs with guard-banded, NaN-poisoned device allocations in the product modules (tests/_guard.py) and fails if a kernel wrote outside
    a tensor it was handed -- whatever the test itself compares.  RBA_TEST_CANARY=0 switches it off (A/B of the suite's run time)."""
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("no_canary") is not None \
            or os.environ.get("RBA_TEST_CANARY", "1") == "0":
        yield None
        return
    from tests import _guard
    _guard.install()
    try:
        yield _guard
        _guard.check()
    finally:
        _guard.REGISTRY.entries.clear()
        _guard.REGISTRY.bytes = 0
        _guard.REGISTRY.pending_error = None
        _guard.uninstall()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture
def knobs():
    """The block of a test that selects a kernel VARIANT runs on the knobs build of the library (librba_hip_knobs.so: the same sources with the tuning knobs of
    csrc/knobs.h as writable ints); the product library exports no such state.  `ctypes.c_int.in_dll(_lib.load(), name)` then finds the knob."""
    from rba_amd import _lib
    with _lib.use_library(_lib.KNOBS_LIB_PATH) as lib:
        yield lib
