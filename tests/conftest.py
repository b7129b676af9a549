import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A test that hangs (a kernel that never ends, a lost wake-up) must fail with the stacks of its threads instead of holding the
    # whole run until somebody's outer limit kills it without a line of output: 15 minutes per test where pytest-timeout is installed
    # (the longest test, the 1 000-frame stream, takes three).
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(900, method="thread"))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding

    binding.build()
    return binding.lib()


@pytest.fixture
def knobs():
    """Test / experiment knobs of the library (mlm_debug_set): read by the next mlm_create, forgotten again after the test."""
    from mlmapping_amd import mlmap

    class Knobs:
        def set(self, name, value):
            mlmap.debug_set(name, int(value))

    yield Knobs()
    mlmap.debug_reset()
