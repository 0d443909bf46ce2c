import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "spawns: starts child processes that use the GPU; scheduled before every other "
                                       "test, while this process has not initialised the GPU yet")


def pytest_collection_modifyitems(config, items):
    """Tests that start GPU child processes run first: on the GPU pool a process that has initialised the GPU must
    not exec another program, and forked children of such a process are refused too."""
    items.sort(key=lambda it: 0 if it.get_closest_marker("spawns") else 1)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def lib():
    """The HIP C-ABI library; GPU tests fail loudly if it is missing."""
    from facerecognition_infrenceengine_amd import _lib
    return _lib.load()
