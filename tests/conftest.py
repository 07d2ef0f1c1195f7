import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _fresh_tuning_switches():
    """The library caches its MDG_* switches; a test that flipped one (helpers.set_switch) must not leak it into the next."""
    yield
    try:
        from madrigal_amd import _lib
        if _lib._lib is not None:
            _lib._lib.mdg_tuning_reload()
    except Exception:
        pass
