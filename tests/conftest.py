import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


@pytest.fixture(params=[True, False], ids=["producer_stats", "stats_pass"])
def gn_stats_route(request, monkeypatch):
    """Both GroupNorm statistics routes, explicitly: entries emitted by the producing convolution (the default,
    ops.PRODUCER_GN_STATS) and one lc_groupnorm_stats pass per GroupNorm (LC_GN_PRODUCER_STATS=0).  A test that
    takes this fixture runs once per route; which route a test covers never depends on its name."""
    from lidarcrafter_amd import ops as K
    monkeypatch.setattr(K, "PRODUCER_GN_STATS", request.param)
    return request.param
