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


@pytest.fixture(autouse=True)
def _producer_stats_where_tested(request, monkeypatch):
    """Producer-side GroupNorm statistics are off by default (ops.PRODUCER_GN_STATS: open store hazard,
    profiles/r03_conv_phases.txt); the tests that exercise them switch them on for themselves."""
    name = request.node.name
    if any(k in name for k in ("stats", "under_load", "producer", "split_k", "concat_segments", "groupnorm_split_matches",
                                "presplit_vs_oracle")):
        from lidarcrafter_amd import ops as K
        monkeypatch.setattr(K, "PRODUCER_GN_STATS", True)
    yield
